"""saveGraph / loadGraph in the reference's own file format (src/hnsw.cpp:1178-1311)."""
import os

import numpy as np

from hnsw_amd import HNSW
from tests.helpers import GOLD, Golden

REF_FILE = os.path.join(GOLD, "int_d16_n1000_m5.savegraph.bin")


def test_load_file_written_by_the_reference():
    g = Golden("int_d16_n1000_m5")
    h = HNSW(5, 5, 20)
    h.loadGraph(REF_FILE)
    fg = h.export_flat()
    assert fg.same_adjacency(g.graph)
    assert h.vector_size_ == 16 and h.actual_node_count_ == 999


def test_save_is_byte_identical_to_the_reference_file(tmp_path):
    g = Golden("int_d16_n1000_m5")
    h = HNSW(5, 5, 20)
    h.init(16, 1000)
    h.insert_batch(g.base, threads=1)
    out = str(tmp_path / "ours.bin")
    h.saveGraph(out)
    assert open(out, "rb").read() == open(REF_FILE, "rb").read()


def test_round_trip_with_default_ctor_degrees(tmp_path):
    g = Golden("int_d32_n2000")
    h = HNSW(16, 16, 200)
    h.import_flat(g.graph, g.base)
    p = str(tmp_path / "g.bin")
    h.saveGraph(p)
    h2 = HNSW(16, 16, 200)
    h2.loadGraph(p)
    assert h2.export_flat().same_adjacency(g.graph)


def test_truncated_file_is_an_io_error(tmp_path):
    from hnsw_amd import capi
    import pytest
    data = open(REF_FILE, "rb").read()
    p = str(tmp_path / "cut.bin")
    open(p, "wb").write(data[: len(data) // 2])
    h = HNSW(5, 5, 20)
    with pytest.raises(capi.HnswAmdError) as e:
        h.loadGraph(p)
    assert e.value.code == capi.E_IO
