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


def _layout(data):
    """Byte offsets inside a saveGraph file: (per layer: offset of ep_node_order, and per node the offset of its first
    (neighbour index, neighbour node_order) pair and its degree)."""
    import struct
    maxn, d, _ = struct.unpack_from("<iIi", data, 0)
    pos = 12 + 4 * maxn * d
    (nl,) = struct.unpack_from("<Q", data, pos); pos += 8
    counts = []
    for _ in range(nl):
        (c,) = struct.unpack_from("<Q", data, pos); pos += 8 + 8 * c
        counts.append(c)
    layers = []
    for c in counts:
        ep_at = pos; pos += 8
        nodes = []
        for _ in range(c):
            pos += 4
            (deg,) = struct.unpack_from("<Q", data, pos); pos += 8
            nodes.append((pos, deg)); pos += 8 * deg
        layers.append((ep_at, nodes))
    assert pos == len(data)
    return layers


def test_corrupt_graph_files_are_refused_and_leave_the_index_untouched(tmp_path):
    """What a search trusts about a loaded graph is checked on load (the GPU kernels index the vector table and
    the adjacency arrays with these numbers): entry points, neighbour ranges, position/node_order agreement,
    duplicates.  A refused file leaves the index as it was."""
    import struct
    from hnsw_amd import capi
    import pytest
    data = bytearray(open(REF_FILE, "rb").read())
    layers = _layout(data)
    g = Golden("int_d16_n1000_m5")
    h = HNSW(5, 5, 20)
    h.loadGraph(REF_FILE)

    def refused(mutated, what):
        p = str(tmp_path / "bad.bin")
        open(p, "wb").write(mutated)
        with pytest.raises(capi.HnswAmdError) as e:
            h.loadGraph(p)
        assert e.value.code in (capi.E_GRAPH, capi.E_IO), what
        assert h.export_flat().same_adjacency(g.graph), f"{what}: the index changed although the load failed"
        assert h.actual_node_count_ == 999

    # entry point's node_order does not belong to the entry point
    ep_at, nodes = layers[1]
    m = bytearray(data); struct.pack_into("<I", m, ep_at, struct.unpack_from("<I", m, ep_at)[0] ^ 1); refused(m, "ep_node_order")
    # a neighbour whose node_order is beyond the filled part of the index
    at, deg = next((a, dg) for a, dg in layers[0][1] if dg >= 2)
    m = bytearray(data); struct.pack_into("<I", m, at + 4, 5000); refused(m, "neighbour out of range")
    # a neighbour whose node_order disagrees with the node it points at
    m = bytearray(data); struct.pack_into("<I", m, at + 4, struct.unpack_from("<I", m, at + 4)[0] ^ 1); refused(m, "position / node_order")
    # the same neighbour twice in one list
    m = bytearray(data); m[at + 8:at + 16] = m[at:at + 8]; refused(m, "duplicate neighbour")
    # a declared degree beyond what the searches are built for
    m = bytearray(data); struct.pack_into("<Q", m, at - 8, 65); refused(m, "degree 65")
    # and the untouched file still loads
    h.loadGraph(REF_FILE)
    assert h.export_flat().same_adjacency(g.graph)
