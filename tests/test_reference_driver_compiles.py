"""The reference's own driver, unmodified, against include/hnsw_amd.hpp.

src/main.cpp of the reference includes "hnsw.h"; here a two-line file of that name includes the drop-in header
and the reference's own settings.h (unchanged: COLLECT_STAT on, so hnsw.stat_ is used).  The driver is fed to
the compiler on stdin so that its quoted include is looked up in the stand-in directory, not next to the
reference's real hnsw.h.  Compiled (syntax + semantics) in its default mode and with LOAD_GRAPH, then linked
against libhnsw_amd.so.  Needs the reference tree, so it runs in the build container only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF_SRC, "main.cpp")), reason="reference tree not present")


def _compile(tmp_path, extra, after=()):
    shim = tmp_path / "shim"
    shim.mkdir(exist_ok=True)
    (shim / "hnsw.h").write_text('#pragma once\n#include "settings.h"\n#include "hnsw_amd.hpp"\n')
    cmd = ["g++", "-std=gnu++17", "-x", "c++", "-I", str(shim), "-I", os.path.join(ROOT, "include"),
           "-idirafter", REF_SRC] + extra + ["-"] + list(after)
    with open(os.path.join(REF_SRC, "main.cpp"), "rb") as src:
        return subprocess.run(cmd, stdin=src, capture_output=True, text=True, cwd=str(tmp_path))


@pytest.mark.parametrize("defines", [[], ["-DLOAD_GRAPH"]])
def test_unmodified_main_cpp_compiles_against_the_drop_in_header(tmp_path, defines):
    r = _compile(tmp_path, ["-fsyntax-only"] + defines)
    assert r.returncode == 0, r.stderr[-3000:]


def test_unmodified_main_cpp_links_against_the_library(tmp_path):
    exe = str(tmp_path / "ref_main")
    r = _compile(tmp_path, ["-O1", "-o", exe], after=["-L" + os.path.join(ROOT, "hnsw_amd"), "-lhnsw_amd",
                                                      "-Wl,-rpath," + os.path.join(ROOT, "hnsw_amd")])
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.exists(exe)
