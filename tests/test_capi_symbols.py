"""The C-ABI library loads and exports every symbol include/hnsw_amd.h declares (no compute)."""
import ctypes
import os
import re

from hnsw_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hnsw_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hnsw_amd_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name


def test_argument_errors_are_status_codes_not_crashes():
    L = capi.lib()
    h = ctypes.c_void_p()
    assert L.hnsw_amd_create(1, 16, 200, ctypes.byref(h)) == capi.E_ARG       # M < 2
    assert b"M" in L.hnsw_amd_last_error()
    assert L.hnsw_amd_create(16, 16, 200, ctypes.byref(h)) == capi.OK
    assert L.hnsw_amd_init(h, 0, 10) == capi.E_ARG
    assert L.hnsw_amd_load_graph(h, b"/nonexistent/graph.bin") == capi.E_IO
    assert L.hnsw_amd_insert(h, None) == capi.E_ARG
    L.hnsw_amd_destroy(h)


def test_missing_library_fails_loudly():
    """There is no CPU fallback: without libhnsw_amd.so the package refuses to work."""
    import subprocess
    import sys
    code = ("import os,sys; sys.path.insert(0, %r); os.environ['HNSW_AMD_LIB']='/nonexistent/libhnsw_amd.so';"
            "from hnsw_amd import capi\n"
            "try:\n    capi.lib()\nexcept ImportError as e:\n    print('REFUSED', e)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "REFUSED" in out.stdout and "no CPU fallback" in out.stdout


def test_product_package_never_imports_the_oracle():
    import re
    pkg = os.path.join(ROOT, "hnsw_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "liboracle" not in text, f


def test_one_hip_runtime_whatever_the_import_order():
    """PyTorch-ROCm ships its own libamdhip64; loading this library first must not bring a second
    copy into the process (two runtimes: the one that initialises second finds no device)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "%s\n"
            "print(len({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))\n")
    for order in ("from hnsw_amd import capi; capi.lib(); import torch",
                  "import torch; from hnsw_amd import capi; capi.lib()"):
        out = subprocess.run([sys.executable, "-c", code % (ROOT, order)], capture_output=True, text=True)
        assert out.stdout.strip() == "1", (order, out.stdout, out.stderr[-500:])
