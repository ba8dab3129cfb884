"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/icebin_hip.h declares, and fails loudly (no fallback) without a GPU."""
import ctypes as C
import os
import re

import pytest

from icebin_amd import _capi
from icebin_amd.build import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build_library()
    return _capi.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "icebin_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ibh_[A-Za-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libicebin_hip.so does not export %s" % n
    assert set(_capi.exported_symbols()) == set(names), set(_capi.exported_symbols()) ^ set(names)


def test_sparse_set_host_logic(lib):
    from icebin_amd import SparseSet
    import numpy as np
    s = SparseSet(100, [7, 3, 99])
    assert s.sparse_extent() == 100 and s.dense_extent() == 3 and s.to_sparse().tolist() == [7, 3, 99]
    i = SparseSet.identity(5)
    assert i.to_sparse().tolist() == [0, 1, 2, 3, 4] and i.sparse_extent() == 5
    with pytest.raises(_capi.IcebinHipError, match="duplicate sparse index"):
        SparseSet(100, [1, 1])
    with pytest.raises(_capi.IcebinHipError, match="outside extent"):
        SparseSet(10, [11])


@pytest.mark.skipif(_capi.device_count() > 0, reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(lib):
    import numpy as np
    from icebin_amd import linear_Weighted, from_synthetic, synthetic
    with pytest.raises(_capi.IcebinHipError, match="no CPU fallback") as ei:
        linear_Weighted.from_coo((1, 1), [0], [0], [1.0], [1.0], [1.0])
    assert ei.value.code == _capi.IBH_ENODEVICE
    with pytest.raises(_capi.IcebinHipError, match="no CPU fallback"):
        from_synthetic(synthetic.make_grids("tiny"))


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under icebin_amd/ may reference it
    pkg = os.path.join(ROOT, "icebin_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "liboracle" not in text and "icebin_oracle.h" not in text, f


def test_cython_module_builds_and_has_the_reference_surface():
    # pylib/_icebin.pyx:50-175: `import icebin` gives GCMRegridder / RegridMatrices with the reference's
    # method names and keyword defaults; compute fails loudly without a GPU (RuntimeError via `except +`)
    import sys
    from icebin_amd.cython.build_ext import build
    build()
    sys.path.insert(0, os.path.join(ROOT, "icebin_amd", "cython"))
    import icebin
    for cls, names in ((icebin.GCMRegridder, ("nA", "nE", "nhc", "wA", "add_sheet", "regrid_matrices")),
                       (icebin.RegridMatrices, ("matrix",)),
                       (icebin.linear_Weighted, ("apply_M", "apply_weight", "apply_wM", "apply_Mw", "to_coo", "get_weights", "shape"))):
        for n in names:
            assert hasattr(cls, n), (cls, n)
    g = icebin.GCMRegridder(dict(nA=4, to_sparse=[0, 1], native_area=[1., 1.]), [0., 100.], True)
    assert (g.nA, g.nE, g.nhc) == (4, 8, 2)
    with pytest.raises(ValueError, match="Invalid arguments"):
        icebin.GCMRegridder(1, 2)
    with pytest.raises(ValueError, match="snative"):
        g.wA("s", "bogus")
    if _capi.device_count() == 0:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            g.add_sheet("s", dict(nI=2), dict(indices=[[0, 0], [1, 1]], overlaps=[1., 1.]))


def test_missing_rccl_is_enotimpl_not_a_crash(lib):
    # with no librccl to be found, ibh_comm_unique_id must return IBH_ENOTIMPL ("multi-GPU exchange needs RCCL")
    # -- it used to build a std::string from dlerror()'s second, NULL, return.  A fresh process without torch (whose bundled
    # librccl would already be in the process), ICEBIN_RCCL_LIB naming a file that does not exist.
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "L = C.CDLL(%r)\n"
        "L.ibh_last_error.restype = C.c_char_p\n"
        "buf = C.create_string_buffer(128)\n"
        "rc = L.ibh_comm_unique_id(buf)\n"
        "print(rc, L.ibh_last_error().decode())\n" % os.path.join(ROOT, "icebin_amd", "lib", "libicebin_hip.so"))
    env = dict(os.environ, ICEBIN_RCCL_LIB="/nonexistent/librccl.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rc, msg = r.stdout.strip().split(" ", 1)
    assert int(rc) == _capi.IBH_ENOTIMPL and "needs RCCL" in msg and "/nonexistent/librccl.so" in msg, r.stdout
