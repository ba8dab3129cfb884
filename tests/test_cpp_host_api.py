"""Compiles the C++ host-API test against libicebin_hip.so (g++, no HIP headers needed) and runs it."""
import os
import subprocess

import numpy as np
import pytest

from icebin_amd import _capi
from icebin_amd.build import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_api")


def compile_exe(allow_compile=True):
    """The executable is built by __graft_entry__.build() (and by the CPU test below) and travels to the GPU box with the
    snapshot: the GPU test only compiles when it is missing altogether."""
    lib = build_library()
    src = os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp")
    hdrs = [os.path.join(ROOT, "icebin_amd", "host", h) for h in ("icebin_hip.hpp", "ncio.hpp")]
    libdir = os.path.dirname(lib)
    stale = (not os.path.exists(EXE)) or os.path.getmtime(EXE) < max(os.path.getmtime(f) for f in [src, lib] + hdrs)
    if stale and (allow_compile or not os.path.exists(EXE)):
        subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-o", EXE, src, "-L" + libdir, "-licebin_hip",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_cpp_host_api_compiles_and_fails_loudly_without_gpu():
    exe = compile_exe()
    if _capi.device_count() > 0:
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr
    assert "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_cpp_host_api_on_gpu(tmp_path):
    exe = compile_exe(allow_compile=False)
    fn = str(tmp_path / "host_api.nc")
    r = subprocess.run([exe, fn], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
    # the file the C++ host wrote (GCMRegridder_Standard::ncio + two Weighted::ncio appends) through the Python side:
    # the same regridder builds the same matrix as the one stored beside it
    import icebin_amd
    mm = icebin_amd.GCMRegridder(fn)
    assert (mm.nA, mm.nhc) == (32, 40)
    stored = icebin_amd.nc_read_weighted(fn, "AvI")
    nx, ny = 12, 10
    em = np.array([[np.nan if (ix == 0 or iy == ny - 1) else 100.0 * ix + 37.0 * iy for iy in range(ny)] for ix in range(nx)]).reshape(-1)
    w = mm.regrid_matrices("greenland", em, scale=True, correctA=False).matrix("AvI")
    for a, b in zip(w.coo_dense(), stored.coo_dense()):
        assert np.array_equal(a, b)
    assert np.array_equal(w.wM.view(np.uint64), stored.wM.view(np.uint64)) and np.array_equal(w.dim(1), stored.dim(1))
    assert stored.shape == w.shape
