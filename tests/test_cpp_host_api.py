"""Compiles the C++ host-API test against libicebin_hip.so (g++, no HIP headers needed) and runs it."""
import os
import subprocess

import pytest

from icebin_amd import _capi
from icebin_amd.build import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_api")


def compile_exe():
    lib = build_library()
    src = os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp")
    libdir = os.path.dirname(lib)
    if (not os.path.exists(EXE)) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(lib)):
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
def test_cpp_host_api_on_gpu():
    exe = compile_exe()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
