"""Builds the Cython module _icebin in-tree (cythonize + g++; links libicebin_hip.so by rpath)."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(os.path.dirname(HERE), "lib")


def ext_path():
    return os.path.join(HERE, "_icebin" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False, verbose=False):
    from icebin_amd.build import build_library
    build_library()
    pyx, pxd = os.path.join(HERE, "_icebin.pyx"), os.path.join(HERE, "cicebin.pxd")
    hpp = os.path.join(os.path.dirname(HERE), "host", "icebin_hip.hpp")
    cpp, out = os.path.join(HERE, "_icebin.cpp"), ext_path()
    deps = [pyx, pxd, hpp, os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "icebin_hip.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=HERE)
        if r.returncode != 0:
            raise RuntimeError("%s failed:\n%s\n%s" % (cmd[0], r.stdout, r.stderr))
    run([sys.executable, "-m", "cython", "--cplus", "-3", "-I", HERE, pyx, "-o", cpp])
    inc = sysconfig.get_paths()["include"]
    run(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-Wno-deprecated-declarations", "-I" + inc, "-I" + HERE, cpp, "-o", out,
         "-L" + LIBDIR, "-licebin_hip", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib"])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
