"""Builds icebin_amd/lib/libicebin_hip.so (hipcc, gfx950 only, in-tree)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libicebin_hip.so")
SOURCES = ["capi.hip", "spmm.hip", "prims.hip", "assemble.hip", "gridgen.hip", "comm.hip"]
HEADERS = ["common.h", "prims.h", "assemble.h", "fastasm.inl", "streamasm.inl", "sweep_kernel.inl", os.path.join("..", "..", "include", "icebin_hip.h")]
# -ffp-contract=off: the bookkeeping kernels must round every multiply and add separately
# (bit-exact weights); the SpMM kernels ask for FMA explicitly with fma().
# -amdgpu-kernarg-preload-count=16: the first 16 dwords of the kernel arguments arrive in SGPRs with the wave instead of
# through a scalar load in front of everything else (gfx950: measured -0.2..-0.4 us on a latency-sized 64-field apply).
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-result",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, variant=None, extra_flags=()):
    """variant/extra_flags: experiment builds (scratch/), e.g. variant="nt", extra_flags=["-DIBH_X_AUX=2"]
    -> lib/libicebin_hip_nt.so, loaded with ICEBIN_HIP_LIB=<path>.  The product build takes neither."""
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    tag = "" if not variant else "_" + variant
    lib = LIB if not variant else os.path.join(LIBDIR, "libicebin_hip%s.so" % tag)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", tag + ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc()] + FLAGS + list(extra_flags) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(lib, objs):
        run([_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    variant = args[0] if args else None
    print(build_library(force="--force" in sys.argv, verbose=True, variant=variant, extra_flags=args[1:]))
