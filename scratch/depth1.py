"""Kernel duration (HIP events attached to the dispatch) of single applies under ibh_set_tuning overrides.
usage: depth1.py [--depth D] "config:matrix:nvar:key=val,key=val" ...   (one process, many experiments)
Prints median / min kernel us per apply over rotating cold batches, and the fraction of 8 TB/s."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn

args = sys.argv[1:]
depth = 1
if args and args[0] == "--depth":
    depth = int(args[1]); args = args[2:]
L = _capi.lib(); st = torch.cuda.current_stream(); cs = C.c_void_p(st.cuda_stream)
cache = {}
KEYS = set()


def matrix(cfg, name):
    if (cfg, name) not in cache:
        if cfg not in cache:
            g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
            cache[cfg] = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
        cache[(cfg, name)] = cache[cfg].matrix(name)
    return cache[(cfg, name)]


def ev():
    e = C.c_void_p(); _capi.check(L.ibh_event_create(C.byref(e))); return e


for spec in args:
    parts = spec.split(":")
    cfg, name, nf = parts[0], parts[1], int(parts[2])
    tune = dict(kv.split("=") for kv in parts[3].split(",") if kv) if len(parts) > 3 else {}
    tune_all = dict(tune)
    tune = {k: v for k, v in tune.items() if k not in ("nbuf", "raw", "kernel")}
    for k in KEYS - set(tune):
        icebin_amd.set_tuning(k, -2 ** 31)          # back to the built-in default
    for k, v in tune.items():
        icebin_amd.set_tuning(k, int(v)); KEYS.add(k)
    W = matrix(cfg, name)
    W.set_kernel(tune_all.get("kernel", "auto"))
    nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
    B = 12 * nnz + 4 * (nrow + 1) + 8 * nf * (ncol + nrow)
    ldy = (nrow + 63) // 64 * 64 if nrow >= 4096 else nrow
    nbuf = int(tune_all.get("nbuf", 0)) or min(32, max(2, -(-(512 << 20) // (8 * nf * ncol))))
    raw = int(tune_all.get("raw", 0))
    x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda()
    X = [x0 + 1e-3 * b for b in range(nbuf)]
    Y = [torch.zeros((nf, ldy), dtype=torch.float64, device="cuda") for _ in range(depth)]
    W.prepare(nf, depth)
    n = 200 if B < 2e8 else 40
    ya = (C.c_void_p * depth)(*[y.data_ptr() for y in Y])

    def launch(i, pair=None):
        if pair:
            L.ibh_set_launch_events(pair[0], pair[1])
        xa = (C.c_void_p * depth)(*[X[(i * depth + j) % nbuf].data_ptr() for j in range(depth)])
        if depth == 1:
            rc = L.ibh_weighted_apply_device(W._h, xa[0], nf, ncol, ya[0], ldy, float("nan"), 0, cs)
        else:
            rc = L.ibh_weighted_apply_many_device(W._h, depth, xa, nf, ncol, ya, ldy, float("nan"), 0, cs)
        assert rc == 0, L.ibh_last_error()
    for i in range(20):
        launch(i)
    torch.cuda.synchronize()
    pairs = [(ev(), ev()) for _ in range(n)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(n):
        launch(i, pairs[i])
    e1.record(st)
    torch.cuda.synchronize()
    us = []
    for a, b in pairs:
        ms = C.c_float(); _capi.check(L.ibh_event_elapsed_ms(a, b, C.byref(ms))); us.append(ms.value * 1e3 / depth)
        L.ibh_event_destroy(a); L.ibh_event_destroy(b)
    if raw:
        print("   raw (launch i uses batch i %% %d): " % nbuf + " ".join("%.1f" % u for u in us[:raw]))
    us.sort()
    med = us[len(us) // 2]
    print("%-4s %-4s nf=%-3d d=%-2d %-44s kernel %-9s med %7.2f min %7.2f p90 %7.2f us  wall %7.2f  %5.1f%% of 8 TB/s" %
          (cfg, name, nf, depth, parts[3] if len(parts) > 3 else "", W.last_kernel(), med, us[0], us[int(len(us) * 0.9)],
           e0.elapsed_time(e1) * 1e3 / n / depth, B / med / 1e3 / 8000 * 100), flush=True)
    del X, Y, x0
