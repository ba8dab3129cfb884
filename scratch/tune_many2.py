"""Short sweep of the batched launch for a given library build (ICEBIN_HIP_LIB): a few settings at depth 8/16."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
cfg = sys.argv[1] if len(sys.argv) > 1 else "g5"
name = sys.argv[2] if len(sys.argv) > 2 else "AvI"
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 64
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
W = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True).matrix(name)
nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
B = 12 * nnz + 4 * (nrow + 1) + 8 * nf * (ncol + nrow)
nbuf = max(2, -(-(512 << 20) // (8 * nf * ncol)))
x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda()
X = [x0 + 1e-3 * b for b in range(nbuf)]
Y = [torch.zeros((nf, nrow), dtype=torch.float64, device="cuda") for _ in range(32)]
L = _capi.lib(); st = torch.cuda.current_stream(); cs = C.c_void_p(st.cuda_stream)
def timeit(depth, steps=960):
    plans = []
    i = 0
    while i < steps:
        m = min(depth, steps - i)
        xa = (C.c_void_p * m)(*[X[(i + j) % nbuf].data_ptr() for j in range(m)])
        ya = (C.c_void_p * m)(*[Y[j].data_ptr() for j in range(m)])
        plans.append((m, xa, ya)); i += m
    def run():
        for m, xa, ya in plans:
            rc = L.ibh_weighted_apply_many_device(W._h, m, xa, nf, ncol, ya, nrow, float("nan"), 0, cs)
            assert rc == 0, _capi.lib().ibh_last_error()
    run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); run(); e1.record(st); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps * 1e3)
    return best
print("lib %s shape %d x %d nnz %d  B=%.1f MB" % (os.environ.get("ICEBIN_HIP_LIB", "default"), nrow, ncol, nnz, B / 1e6))
for waves, fpw, unroll, qi in [(4, 1, 8, 2), (4, 1, 4, 2), (4, 2, 4, 1), (4, 2, 8, 2), (8, 1, 8, 2), (8, 1, 4, 2), (8, 2, 4, 2), (4, 1, 8, 1), (4, 0, 0, 0)]:
    icebin_amd.set_tuning("rowblock_waves", waves)
    icebin_amd.set_tuning("rowblock_many_fpw", fpw); icebin_amd.set_tuning("rowblock_fpw", fpw)
    icebin_amd.set_tuning("rowblock_wk", 1 if fpw else 0)
    icebin_amd.set_tuning("rowblock_unroll", unroll); icebin_amd.set_tuning("rowblock_many_qi", qi)
    out = []
    for depth in (1, 4, 8, 16):
        us = timeit(depth)
        out.append("d%-2d %6.2f us %4.1f%%" % (depth, us, B / us / 1e3 / 8000 * 100))
    print("waves %d fpw %d unroll %d qi %d : %s" % (waves, fpw, unroll, qi, " | ".join(out)), flush=True)
