"""One matrix, 64 fields: us per apply at queue depths 1 / 16 under ibh_set_tuning overrides.
usage: TUNE=key=val,key=val one_matrix.py config matrix"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
cfg, name, nf = sys.argv[1], sys.argv[2], int(os.environ.get("NF", "64"))
for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]: icebin_amd.set_tuning(k, int(v))
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
W = rm.matrix(name)
nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
L = _capi.lib(); st = torch.cuda.current_stream(); cs = C.c_void_p(st.cuda_stream)
B = 12 * nnz + 4 * (nrow + 1) + 8 * nf * (ncol + nrow)
ldy = (nrow + 63) // 64 * 64 if nrow >= 4096 else nrow
nbuf = min(64, max(2, -(-(512 << 20) // (8 * nf * ncol))))
x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda()
X = [x0 + 1e-3 * b for b in range(nbuf)]
Y = [torch.zeros((nf, ldy), dtype=torch.float64, device="cuda") for _ in range(16)]
out = []
for depth in (1, 16):
    steps = 320 if B < 2e8 else 64
    plans = []
    for i in range(0, steps, depth):
        m = min(depth, steps - i)
        plans.append((m, (C.c_void_p * m)(*[X[(i + j) % nbuf].data_ptr() for j in range(m)]), (C.c_void_p * m)(*[Y[j].data_ptr() for j in range(m)])))
    def run():
        for m, xa, ya in plans:
            assert L.ibh_weighted_apply_many_device(W._h, m, xa, nf, ncol, ya, ldy, float("nan"), 0, cs) == 0
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st); run(); e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    out.append("d%-2d %7.2f us %4.1f%%" % (depth, us, B / us / 1e3 / 8000 * 100))
print("%s %s TUNE=%s kernel %s: %s" % (cfg, name, os.environ.get("TUNE", ""), W.last_kernel(), " | ".join(out)), flush=True)
