"""Apply time of every matrix shape of a configuration: one launch per apply (depth 1) and batched (depth 16),
cold field batches, us per apply and fraction of the 8 TB/s HBM peak on the algorithmic bytes."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
L = _capi.lib(); st = torch.cuda.current_stream(); cs = C.c_void_p(st.cuda_stream)
for cfg in sys.argv[1].split(","):
    g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
    rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
    for name in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["AvI", "IvA", "EvI", "IvE", "EvA", "AvE"]):
        for nf in (64, 16):
            W = rm.matrix(name)
            nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
            B = 12 * nnz + 4 * (nrow + 1) + 8 * nf * (ncol + nrow)
            nbuf = max(2, min(64, -(-(512 << 20) // max(8 * nf * ncol, 1))))
            ldy = (nrow + 63) // 64 * 64
            x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda()
            X = [x0 + 1e-3 * b for b in range(nbuf)]
            Y = [torch.zeros((nf, ldy), dtype=torch.float64, device="cuda") for _ in range(16)]
            out = []
            for depth in (1, 16):
                steps = 320 if B < 2e8 else 48
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
                out.append("depth %2d %8.2f us %5.1f %%" % (depth, us, B / us / 1e3 / 8000 * 100))
            print("%-4s %-4s nf=%2d %9d x %9d nnz %9d B=%8.1f MB kernel %-8s %s" % (cfg, name, nf, nrow, ncol, nnz, B / 1e6, W.last_kernel(), " | ".join(out)), flush=True)
            del X, Y, W
