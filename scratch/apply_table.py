"""Every matrix of a configuration, 64 and 16 fields: kernel microseconds per apply (median of HIP events attached to the
dispatches) one launch per apply and 16 applies per launch, cold rotating field batches, on handles prepared with
ibh_weighted_prepare; fraction of the 8 TB/s HBM peak on the algorithmic bytes.  usage: apply_table.py g5,g1 [AvI,IvA,...]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
L = _capi.lib(); st = torch.cuda.current_stream(); cs = C.c_void_p(st.cuda_stream)
def ev():
    e = C.c_void_p(); _capi.check(L.ibh_event_create(C.byref(e))); return e
print("%-4s %-4s %-5s %21s %10s %9s | %-9s %8s %6s | %-9s %8s %6s" % ("cfg", "M", "nf", "rows x cols", "nnz", "B (MB)", "kernel d1", "us", "%peak", "kernel d16", "us", "%peak"))
for cfg in sys.argv[1].split(","):
    g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
    rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
    for name in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["AvI", "IvA", "EvI", "IvE", "EvA", "AvE"]):
        for nf in (64, 16, 1):
            W = rm.matrix(name)
            nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
            B = 12 * nnz + 4 * (nrow + 1) + 8 * nf * (ncol + nrow)
            nbuf = max(2, min(32, -(-(512 << 20) // max(8 * nf * ncol, 1))))
            ldy = (nrow + 63) // 64 * 64 if nrow >= 4096 else nrow
            x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda()
            X = [x0 + 1e-3 * b for b in range(nbuf)]
            cols = []
            for depth in (1, 16):
                if depth > 1 and 8 * nf * ldy * depth > (6 << 30): depth = 4
                Y = [torch.zeros((nf, ldy), dtype=torch.float64, device="cuda") for _ in range(depth)]
                W.prepare(nf, depth)
                n = (120 if B < 2e8 else 24) // (1 if depth == 1 else 4)
                ya = (C.c_void_p * depth)(*[y.data_ptr() for y in Y])
                def launch(i, pair=None):
                    if pair: L.ibh_set_launch_events(pair[0], pair[1])
                    xa = (C.c_void_p * depth)(*[X[(i * depth + j) % nbuf].data_ptr() for j in range(depth)])
                    rc = L.ibh_weighted_apply_device(W._h, xa[0], nf, ncol, ya[0], ldy, float("nan"), 0, cs) if depth == 1 else \
                         L.ibh_weighted_apply_many_device(W._h, depth, xa, nf, ncol, ya, ldy, float("nan"), 0, cs)
                    assert rc == 0, L.ibh_last_error()
                for i in range(6): launch(i)
                torch.cuda.synchronize()
                pairs = [(ev(), ev()) for _ in range(n)]
                for i in range(n): launch(i, pairs[i])
                torch.cuda.synchronize()
                us = []
                for a, b in pairs:
                    ms = C.c_float(); _capi.check(L.ibh_event_elapsed_ms(a, b, C.byref(ms))); us.append(ms.value * 1e3 / depth)
                    L.ibh_event_destroy(a); L.ibh_event_destroy(b)
                us.sort(); med = us[len(us) // 2]
                cols.append("%-9s %8.2f %5.1f%%" % (W.last_kernel() + ("" if depth in (1, 16) else "/d%d" % depth), med, B / med / 1e3 / 8000 * 100))
                del Y
            print("%-4s %-4s %-5d %9d x %9d %10d %9.1f | %s | %s" % (cfg, name, nf, nrow, ncol, nnz, B / 1e6, cols[0], cols[1]), flush=True)
            del X, W, x0
