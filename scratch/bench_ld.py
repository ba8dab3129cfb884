"""Apply timings vs leading-dimension padding of the field batches (dev tool)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
cfg = sys.argv[1]; nf = int(sys.argv[2]); names = sys.argv[3].split(",")
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em)
L = _capi.lib(); fn = L.ibh_weighted_apply_device
cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def pad(n, a): return (n + a - 1) // a * a
for name in names:
    W = rm.matrix(name)
    nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
    B = 12*nnz + 4*(nrow+1) + 8*nf*ncol + 8*nf*nrow
    for align in (1, 8, 64, 512):
        ldx, ldy = pad(ncol, align), pad(nrow, align)
        nb = max(2, min(64, -(-(600 << 20)//(8*nf*max(ldx, ldy)))))
        X = [torch.randn((nf, ldx), dtype=torch.float64, device="cuda") for _ in range(nb)]
        Y = [torch.empty((nf, ldy), dtype=torch.float64, device="cuda") for _ in range(nb)]
        def run(steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(steps): fn(W._h, C.c_void_p(X[i % nb].data_ptr()), nf, ldx, C.c_void_p(Y[i % nb].data_ptr()), ldy, float("nan"), 0, cs)
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/steps*1e3
        run(20); us = run(60 if max(nrow, ncol) > 1000000 else 400)
        print("%-4s %-4s nf=%3d align=%3d ldx=%8d ldy=%8d %-8s %9.2f us  %7.0f GB/s  %5.1f%%" % (cfg, name, nf, align, ldx, ldy, W.last_kernel(), us, B/us/1e3, B/us/1e3/80), flush=True)
        del X, Y
