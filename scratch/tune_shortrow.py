"""Sweep the shortrow launch parameters (dev tool): fields per thread x gathers in flight x store kind."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
from icebin_amd.linear import set_tuning
cfg = sys.argv[1]; nf = int(sys.argv[2]); names = sys.argv[3].split(",")
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em)
L = _capi.lib(); fn = L.ibh_weighted_apply_device
cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name in names:
    W = rm.matrix(name)
    nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
    B = 12*nnz + 4*(nrow+1) + 8*nf*ncol + 8*nf*nrow
    nb = max(2, min(64, -(-(600 << 20)//(8*nf*max(ncol, nrow)))))
    align = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    ldx, ldy = (ncol + align - 1) // align * align, (nrow + align - 1) // align * align
    X = [torch.randn((nf, ldx), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Y = [torch.empty((nf, ldy), dtype=torch.float64, device="cuda") for _ in range(nb)]
    def run(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps): fn(W._h, C.c_void_p(X[i % nb].data_ptr()), nf, ldx, C.c_void_p(Y[i % nb].data_ptr()), ldy, float("nan"), 0, cs)
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/steps*1e3
    for fper in (4, 8, 16, 32, 64):
        for grp in (4, 8, 16, 32):
            if grp > fper: continue
            for nt in (1, 0):
                set_tuning("shortrow_fper", fper); set_tuning("shortrow_group", grp); set_tuning("shortrow_nt", nt)
                run(5); us = run(60 if nrow > 1000000 else 300)
                print("%-4s %-4s nf=%3d fper=%2d grp=%2d nt=%d %9.2f us  %7.0f GB/s  %5.1f%%" % (cfg, name, nf, fper, grp, nt, us, B/us/1e3, B/us/1e3/80), flush=True)
