"""Shortrow launch parameters on the big sheets (dev tool): usage tune_shortrow2.py config matrix nf[,nf]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
from icebin_amd.linear import set_tuning
cfg, name = sys.argv[1], sys.argv[2]
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em)
L = _capi.lib(); fn = L.ibh_weighted_apply_device
cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
W = rm.matrix(name)
nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
for nf in [int(v) for v in sys.argv[3].split(",")]:
    B = 12*nnz + 4*(nrow+1) + 8*nf*ncol + 8*nf*nrow
    nb = max(2, min(16, -(-(600 << 20)//(8*nf*max(ncol, nrow)))))
    ldy = (nrow + 63) // 64 * 64
    X = [torch.randn((nf, ncol), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Y = [torch.empty((nf, ldy), dtype=torch.float64, device="cuda") for _ in range(nb)]
    def run(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps): assert fn(W._h, C.c_void_p(X[i % nb].data_ptr()), nf, ncol, C.c_void_p(Y[i % nb].data_ptr()), ldy, float("nan"), 0, cs) == 0
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/steps*1e3
    for fper, grp, xt in [(-1, -1, -1), (16, 4, 1), (16, 8, 1), (16, 16, 1), (32, 4, 1), (32, 8, 1), (32, 16, 1), (64, 8, 1), (64, 16, 1), (128, 8, 1), (128, 16, 1), (32, 4, 0), (16, 8, 0)]:
        if fper > max(nf, 16): continue
        set_tuning("shortrow_fper", fper if fper > 0 else -2**31); set_tuning("shortrow_group", grp if grp > 0 else -2**31); set_tuning("shortrow_xt", xt if xt >= 0 else -2**31)
        run(2); us = run(10)
        print("%-4s %-4s nf=%3d fper=%3d grp=%2d xt=%2d %9.2f us  %7.0f GB/s  %5.1f%%" % (cfg, name, nf, fper, grp, xt, us, B/us/1e3, B/us/1e3/80), flush=True)
    del X, Y; torch.cuda.empty_cache()
