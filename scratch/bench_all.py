"""Per-matrix apply timings for a config (dev tool): cold, events, auto kernel."""
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
for kv in os.environ.get("IBH_TUNE", "").split(","):       # e.g. IBH_TUNE=shortrow_xt=1,shortrow_fper=8
    if kv:
        from icebin_amd.linear import set_tuning
        set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
for name in names:
    W = rm.matrix(name)
    nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
    B = 12*nnz + 4*(nrow+1) + 8*nf*ncol + 8*nf*nrow
    nb = max(2, min(64, -(-(600 << 20)//(8*nf*max(ncol, nrow)))))
    X = [torch.randn((nf, ncol), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Y = [torch.empty((nf, nrow), dtype=torch.float64, device="cuda") for _ in range(nb)]
    def run(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps): fn(W._h, C.c_void_p(X[i % nb].data_ptr()), nf, ncol, C.c_void_p(Y[i % nb].data_ptr()), nrow, float("nan"), 0, cs)
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/steps*1e3
    run(20); us = run(200)
    print("%-4s %-4s nf=%3d %8d x %8d nnz=%9d %-8s %9.2f us  %7.0f GB/s  %5.1f%%" % (cfg, name, nf, nrow, ncol, nnz, W.last_kernel(), us, B/us/1e3, B/us/1e3/80), flush=True)
