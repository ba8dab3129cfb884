"""Which apply kernel for a (config, matrix, fields) shape: one apply per launch, kernel forced in turn (dev tool).
usage: kernel_choice.py config matrix nf[,nf...] [kernel,kernel,...]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
cfg, name = sys.argv[1], sys.argv[2]
for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]: icebin_amd.set_tuning(k, int(v))
nfs = [int(v) for v in sys.argv[3].split(",")]
kernels = (sys.argv[4] if len(sys.argv) > 4 else "auto,rowblock,rowdual,colsweep,rowgroup").split(",")
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
L = _capi.lib(); fn = L.ibh_weighted_apply_device
cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for nf in nfs:
    W = rm.matrix(name)
    nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
    B = 12 * nnz + 4 * (nrow + 1) + 8 * nf * ncol + 8 * nf * nrow
    nb = max(2, min(32, -(-(600 << 20) // (8 * nf * max(ncol, nrow)))))
    ldy = (nrow + 63) // 64 * 64
    X = [torch.randn((nf, ncol), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Y = [torch.empty((nf, ldy), dtype=torch.float64, device="cuda") for _ in range(nb)]
    def run(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            rc = fn(W._h, C.c_void_p(X[i % nb].data_ptr()), nf, ncol, C.c_void_p(Y[i % nb].data_ptr()), ldy, float("nan"), 0, cs)
            assert rc == 0, _capi.lib().ibh_last_error()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / steps * 1e3
    for k in kernels:
        try:
            W.set_kernel(k)
            W.prepare(nf, 1)
            run(3); us = run(12 if B > 2e9 else 60)
            print("%-4s %-4s nf=%3d forced=%-9s ran=%-9s %10.2f us  %7.0f GB/s  %5.1f%%" % (cfg, name, nf, k, W.last_kernel(), us, B / us / 1e3, B / us / 1e3 / 80), flush=True)
        except Exception as e:      # noqa: BLE001
            print("%-4s %-4s nf=%3d forced=%-9s failed: %s" % (cfg, name, nf, k, str(e).splitlines()[0]), flush=True)
    del X, Y, W
    torch.cuda.empty_cache()
