"""`reps` single applies of one matrix with a forced kernel (for rocprofv3 passes).  usage: one.py cfg matrix nf kernel reps; TUNE=key=val,..."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
cfg, name, nf, kern, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]: icebin_amd.set_tuning(k, int(v))
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
W = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True).matrix(name)
W.set_kernel(kern); W.prepare(nf, 1)
nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
L = _capi.lib(); cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
nbuf = max(2, min(32, -(-(600 << 20) // (8 * nf * ncol))))
x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda()
X = [x0 + 1e-3 * b for b in range(nbuf)]          # (torch's elementwise add: the PMC calibration kernel, reads 8*nf*ncol bytes)
Y = torch.zeros((nf, nrow), dtype=torch.float64, device="cuda")
for i in range(reps):
    assert L.ibh_weighted_apply_device(W._h, C.c_void_p(X[i % nbuf].data_ptr()), nf, ncol, C.c_void_p(Y.data_ptr()), nrow, float("nan"), 0, cs) == 0
torch.cuda.synchronize()
B = 12 * nnz + 4 * (nrow + 1) + 8 * nf * (ncol + nrow)
print("%s %s nf %d kernel %s (%s): %d x %d nnz %d, algorithmic bytes per apply %d, calibration read bytes %d" % (cfg, name, nf, W.last_kernel(), os.environ.get("TUNE", ""), nrow, ncol, nnz, B, 8 * nf * ncol))
