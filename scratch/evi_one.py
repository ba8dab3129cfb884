"""One kernel choice of the EvI apply, `reps` launches of `depth` applies each (for rocprofv3 passes: kernel stats, PMC).
usage: evi_one.py config mode[rowblock|rowdual|colsweep] depth reps"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
cfg, mode, depth, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
nf = 64
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
icebin_amd.set_tuning("rowdual_auto", 1 if mode == "rowdual" else 0)
icebin_amd.set_tuning("sweep_auto", 1 if mode == "colsweep" else 0)
icebin_amd.set_tuning("rowdual_min_work", 1); icebin_amd.set_tuning("sweep_min_work", 1)
W = rm.matrix("EvI")
nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
L = _capi.lib(); st = torch.cuda.current_stream(); cs = C.c_void_p(st.cuda_stream)
nbuf = max(2, -(-(512 << 20) // (8 * nf * ncol)))
x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda()
X = [x0 + 1e-3 * b for b in range(nbuf)]          # (torch's elementwise add: the PMC calibration kernel, reads 8*nf*ncol bytes)
Y = [torch.zeros((nf, nrow), dtype=torch.float64, device="cuda") for _ in range(depth)]
for i in range(reps + 2):
    xa = (C.c_void_p * depth)(*[X[(i * depth + j) % nbuf].data_ptr() for j in range(depth)])
    ya = (C.c_void_p * depth)(*[y.data_ptr() for y in Y])
    assert L.ibh_weighted_apply_many_device(W._h, depth, xa, nf, ncol, ya, nrow, float("nan"), 0, cs) == 0
torch.cuda.synchronize()
B = 12 * nnz + 4 * (nrow + 1) + 8 * nf * (ncol + nrow)
print("%s EvI %s kernel %s depth %d: %d x %d nnz %d, algorithmic bytes per apply %d, calibration read bytes %d" % (cfg, mode, W.last_kernel(), depth, nrow, ncol, nnz, B, 8 * nf * ncol))
