"""The coupler's IvE {identity dimI, dimE as EvI left it} a few times (for a rocprofv3 --kernel-trace run)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, icebin_amd
from icebin_amd import synthetic as syn
cfg = sys.argv[1]
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=False)
nI, nX, nE = g["nI"], len(g["ex_area"]), g["nA"] * 40
for i in range(8):
    dimI, dimE = icebin_amd.SparseSet.identity(nI), icebin_amd.SparseSet(nE)
    EvI = rm.matrix_d("EvI", (dimE, dimI), scale=False, correctA=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    IvE = rm.matrix_d("IvE", (dimI, dimE), scale=True, correctA=False)
    t1 = time.perf_counter()
print("coupler IvE: %.1f us" % ((t1 - t0) * 1e6))
