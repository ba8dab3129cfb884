"""Only batched coupler-step builds (for kernel traces): prints wall time per batched call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import synthetic as syn
cfg = sys.argv[1]
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
mm = icebin_amd.from_synthetic(g)
nI, nX, nE = g["nI"], len(g["ex_area"]), g["nA"] * 40
rm = mm.regrid_matrices("greenland", em, scale=True, correctA=False)
ts = []
for step in range(12):
    dimI, dimE, dimX = icebin_amd.SparseSet.identity(nI), icebin_amd.SparseSet(nE), icebin_amd.SparseSet.identity(nX)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ws = rm.matrix_batch([("EvI", (dimE, dimI), False, False), ("AvI", (None, dimI), False, False),
                          ("IvE", (dimI, dimE), True, False), ("XvE", (dimX, dimE), False, False)])
    ts.append((time.perf_counter() - t0) * 1e3)
print("batched call: best %.3f ms, median %.3f ms" % (min(ts[2:]), sorted(ts[2:])[len(ts[2:]) // 2]))
