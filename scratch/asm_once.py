import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import icebin_amd
from icebin_amd import synthetic as syn
cfg, name = sys.argv[1], sys.argv[2]
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
for _ in range(3):
    w = rm.matrix(name); torch.cuda.synchronize()
print(w.nnz)
