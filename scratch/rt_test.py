import sys, os
sys.path.insert(0, '.')
order = sys.argv[1]
import numpy as np
if order == "torch_first":
    import torch
    print("torch cuda:", torch.cuda.is_available())
import icebin_amd
from icebin_amd import synthetic as syn
g = syn.make_grids("g20"); em = syn.dome_elevmask(g)
mm = icebin_amd.from_synthetic(g)
w = mm.regrid_matrices("greenland", em).matrix("AvI")
print("nnz", w.nnz)
import torch
print("torch cuda:", torch.cuda.is_available(), torch.cuda.device_count())
x = torch.ones((4, w.ncol_d), dtype=torch.float64, device="cuda")
y = w.apply_device(x, force_conservation=False); torch.cuda.synchronize()
print(y[0,:3])
os.system("grep -E 'libamdhip64|libhsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
