"""Wave-level stamps of k_fa_pelem<EMIT> (library built with -DFA_TIMELINE). usage: pelem_timeline.py cfg matrix"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import icebin_amd
from icebin_amd import synthetic as syn
cfg, name = sys.argv[1], sys.argv[2]
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
for i in range(5): W = rm.matrix(name)
NB = 1 << 15
dbg = torch.zeros(NB * 8, dtype=torch.int64, device="cuda")
p = dbg.data_ptr()
icebin_amd.set_tuning("gt_dbg_lo", (p & 0xffffffff) - (1 << 32) if (p & 0x80000000) else p & 0xffffffff)
icebin_amd.set_tuning("gt_dbg_hi", p >> 32)
W = rm.matrix(name)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(NB, 8)
d = d[(d[:, 0] != 0) & (d[:, 3] != 0)]
def q(a):
    a = np.sort(a); return "med %7d  p10 %7d  p90 %7d  max %7d" % (a[len(a) // 2], a[len(a) // 10], a[len(a) * 9 // 10], a[-1])
print("%s %s: %d waves stamped (all four stamps)" % (cfg, name, len(d)))
for i, nm in ((1, "prologue (mask, first-seen, dense id, list bounds)"), (2, "first visit of the cells"), (3, "row written")):
    print("%-52s since previous: %s" % (nm, q(d[:, i] - d[:, i - 1])))
print("whole wave: " + q(d[:, 3] - d[:, 0]))
