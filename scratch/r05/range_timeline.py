"""In-kernel stamps of k_fa_range (library built with -DFA_TIMELINE: python -m icebin_amd.build fatl -DFA_TIMELINE).
usage: range_timeline.py cfg matrix"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import icebin_amd
from icebin_amd import synthetic as syn
cfg, name = sys.argv[1], sys.argv[2]
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
for i in range(5): W = rm.matrix(name)
NB = 1 << 14
dbg = torch.zeros(NB * 8, dtype=torch.int64, device="cuda")
p = dbg.data_ptr()
icebin_amd.set_tuning("gt_dbg_lo", (p & 0xffffffff) - (1 << 32) if (p & 0x80000000) else p & 0xffffffff)
icebin_amd.set_tuning("gt_dbg_hi", p >> 32)
W = rm.matrix(name)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(NB, 8)
d = d[d[:, 0] != 0]
t00 = d[:, 0].min()
print("%s %s: %d ranges stamped; first start .. last end %d ticks" % (cfg, name, len(d), d[:, 7].max() - t00))
names = ["start", "prologue done", "3 load rounds landed", "groups formed", "entries placed (all passes)", "straddlers ranked + barrier", "segments summed", "scaled, end"]
def q(a):
    a = np.sort(a); return "med %7d  p10 %7d  p90 %7d  max %7d" % (a[len(a) // 2], a[len(a) // 10], a[len(a) * 9 // 10], a[-1])
print("start offset: " + q(d[:, 0] - t00))
for i in range(1, 8):
    print("%-30s since previous: %s" % (names[i], q(d[:, i] - d[:, i - 1])))
print("whole workgroup: " + q(d[:, 7] - d[:, 0]))
