"""In-kernel stamps of spmm_grouptile_kernel (library built with -DGT_TIMELINE: scratch/r05).
usage: timeline.py cfg matrix nf [key=val,...]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
cfg, name, nf = sys.argv[1], sys.argv[2], int(sys.argv[3])
tune = dict(kv.split("=") for kv in sys.argv[4].split(",") if kv) if len(sys.argv) > 4 else {}
for k, v in tune.items():
    icebin_amd.set_tuning(k, int(v))
L = _capi.lib(); st = torch.cuda.current_stream(); cs = C.c_void_p(st.cuda_stream)
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
W = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True).matrix(name)
W.set_kernel("rowgroup")
nrow, ncol = W.nrow_d, W.ncol_d
nbuf = min(32, max(2, -(-(512 << 20) // (8 * nf * ncol))))
x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda()
X = [x0 + 1e-3 * b for b in range(nbuf)]
Y = torch.zeros((nf, nrow), dtype=torch.float64, device="cuda")
W.prepare(nf, 1)
NB = 1 << 16
dbg = torch.zeros(NB * 16, dtype=torch.int64, device="cuda")
for i in range(10):
    assert L.ibh_weighted_apply_device(W._h, X[i % nbuf].data_ptr(), nf, ncol, Y.data_ptr(), nrow, float("nan"), 0, cs) == 0
torch.cuda.synchronize()
p = dbg.data_ptr()
icebin_amd.set_tuning("gt_dbg_lo", (p & 0xffffffff) - (1 << 32) if (p & 0x80000000) else p & 0xffffffff)
icebin_amd.set_tuning("gt_dbg_hi", p >> 32)
assert L.ibh_weighted_apply_device(W._h, X[10 % nbuf].data_ptr(), nf, ncol, Y.data_ptr(), nrow, float("nan"), 0, cs) == 0
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(NB, 16)
d = d[d[:, 0] != 0]
print("%s %s nf=%d %s: %d workgroups stamped, kernel %s" % (cfg, name, nf, tune, len(d), W.last_kernel()))
t00 = d[:, 0].min()
span = d[:, 12].max() - t00
print("span first start .. last end: %d ticks" % span)
names = ["start", "hdr", "x0 landed(top)", "stored0", "barrier0", "proc0 done", "top1", "stored1", "barrier1", "proc1 done", "loop done", "barrier E", "end"]
def q(a):
    a = np.sort(a); return "med %7d  p10 %7d  p90 %7d" % (a[len(a) // 2], a[len(a) // 10], a[len(a) * 9 // 10])
print("start offset from first start: " + q(d[:, 0] - t00))
prev = 0
for i in range(1, 13):
    m = d[:, i] != 0
    if i in (6, 7, 8, 9):
        m &= d[:, 6] != 0
    if not m.any():
        continue
    base = 5 if i == 6 else (9 if (i == 10) else i - 1)
    if i == 10:
        mm = m & (d[:, 9] != 0); 
        print("%-16s since proc1 done (groups of >2 tiles): %s" % (names[i], q(d[mm, 10] - d[mm, 9])) if mm.any() else "")
        mm = m & (d[:, 9] == 0) & (d[:, 5] != 0)
        if mm.any(): print("%-16s since proc0 done (1-tile groups): %s" % (names[i], q(d[mm, 10] - d[mm, 5])))
        continue
    print("%-16s since %-16s: %s   (n=%d)" % (names[i], names[base], q(d[m, i] - d[m, base]), m.sum()))
m = d[:, 13] != 0
print("tile 0, wave 0: loads of the next tile issued since barrier0: %s" % q(d[m, 13] - d[m, 4]))
print("tile 0, wave 0: unit boundaries since loads issued:           %s" % q(d[m, 14] - d[m, 13]))
print("tile 0, wave 0: units summed since boundaries:                %s" % q(d[m, 5] - d[m, 14]))
print("tile 0, wave 0: steps of the run %s ; steps of unit 0 %s" % (q(d[m, 15] >> 32), q(d[m, 15] & 0xffffffff)))
print("whole workgroup: " + q(d[:, 12] - d[:, 0]))
