"""Experiment: how much of the SpMM time is due to straddler (foreign-range) gathers?"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
nf = 64
g = syn.make_grids("g5"); em = syn.dome_elevmask(g)
W = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em).matrix("AvI")
rowptr, col, val = W.csr_dense()
nrow, ncol = W.nrow_d, W.ncol_d
row = np.repeat(np.arange(nrow), np.diff(rowptr))
owner = np.full(ncol, nrow, np.int64); np.minimum.at(owner, col, row)
own = owner[col] == row
print("nnz", len(col), "own", own.sum(), "foreign", (~own).sum())
d = (row - owner[col])[~own]
vals_, cnts = np.unique(d, return_counts=True); print("foreign row distance histogram:", dict(zip(vals_.tolist(), cnts.tolist())))
def mk(mask):
    r2 = np.zeros(nrow + 1, np.int32); np.add.at(r2, row[mask] + 1, 1); r2 = np.cumsum(r2).astype(np.int32)
    return icebin_amd.linear_Weighted.from_csr((nrow, ncol), r2, col[mask], val[mask], W.wM, W.Mw)
mats = {"full": W, "own-only": mk(own), "own+left(d==1)": mk(own | ((row - owner[col]) == 1))}
nbuf = 14
x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda(); X = [x0 + 1e-3 * b for b in range(nbuf)]
Y = torch.empty((nf, nrow), dtype=torch.float64, device="cuda")
L = _capi.lib(); fn = L.ibh_weighted_apply_device
cs = C.c_void_p(torch.cuda.current_stream().cuda_stream); xp = [C.c_void_p(x.data_ptr()) for x in X]; yp = C.c_void_p(Y.data_ptr())
def run(w, steps=300):
    for i in range(30): fn(w._h, xp[i % nbuf], nf, ncol, yp, nrow, float("nan"), 0, cs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): fn(w._h, xp[i % nbuf], nf, ncol, yp, nrow, float("nan"), 0, cs)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
for name, w in mats.items():
    print("%-16s nnz=%6d  %.2f us" % (name, w.nnz, run(w)))
