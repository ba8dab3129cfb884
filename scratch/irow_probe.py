"""Why is the I-row apply slower on the Antarctic sheet than on the 1 km Greenland sheet?  The real IvE / IvA against copies of it
with the column pattern replaced (all columns 0 / columns sorted into long runs / random), 16 and 128 fields, one launch (dev tool)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
from icebin_amd.linear import linear_Weighted
cfg, name = sys.argv[1], sys.argv[2]
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
W0 = rm.matrix(name)
rowptr, colind, val = W0.csr_dense()

nrow, ncol = W0.nrow_d, W0.ncol_d
wM = np.ones(nrow); Mw = np.ones(ncol)
rowptr = np.asarray(rowptr); colind = np.asarray(colind); val = np.asarray(val)
print(cfg, name, "nrow", nrow, "ncol", ncol, "nnz", len(val), "entries/row", len(val) / nrow, "distinct col changes along the entries: %.3f per entry" % (np.count_nonzero(np.diff(colind)) / len(colind)), flush=True)
d = np.abs(np.diff(colind.astype(np.int64)))
cnt = np.diff(rowptr); print("   row lengths:", np.bincount(cnt)[:12], "waves (64 rows) with a row > 4:", float(np.mean(np.maximum.reduceat(cnt, np.arange(0, len(cnt), 64)) > 4)), flush=True)
print("   |delta col| between consecutive entries: median %d, 90%% %d, max %d" % (np.median(d), np.percentile(d, 90), d.max()), flush=True)
# per wave of 64 rows and entry slot: how many rows name the column the kernel would share (first / last row's, the more frequent)
nw = nrow // 64
for e in range(4):
    has = cnt[:nw * 64] > e
    ce = np.where(has, colind[np.minimum(rowptr[:nw * 64] + e, len(colind) - 1)], -1).reshape(nw, 64)
    first = ce[:, 0]; last = ce[:, -1]
    ma = (ce == first[:, None]).sum(1); mb = (ce == last[:, None]).sum(1)
    best = np.maximum(ma, mb)
    A = np.where(ma >= mb, first, last)        # (first / last lane having the entry would be exact; lanes 0 / 63 as a stand-in)
    restm = (ce != A[:, None]) & (ce >= 0)
    idx = restm.argmax(1); Bc = np.where(restm.any(1), ce[np.arange(nw), idx], -2)
    left = restm & (ce != Bc[:, None])
    print("   entry %d: waves that have it %.3f; with lanes outside A %.3f; outside A and B %.3f (mean %.1f lanes there)" % (e, float(np.mean((ce >= 0).any(1))), float(np.mean(restm.any(1))), float(np.mean(left.any(1))), float(left.sum(1)[left.any(1)].mean()) if left.any() else 0.0), flush=True)
    print("   entry %d: rows per wave in the shared column: mean %.1f of 64; waves with >= 56: %.3f; distinct columns per wave: mean %.2f" % (e, best.mean(), float(np.mean(best >= 56)), float(np.mean([len(np.unique(r)) for r in ce[::997]]))), flush=True)
print("   first rows, (row: cols):", [(int(r), colind[rowptr[r]:rowptr[r + 1]].tolist()) for r in range(100000, 100012)], flush=True)
L = _capi.lib(); fn = L.ibh_weighted_apply_device
cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
variants = {"real": colind, "all-zero": np.zeros_like(colind), "row/64 blocks": ((np.repeat(np.arange(nrow), np.diff(rowptr)) // 64) % ncol).astype(np.int32),
            "pairs c,c+1 by row/64": None, "random": rng.integers(0, ncol, len(colind)).astype(np.int32)}
rr = np.repeat(np.arange(nrow), np.diff(rowptr)); k_in_row = np.arange(len(colind)) - rowptr[rr]
variants["pairs c,c+1 by row/64"] = (((rr // 64) * 2 + k_in_row) % ncol).astype(np.int32)
for vname, ci in variants.items():
    W = linear_Weighted.from_csr((nrow, ncol), rowptr, np.ascontiguousarray(ci, np.int32), val, wM, Mw)
    for nf in (16, 128):
        B = 12 * len(val) + 4 * (nrow + 1) + 8 * nf * (ncol + nrow)
        nb = 2
        ldy = (nrow + 63) // 64 * 64
        X = [torch.randn((nf, ncol), dtype=torch.float64, device="cuda") for _ in range(nb)]
        Y = [torch.empty((nf, ldy), dtype=torch.float64, device="cuda") for _ in range(nb)]
        def run(steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(steps): assert fn(W._h, C.c_void_p(X[i % nb].data_ptr()), nf, ncol, C.c_void_p(Y[i % nb].data_ptr()), ldy, float("nan"), 0, cs) == 0
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / steps * 1e3
        run(2); us = run(8)
        print("   %-24s nf=%3d %9.1f us %6.0f GB/s %5.1f%%  kernel %s" % (vname, nf, us, B / us / 1e3, B / us / 1e3 / 80, W.last_kernel()), flush=True)
        del X, Y
    del W; torch.cuda.empty_cache()
