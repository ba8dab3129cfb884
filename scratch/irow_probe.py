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
print("   |delta col| between consecutive entries: median %d, 90%% %d, max %d" % (np.median(d), np.percentile(d, 90), d.max()), flush=True)
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
