"""I-row apply on synthetic matrices of the Antarctic sheet's size: exactly k entries in every row, or the real mix (dev tool)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi
from icebin_amd.linear import linear_Weighted
nrow, ncol = 16336745, 49771
L = _capi.lib(); fn = L.ibh_weighted_apply_device
cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(3)
def mk(counts):
    rowptr = np.zeros(nrow + 1, np.int64); np.cumsum(counts, out=rowptr[1:])
    rr = np.repeat(np.arange(nrow), counts); k = np.arange(rowptr[-1]) - rowptr[rr]
    col = (((rr // 64) * 2 + k) % ncol).astype(np.int32)
    return rowptr.astype(np.int32), col, np.full(len(col), 0.5)
cases = {"1 entry / row": np.ones(nrow, np.int64), "2 entries / row": np.full(nrow, 2, np.int64), "4 entries / row": np.full(nrow, 4, np.int64),
         "2, 7 % of rows 4 (in runs)": np.where((np.arange(nrow) // 40) % 14 == 0, 4, 2), "2, 7 % of rows 4 (scattered)": np.where(rng.random(nrow) < 0.07, 4, 2),
         "2, 0.1 % of rows 8": np.where(rng.random(nrow) < 0.001, 8, 2)}
for cname, counts in cases.items():
    rowptr, col, val = mk(counts)
    W = linear_Weighted.from_csr((nrow, ncol), rowptr, col, val, np.ones(nrow), np.ones(ncol))
    for nf in (16, 128):
        B = 12 * len(val) + 4 * (nrow + 1) + 8 * nf * (ncol + nrow)
        ldy = (nrow + 63) // 64 * 64
        X = [torch.randn((nf, ncol), dtype=torch.float64, device="cuda") for _ in range(2)]
        Y = [torch.empty((nf, ldy), dtype=torch.float64, device="cuda") for _ in range(2)]
        def run(steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(steps): assert fn(W._h, C.c_void_p(X[i % 2].data_ptr()), nf, ncol, C.c_void_p(Y[i % 2].data_ptr()), ldy, float("nan"), 0, cs) == 0
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / steps * 1e3
        run(2); us = run(8)
        print("%-32s nnz %9d nf=%3d %9.1f us %6.0f GB/s %5.1f%%" % (cname, len(val), nf, us, B / us / 1e3, B / us / 1e3 / 80), flush=True)
        del X, Y
    del W; torch.cuda.empty_cache()
