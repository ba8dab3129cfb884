"""In-kernel timeline of the headline SpMM (dev tool): s_memtime stamps per workgroup."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
nf = 64
g = syn.make_grids("g5"); em = syn.dome_elevmask(g)
W = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em).matrix("AvI")
nrow, ncol = W.nrow_d, W.ncol_d
nbuf = 14
x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda(); X = [x0 + 1e-3 * b for b in range(nbuf)]
Y = torch.empty((nf, nrow), dtype=torch.float64, device="cuda")
L = _capi.lib(); fn = L.ibh_weighted_apply_device
cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
diag = torch.zeros(((1 << 20) + 64,), dtype=torch.int64, device="cuda")
for i in range(50): fn(W._h, C.c_void_p(X[i % nbuf].data_ptr()), nf, ncol, C.c_void_p(Y.data_ptr()), nrow, float("nan"), 0, cs)
p = diag.data_ptr()
icebin_amd.set_tuning("rowblock_diag_lo", (p & 0xffffffff) - (1 << 32) if (p & 0x80000000) else (p & 0xffffffff))
icebin_amd.set_tuning("rowblock_diag_hi", p >> 32)
for i in range(5): fn(W._h, C.c_void_p(X[(i + 3) % nbuf].data_ptr()), nf, ncol, C.c_void_p(Y.data_ptr()), nrow, float("nan"), 0, cs)
torch.cuda.synchronize()
nb = nrow * 16
d = diag[: nb * 8].cpu().numpy().reshape(nb, 8).astype(np.float64)
GHZ = 100.0   # s_memrealtime ticks per microsecond (constant 100 MHz)
xcd = np.arange(nb) % 8
names = ["start", "rowptr", "staged", "batches", "consumed", "reduced", "stored"]
print("blocks", nb, "row len min/med/max", d[:, 7].min(), np.median(d[:, 7]), d[:, 7].max())
rel = np.empty_like(d[:, :7])
for x in range(8):
    m = xcd == x
    rel[m] = (d[m, :7] - d[m, 0].min()) / GHZ
for k, nme in enumerate(names):
    c = rel[:, k]
    print("%-9s first %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  last %6.2f us (relative to the first start on the same XCD)" % (nme, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
dur = (d[:, 6] - d[:, 0]) / GHZ
print("block duration: median %.2f p90 %.2f max %.2f us" % (np.median(dur), np.percentile(dur, 90), dur.max()))
for a, b, nme in ((0, 1, "rowptr"), (1, 2, "staging+barrier"), (2, 3, "full batches"), (3, 4, "tail batch"), (4, 5, "reduce"), (5, 6, "store")):
    ph = (d[:, b] - d[:, a]) / GHZ
    print("  phase %-16s median %.2f p90 %.2f us" % (nme, np.median(ph), np.percentile(ph, 90)))
for x in range(8):
    m = xcd == x
    print("  XCD-group %d: blocks %d  starts span %.2f us  last store %.2f us" % (x, m.sum(), rel[m, 0].max(), rel[m, 6].max()))
