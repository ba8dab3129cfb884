"""Sweep SpMM launch variants on the headline shape (dev tool, not part of the product)."""
import ctypes as C, sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn

cfg = sys.argv[1] if len(sys.argv) > 1 else "g5"
name = sys.argv[2] if len(sys.argv) > 2 else "AvI"
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 64
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
mm = icebin_amd.from_synthetic(g)
W = mm.regrid_matrices("greenland", em).matrix(name)
nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
B = 12*nnz + 4*(nrow+1) + 8*nf*ncol + 8*nf*nrow
nbuf = max(2, -(-(512 << 20)//(8*nf*ncol)))
x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda()
X = [x0 + 1e-3*b for b in range(nbuf)]
Y = torch.empty((nf, nrow), dtype=torch.float64, device="cuda")
L = _capi.lib(); fn = L.ibh_weighted_apply_device
st = torch.cuda.current_stream(); cs = C.c_void_p(st.cuda_stream)
xp = [C.c_void_p(x.data_ptr()) for x in X]; yp = C.c_void_p(Y.data_ptr())
print("shape", nrow, ncol, nnz, "nf", nf, "bytes", B, "nbuf", nbuf)
def run(steps=300):
    for i in range(30): fn(W._h, xp[i % nbuf], nf, ncol, yp, nrow, float("nan"), 0, cs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): fn(W._h, xp[i % nbuf], nf, ncol, yp, nrow, float("nan"), 0, cs)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/steps*1e3
variants = eval(sys.argv[4]) if len(sys.argv) > 4 else None
if variants is None:
    variants = [dict(rowblock_fpw=f, rowblock_wk=k, rowblock_unroll=u) for f,k,u in
                [(4,1,4),(4,1,2),(4,1,8),(2,1,4),(2,1,8),(8,1,2),(8,1,4),(1,1,8),(4,2,4),(2,2,4)]]
for v in variants:
    for k, val in v.items(): icebin_amd.set_tuning(k, val)
    us = run()
    print(v, "%.2f us  %.0f GB/s  %.1f%%" % (us, B/us/1e3, B/us/1e3/80))
