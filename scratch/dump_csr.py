"""Dump the CSR structure of a matrix for scratch/chain_bench: int32 rowptr[nrow+1] then colind[nnz].  usage: dump_csr.py config matrix out"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icebin_amd
from icebin_amd import synthetic as syn
cfg, name, out = sys.argv[1:4]
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
W = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True).matrix(name)
rowptr, col, val = W.csr_dense()
with open(out, "wb") as f:
    f.write(rowptr.astype(np.int32).tobytes()); f.write(col.astype(np.int32).tobytes())
print("rows", W.nrow_d, "cols", W.ncol_d, "nnz", W.nnz, "row lens", " ".join(str(int(v)) for v in np.diff(rowptr)[:12]), "...")
