"""Where a batched coupler step spends its time beyond its builds: the raw C call against the Python wrapper."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
from icebin_amd.linear import linear_Weighted
cfg = sys.argv[1] if len(sys.argv) > 1 else "g5"
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
mm = icebin_amd.from_synthetic(g)
nI, nX, nE = g["nI"], len(g["ex_area"]), g["nA"] * 40
rm = mm.regrid_matrices("greenland", em, scale=True, correctA=False)
L = _capi.lib()
def sets():
    return icebin_amd.SparseSet.identity(nI), icebin_amd.SparseSet(nE), icebin_amd.SparseSet.identity(nX)
best = {"python": 1e9, "raw": 1e9, "raw+destroy": 1e9}
for step in range(12):
    dimI, dimE, dimX = sets()
    jobs = [("EvI", (dimE, dimI), False, False), ("AvI", (None, dimI), False, False), ("IvE", (dimI, dimE), True, False), ("XvE", (dimX, dimE), False, False)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ws = rm.matrix_batch(jobs)
    t1 = time.perf_counter()
    del ws
    dimI, dimE, dimX = sets()
    jobs = [("EvI", (dimE, dimI), False, False), ("AvI", (None, dimI), False, False), ("IvE", (dimI, dimE), True, False), ("XvE", (dimX, dimE), False, False)]
    n = 4
    names = (C.c_char_p * n)(*[j[0].encode() for j in jobs])
    d0 = (C.c_void_p * n)(*[(j[1][0]._h.value if j[1][0] is not None else None) for j in jobs])
    d1 = (C.c_void_p * n)(*[(j[1][1]._h.value if j[1][1] is not None else None) for j in jobs])
    sc = (C.c_int32 * n)(*[int(j[2]) for j in jobs]); ca = (C.c_int32 * n)(*[int(j[3]) for j in jobs])
    out = (C.c_void_p * n)()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    rc = L.ibh_regrid_matrices_matrix_batch(rm._h, n, names, d0, d1, sc, ca, None, out)
    t3 = time.perf_counter()
    assert rc == 0
    for k in range(n): L.ibh_weighted_destroy(C.c_void_p(out[k]))
    t4 = time.perf_counter()
    if step >= 2:
        best["python"] = min(best["python"], t1 - t0); best["raw"] = min(best["raw"], t3 - t2); best["raw+destroy"] = min(best["raw+destroy"], t4 - t2)
print(cfg, {k: "%.1f us" % (v * 1e6) for k, v in best.items()})
