import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, time
import icebin_amd
from icebin_amd import synthetic as syn
# tile path against the triplet pipeline at 5 km (the oracle is too slow here): same structure, entries to rounding
g = syn.make_grids("g5"); em = syn.dome_elevmask(g)
mm = icebin_amd.from_synthetic(g)
for sig in ((50e3, 50e3, 100.), (25e3, 25e3, 100.)):
    for name in ("IvA", "IvE"):
        res = {}
        for tile in (1, 0):
            icebin_amd.set_tuning("smooth_tile", tile)
            rm = mm.regrid_matrices("greenland", em, scale=True, correctA=True, sigma=sig)
            w = rm.matrix(name); torch.cuda.synchronize()
            t0 = time.perf_counter(); w = rm.matrix(name); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            res[tile] = (w.coo_dense(), dt)
        (r1, c1, v1), t1 = res[1]; (r0, c0, v0), t0_ = res[0]
        same = np.array_equal(r1, r0) and np.array_equal(c1, c0)
        err = np.max(np.abs(v1 - v0) / np.abs(v0)) if same else -1
        print("g5 %s sigma=%s nnz %d: tile %.2f ms, triplets %.2f ms; same structure %s, max rel diff %.2e" % (name, sig, len(v1), t1 * 1e3, t0_ * 1e3, same, err), flush=True)
