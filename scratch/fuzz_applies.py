"""Soak run of the apply kernels: matrices of random exchange grids (sorted and not), random field counts, every kernel the
matrix admits (auto, rowblock, shortrow, rowgroup, colsweep), one launch and batched, NaN fields, fill -- against the oracle's
apply to 1e-12; batched launches bitwise the single ones.  usage: fuzz_applies.py [first_seed] [count]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import icebin_amd
import test_gpu_parity as T
from icebin_amd import synthetic as syn
orc = T.orc
for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]: icebin_amd.set_tuning(k, int(v))      # e.g. shortrow_xt=1,shortrow_group=4

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = n = 0
import collections
seen = collections.Counter()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    g, em = T._random_grid(seed, force_sorted=bool(seed % 2))
    mm, rg = icebin_amd.from_synthetic(g), orc.Regridder(g)
    rm = mm.regrid_matrices("greenland", em, scale=True, correctA=bool(seed % 3))
    for name in ("AvI", "IvA", "EvI", "IvE", "EvA", "AvE", "EvX", "XvE"):
        w = rm.matrix(name)
        o = rg.matrix_d(name, em, scale=True, correctA=bool(seed % 3))
        if w.nnz == 0:
            continue
        nvar = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 33, 64, 70]))
        x = syn.fields(nvar, w.ncol_d, seed=seed)
        if nvar > 1:
            x[nvar // 2, ::5] = np.nan
        ref = o.apply(x, fill=-7.0, force_conservation=False)
        kernels = ["auto", "rowblock", "shortrow"] + (["rowgroup", "colsweep"] if name in ("EvI", "EvX") else [])
        for k in kernels:
            try:
                w.set_kernel(k)
                y = w.apply(x, fill=-7.0, force_conservation=False)
                n += 1
                seen[w.last_kernel() + " " + w.last_launch().split("<")[0]] += 1
                if T.rel_linf(y, ref) > T.FIELD_RTOL:
                    bad += 1; print("MISMATCH", seed, name, k, nvar, T.rel_linf(y, ref), flush=True)
                dx = torch.from_numpy(x).cuda()
                one = w.apply_device(dx, fill=-7.0, force_conservation=False).clone()
                many = w.apply_many_device([dx, dx, dx], fill=-7.0, force_conservation=False)
                torch.cuda.synchronize()
                if not (torch.equal(many[2].view(torch.int64), one.view(torch.int64))):
                    bad += 1; print("BATCH != SINGLE", seed, name, k, nvar, flush=True)
            except icebin_amd.IcebinHipError as e:
                print("refused", seed, name, k, str(e)[:80], flush=True)
        w.set_kernel("auto")
    # the fused pair EvI -> AvE where the two matrices pair (own dims: paired through sparse indices)
    EvI, AvE = rm.matrix("EvI"), rm.matrix("AvE")
    if EvI.nnz and AvE.nnz:
        try:
            EvI.pair_prepare(AvE, 8)
            xx = torch.from_numpy(syn.fields(5, EvI.ncol_d, seed=seed + 1)).cuda()
            xx[2, ::3] = float("nan")
            b1, b2 = EvI.apply_pair_device(AvE, xx, fill=-7.0)
            EvI.set_kernel("rowgroup")
            e = EvI.apply_device(xx, fill=-7.0, force_conservation=False)
            perm = np.argsort(EvI.dim(0))[np.searchsorted(np.sort(EvI.dim(0)), AvE.dim(1))]
            a = AvE.apply_device(e[:, torch.from_numpy(perm).cuda()].contiguous(), fill=-7.0, force_conservation=False)
            torch.cuda.synchronize()
            n += 1; npair = globals().get("npair", 0) + 1; globals()["npair"] = npair
            if not torch.equal(b1.view(torch.int64), e.view(torch.int64)) or T.rel_linf(b2.cpu().numpy(), a.cpu().numpy()) > T.FIELD_RTOL:
                bad += 1; print("PAIR MISMATCH", seed, flush=True)
        except icebin_amd.IcebinHipError as ex:
            print("pair refused", seed, str(ex)[:90], flush=True)
    print("seed %d done (%d applies so far)" % (seed, n), flush=True)
print("kernels:", dict(seen))
print("applies checked:", n, "of them fused pairs:", globals().get("npair", 0), "mismatches:", bad)
sys.exit(1 if bad else 0)
