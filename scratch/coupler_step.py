"""Time one coupling step's worth of this path (dev tool), shaped like IceCoupler.cpp:361-468 + :203-252, :445:
new elevation mask -> regrid_matrices -> EvI (unscaled, dimE1 fresh, dimI identity), AvI (unscaled),
IvE (scaled, shared dimE1), XvE (unscaled, shared dimE1, dimX identity) -> the two fused applies
IvE*(V*T+b) and AvI*(V*T+b).  Wall-clock per phase, best of 7 steps after 2 warm-ups."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import synthetic as syn
for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]: icebin_amd.set_tuning(k, int(v))
for cfg in sys.argv[1].split(","):
    g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
    mm = icebin_amd.from_synthetic(g)
    nI, nX, nE = g["nI"], len(g["ex_area"]), g["nA"] * 40
    nvar_in, nvar_out = 8, 6
    T = np.zeros((nvar_in, nvar_out)); T[np.arange(nvar_out), np.arange(nvar_out)] = 1.0; T[6, 2] = 0.5
    b = np.linspace(0.0, 1.0, nvar_out)
    best = None
    best_batch = None
    # an ice model that keeps its elevation mask in HBM hands over a device pointer (ibh_regrid_matrices_create_device):
    em_dev = torch.from_numpy(em).cuda()
    best_dev = None
    for step in range(9):
        em_step = em_dev + 0.01 * step
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rm_dev = mm.regrid_matrices("greenland", em_step, scale=True, correctA=False)
        torch.cuda.synchronize(); td = (time.perf_counter() - t0) * 1e3
        if step >= 2:
            best_dev = td if best_dev is None else min(best_dev, td)
        del rm_dev
    for step in range(9):
        torch.cuda.synchronize(); t = [time.perf_counter()]
        rm = mm.regrid_matrices("greenland", em + 0.01 * step, scale=True, correctA=False)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        dimI, dimE = icebin_amd.SparseSet.identity(nI), icebin_amd.SparseSet(nE)
        EvI = rm.matrix_d("EvI", (dimE, dimI), scale=False, correctA=False); t.append(time.perf_counter())
        AvI = rm.matrix_d("AvI", (None, dimI), scale=False, correctA=False); t.append(time.perf_counter())
        IvE = rm.matrix_d("IvE", (dimI, dimE), scale=True, correctA=False); t.append(time.perf_counter())
        dimX = icebin_amd.SparseSet.identity(nX)
        XvE = rm.matrix_d("XvE", (dimX, dimE), scale=False, correctA=False); t.append(time.perf_counter())
        vE = torch.randn((nvar_in, IvE.ncol_d), dtype=torch.float64, device="cuda")
        vI = torch.randn((nvar_in, AvI.ncol_d), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize(); t.append(time.perf_counter())
        yI = IvE.apply_transformed_device(vE, T, b)
        yA = AvI.apply_transformed_device(vI, T, b)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        d = np.diff(t) * 1e3
        if step >= 2 and (best is None or d.sum() < best.sum()):
            best = d
        # the same four matrices through ONE batched call (ibh_regrid_matrices_matrix_batch)
        dimI, dimE, dimX = icebin_amd.SparseSet.identity(nI), icebin_amd.SparseSet(nE), icebin_amd.SparseSet.identity(nX)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ws = rm.matrix_batch([("EvI", (dimE, dimI), False, False), ("AvI", (None, dimI), False, False),
                              ("IvE", (dimI, dimE), True, False), ("XvE", (dimX, dimE), False, False)])
        tb = (time.perf_counter() - t0) * 1e3
        assert ws[2].nnz == IvE.nnz and ws[0].nrow_d == EvI.nrow_d
        if step >= 2:
            best_batch = tb if best_batch is None else min(best_batch, tb)
    names = ["regrid_matrices(mask from pageable host memory)", "EvI", "AvI", "IvE", "XvE", "(field setup)", "2 fused applies"]
    print("%-5s nX=%d" % (cfg, nX), "regrid_matrices(mask already in HBM) %.3f ms " % best_dev,
          "  ".join("%s %.3f ms" % (n, v) for n, v in zip(names, best) if not n.startswith("(")),
          " | matrices %.3f ms | batched call %.3f ms" % (best[1:5].sum(), best_batch), flush=True)
