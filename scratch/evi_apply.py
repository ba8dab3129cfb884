"""EvI apply: row-by-row (rowblock) against the band structure (rowdual) and the column sweep (colsweep), single launches and
batched, us per 64-field apply.  usage: evi_apply.py config [matrix [sweep_tb]]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import _capi, synthetic as syn
cfg = sys.argv[1]; name = sys.argv[2] if len(sys.argv) > 2 else "EvI"; nf = 64
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
L = _capi.lib(); st = torch.cuda.current_stream(); cs = C.c_void_p(st.cuda_stream)
modes = os.environ.get("EVI_MODES", "rowblock,rowdual,colsweep").split(",")
for k, v in [kv.split("=") for kv in os.environ.get("EVI_TUNE", "").split(",") if kv]: icebin_amd.set_tuning(k, int(v))
for mode in modes:
    icebin_amd.set_tuning("rowdual_auto", 1 if mode == "rowdual" else 0)
    icebin_amd.set_tuning("sweep_auto", 1 if mode == "colsweep" else 0)
    icebin_amd.set_tuning("rowdual_min_work", 1)
    icebin_amd.set_tuning("sweep_min_work", 1)
    if len(sys.argv) > 3: icebin_amd.set_tuning("sweep_tb", int(sys.argv[3]))
    W = rm.matrix(name)
    nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
    B = 12 * nnz + 4 * (nrow + 1) + 8 * nf * (ncol + nrow)
    nbuf = max(2, -(-(512 << 20) // (8 * nf * ncol)))
    x0 = torch.from_numpy(syn.fields(nf, ncol)).cuda()
    X = [x0 + 1e-3 * b for b in range(nbuf)]
    Y = [torch.zeros((nf, nrow), dtype=torch.float64, device="cuda") for _ in range(16)]
    out = []
    for depth in (1, 4, 16):
        steps = 320 if B < 2e8 else 64
        plans = []
        for i in range(0, steps, depth):
            m = min(depth, steps - i)
            plans.append((m, (C.c_void_p * m)(*[X[(i + j) % nbuf].data_ptr() for j in range(m)]), (C.c_void_p * m)(*[Y[j].data_ptr() for j in range(m)])))
        def run():
            for m, xa, ya in plans:
                assert L.ibh_weighted_apply_many_device(W._h, m, xa, nf, ncol, ya, nrow, float("nan"), 0, cs) == 0
        run(); run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); run(); e1.record(st); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / steps * 1e3
        out.append("d%-2d %7.2f us %4.1f%%" % (depth, us, B / us / 1e3 / 8000 * 100))
    print("%s %s %-8s (kernel %s) %d x %d nnz %d B=%.1f MB: %s" % (cfg, name, mode, W.last_kernel(), nrow, ncol, nnz, B / 1e6, " | ".join(out)), flush=True)
