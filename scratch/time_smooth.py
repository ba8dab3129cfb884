import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import synthetic as syn
for cfg, sig in (("g20", (60e3, 60e3, 250.)), ("g5", (50e3, 50e3, 100.)), ("g5", (25e3, 25e3, 100.))):
    g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
    rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True, sigma=sig)
    for name in ("IvA", "IvE"):
        w = rm.matrix(name); torch.cuda.synchronize()
        t0 = time.perf_counter(); w = rm.matrix(name); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        x = torch.randn((16, w.ncol_d), dtype=torch.float64, device="cuda")
        y = w.apply_device(x); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): w.apply_device(x, out=y)
        e1.record(); torch.cuda.synchronize()
        print("%-4s %-4s sigma=%s  %d x %d nnz=%d  build %.2f ms  apply(16 fields, with conservation correction) %.1f us  kernel=%s" % (cfg, name, sig, w.nrow_d, w.ncol_d, w.nnz, dt * 1e3, e0.elapsed_time(e1) / 20 * 1e3, w.last_kernel()), flush=True)
