"""Time matrix assembly on the GPU for several configs/matrices (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import synthetic as syn
for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]: icebin_amd.set_tuning(k, int(v))
for cfg in sys.argv[1].split(","):
    g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
    mm = icebin_amd.from_synthetic(g)
    rm = mm.regrid_matrices("greenland", em, scale=True, correctA=True)
    nX, nI = len(g["ex_area"]), g["nI"]
    # what every matrix of an elevmask shares: the object's copy of the mask and (grids of the streamed build) one class byte per ice cell
    em_dev = torch.from_numpy(em).cuda(); tc = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); rm2 = mm.regrid_matrices("greenland", em_dev, scale=True, correctA=True); torch.cuda.synchronize(); tc.append(time.perf_counter() - t0); del rm2
    print("%-5s regrid_matrices(elevmask in HBM), once per elevmask: %9.3f ms  (copy of %.1f MB%s)" % (cfg, min(tc) * 1e3, 8 * nI / 1e6, " + class bytes" if nX >= 1 << 20 else ""), flush=True)
    for name in sys.argv[2].split(","):
        w = rm.matrix(name); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); w = rm.matrix(name); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        B = 16 * nX + 8 * nI + 12 * w.nnz + 4 * (w.nrow_d + 1) + 8 * (w.nrow_d + w.ncol_d)
        print("%-5s %-4s nX=%9d nnz=%9d  %9.3f ms   B_asm=%8.1f MB  %7.1f GB/s" % (cfg, name, nX, w.nnz, t * 1e3, B / 1e6, B / t / 1e9), flush=True)
