"""Build times of the plan-based assembly under the three workgroup shapes of its per-range kernels
(ibh_set_tuning("assemble_range_shape", 0 | 1 | 2); -1 = by range size).  usage: asm_shapes.py g5,g1,a1h AvI,EvI"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import synthetic as syn
for cfg in sys.argv[1].split(","):
    g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
    rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", torch.from_numpy(em).cuda(), scale=True, correctA=False)
    for name in sys.argv[2].split(","):
        out = []
        for shape in (-1, 0, 1, 2):
            icebin_amd.set_tuning("assemble_range_shape", shape)
            best = 1e9
            for rep in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                w = rm.matrix(name)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
                del w
            out.append("%s %.3f ms" % ({-1: "auto", 0: "128x2", 1: "256x4", 2: "1024x4"}[shape], best * 1e3))
        icebin_amd.set_tuning("assemble_range_shape", -1)
        print("%-4s %-4s  " % (cfg, name) + "   ".join(out), flush=True)
