"""A few builds of one matrix on one grid (for a rocprofv3 --kernel-trace run): usage build_trace.py config matrix [n]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, icebin_amd
from icebin_amd import synthetic as syn
cfg, name, n = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 12
for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]: icebin_amd.set_tuning(k, int(v))
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
for i in range(n):
    W = rm.matrix(name)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(50): W = rm.matrix(name)
torch.cuda.synchronize()
print("%s %s: %.1f us per build (host clock, 50 builds)" % (cfg, name, (time.perf_counter() - t) / 50 * 1e6), flush=True)
