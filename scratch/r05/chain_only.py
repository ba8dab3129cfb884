import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import icebin_amd
from icebin_amd import synthetic as syn
mode = sys.argv[1]
nf = 16
for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]: icebin_amd.set_tuning(k, int(v))
g = syn.make_grids("g5"); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
W = {n: rm.matrix(n) for n in ("EvI", "AvE", "IvA")}
for w in W.values(): w.prepare(nf, 1)
W["EvI"].pair_prepare(W["AvE"], nf)
x = torch.from_numpy(syn.fields(nf, W["EvI"].ncol_d)).cuda()
X = [x + 1e-3 * b for b in range(8)]
yE = torch.zeros((nf, W["EvI"].nrow_d), dtype=torch.float64, device="cuda"); yA = torch.zeros((nf, W["AvE"].nrow_d), dtype=torch.float64, device="cuda")
yI = torch.zeros((nf, W["IvA"].nrow_d), dtype=torch.float64, device="cuda")
for i in range(200):
    if mode == "chain":
        W["EvI"].apply_chain_device(W["AvE"], W["IvA"], X[i % 8], out1=yE, out2=yA, out3=yI)
    else:
        W["EvI"].apply_pair_device(W["AvE"], X[i % 8], out1=yE, out2=yA)
        W["IvA"].apply_device(yA, out=yI, force_conservation=False)
torch.cuda.synchronize()
print("done", mode)
