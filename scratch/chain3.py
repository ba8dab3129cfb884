"""BASELINE config 3: 5 km, 40 elevation classes, the chain EvI -> AvE -> IvA on 16 fields (ice -> elevation classes ->
atmosphere -> ice), (a) as three stream-ordered applies, (b) captured once into a hipGraph and replayed: ONE submission
per chain.  Prints wall microseconds per chain (many back-to-back chains, cold rotating inputs) and checks the result of the
last replay against the eager chain bitwise.  usage: chain3.py [nfields] [config]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icebin_amd
from icebin_amd import synthetic as syn

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = sys.argv[2] if len(sys.argv) > 2 else "g5"
g = syn.make_grids(cfg); em = syn.dome_elevmask(g)
rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
W = {n: rm.matrix(n) for n in ("EvI", "AvE", "IvA")}
for w in W.values():
    w.prepare(nf, 1)
nI_d, nE_d, nA_d = W["EvI"].ncol_d, W["EvI"].nrow_d, W["AvE"].nrow_d
assert W["AvE"].ncol_d == nE_d and W["IvA"].ncol_d == nA_d
nbuf = max(2, -(-(512 << 20) // (8 * nf * nI_d)))
x0 = torch.from_numpy(syn.fields(nf, nI_d)).cuda()
X = [x0 + 1e-3 * b for b in range(min(nbuf, 32))]
xin = torch.empty_like(x0)
yE = torch.zeros((nf, nE_d), dtype=torch.float64, device="cuda")
yA = torch.zeros((nf, nA_d), dtype=torch.float64, device="cuda")
ld = (W["IvA"].nrow_d + 63) // 64 * 64
yI = torch.zeros((nf, ld), dtype=torch.float64, device="cuda")[:, :W["IvA"].nrow_d]


def chain(x):
    W["EvI"].apply_device(x, out=yE, force_conservation=False)
    W["AvE"].apply_device(yE, out=yA, force_conservation=False)
    W["IvA"].apply_device(yA, out=yI, force_conservation=False)


for x in X[:4]:
    chain(x)
torch.cuda.synchronize()
N = 400
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(N):
    chain(X[i % len(X)])
e1.record(); torch.cuda.synchronize()
eager_us = e0.elapsed_time(e1) * 1e3 / N
ref = yI.clone()
# one graph per input buffer (a graph bakes its pointers in): capture the chain on every rotating input
graphs = []
for x in X:
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        chain(x)
    graphs.append(gr)
for gr in graphs[:4]:
    gr.replay()
torch.cuda.synchronize()
e0.record()
for i in range(N):
    graphs[i % len(graphs)].replay()
e1.record(); torch.cuda.synchronize()
graph_us = e0.elapsed_time(e1) * 1e3 / N
same = bool(torch.equal(yI.view(torch.int64), ref.view(torch.int64)))
B = sum(12 * w.nnz + 4 * (w.nrow_d + 1) + 8 * nf * (w.ncol_d + w.nrow_d) for w in W.values())
# (c) EvI and AvE as ONE launch (ibh_weighted_apply_pair_device: AvE rides in the row-group kernel's epilogue), then IvA
W["EvI"].pair_prepare(W["AvE"], nf)


def chain_fused(x):
    W["EvI"].apply_pair_device(W["AvE"], x, out1=yE, out2=yA)
    W["IvA"].apply_device(yA, out=yI, force_conservation=False)


for x in X[:4]:
    chain_fused(x)
torch.cuda.synchronize()
e0.record()
for i in range(N):
    chain_fused(X[i % len(X)])
e1.record(); torch.cuda.synchronize()
fused_eager_us = e0.elapsed_time(e1) * 1e3 / N
fgraphs = []
for x in X:
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        chain_fused(x)
    fgraphs.append(gr)
for gr in fgraphs[:4]:
    gr.replay()
torch.cuda.synchronize()
e0.record()
for i in range(N):
    fgraphs[i % len(fgraphs)].replay()
e1.record(); torch.cuda.synchronize()
fused_graph_us = e0.elapsed_time(e1) * 1e3 / N
fused_same = bool(torch.equal(yI.view(torch.int64), ref.view(torch.int64)))
fused_close = float(((yI - ref).abs().max() / ref.abs().max()).item())
# (c2) the chain call: the fused pair, then IvA dispatched without a queue barrier and waiting on a device counter
def chain_overlapped(x):
    W["EvI"].apply_chain_device(W["AvE"], W["IvA"], x, out1=yE, out2=yA, out3=yI)


yI.zero_()
icebin_amd.set_tuning("chain_overlap", 1)
for x in X[:4]:
    chain_overlapped(x)
torch.cuda.synchronize()
e0.record()
for i in range(N):
    chain_overlapped(X[i % len(X)])
e1.record(); torch.cuda.synchronize()
over_eager_us = e0.elapsed_time(e1) * 1e3 / N
over_same = bool(torch.equal(yI.view(torch.int64), ref.view(torch.int64)))
yI.zero_()
ograph = torch.cuda.CUDAGraph()
with torch.cuda.graph(ograph):
    chain_overlapped(X[(N - 1) % len(X)])
ograph.replay(); ograph.replay(); torch.cuda.synchronize()
over_graph_same = bool(torch.equal(yI.view(torch.int64), ref.view(torch.int64)))
e0.record()
for i in range(N):
    ograph.replay()
e1.record(); torch.cuda.synchronize()
over_graph_us = e0.elapsed_time(e1) * 1e3 / N
icebin_amd.set_tuning("chain_overlap", -2 ** 31)
# (d) 16 consecutive fused chains (16 time steps' worth, rotating inputs) captured as ONE graph: one submission per 16 chains
big = torch.cuda.CUDAGraph()
with torch.cuda.graph(big):
    for j in range(16):
        chain_fused(X[j % len(X)])
big.replay(); torch.cuda.synchronize()
e0.record()
for i in range(N // 16):
    big.replay()
e1.record(); torch.cuda.synchronize()
big_us = e0.elapsed_time(e1) * 1e3 / (N // 16 * 16)
# (e) 16 INDEPENDENT field batches through the chain (e.g. 16 tracer groups): three batched launches for all of them
nb = min(16, len(X))
W["EvI"].prepare(nf, nb); W["AvE"].prepare(nf, nb); W["IvA"].prepare(nf, nb)
yEs = [torch.zeros_like(yE) for _ in range(nb)]
yAs = [torch.zeros_like(yA) for _ in range(nb)]
yIs = [torch.zeros((nf, ld), dtype=torch.float64, device="cuda")[:, :W["IvA"].nrow_d] for _ in range(nb)]


def chains_batched():
    W["EvI"].apply_many_device(X[:nb], yEs, force_conservation=False)
    W["AvE"].apply_many_device(yEs, yAs, force_conservation=False)
    W["IvA"].apply_many_device(yAs, yIs, force_conservation=False)


for _ in range(3):
    chains_batched()
torch.cuda.synchronize()
e0.record()
for i in range(N // nb):
    chains_batched()
e1.record(); torch.cuda.synchronize()
batched_us = e0.elapsed_time(e1) * 1e3 / (N // nb * nb)
print("%s chain EvI->AvE->IvA, %d fields: kernels %s; %.1f MB algorithmic per chain; eager %.2f us per chain, hipGraph replay %.2f us per chain "
      "(%.1f %% of 8 TB/s); graph result == eager result bitwise: %s" %
      (cfg, nf, "/".join(W[n].last_kernel() for n in ("EvI", "AvE", "IvA")), B / 1e6, eager_us, graph_us, B / graph_us / 1e3 / 8000 * 100, same), flush=True)
print("   EvI+AvE fused into one launch (apply_pair_device) + IvA: eager %.2f us per chain, hipGraph replay %.2f us (%.1f %% of 8 TB/s); final field bitwise "
      "the three-launch chain's: %s (max rel diff %.1e); 16 fused chains in one graph: %.2f us per chain" %
      (fused_eager_us, fused_graph_us, B / fused_graph_us / 1e3 / 8000 * 100, fused_same, fused_close, big_us), flush=True)
print("   the chain call (apply_chain_device: IvA dispatched without a barrier behind the pair kernel, waits on a device counter): eager %.2f us per chain, "
      "final field bitwise the three-launch chain's: %s; as a replayed hipGraph %.2f us, bitwise: %s" % (over_eager_us, over_same, over_graph_us, over_graph_same), flush=True)
print("   %d independent %d-field batches through the chain, three batched launches (apply_many_device): %.2f us per chain (%.1f %% of 8 TB/s)"
      % (nb, nf, batched_us, B / batched_us / 1e3 / 8000 * 100), flush=True)
