"""The N>1 path on CPU: world_size-2 (and ragged world_size-3) gloo runs of the field shard +
all-gather reassembly used by bench.py --gpus N; the local SpMM is played by the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nf_total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from icebin_amd import synthetic as syn
        from icebin_amd.distributed import all_gather_fields, field_shard
        from oracle import oracle as orc
        g = syn.make_grids("g50")
        em = syn.dome_elevmask(g)
        o = orc.Regridder(g).matrix_d("AvI", em)            # replicated matrix
        x = syn.fields(nf_total, o.ncol)                    # the global field set (same seed on every rank)
        f0, f1 = field_shard(nf_total, world, rank)
        y_local = torch.from_numpy(o.apply(x[f0:f1]))       # rank-local SpMM of its own fields
        y_all = all_gather_fields(y_local, nf_total).numpy()
        ref = o.apply(x)
        q.put((rank, bool(np.array_equal(y_all, ref)), (f0, f1)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nf_total", [(2, 8), (2, 7), (3, 8)])
def test_field_shard_all_gather(world, nf_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nf_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert all(ok for _, ok, _ in got)
    # shards tile [0, nf_total) contiguously in rank order
    edges = [s for _, _, s in got]
    assert edges[0][0] == 0 and edges[-1][1] == nf_total
    assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))


def test_field_shard_partition():
    from icebin_amd.distributed import field_shard
    for nf in (1, 7, 8, 64, 128):
        for world in (1, 2, 3, 8):
            parts = [field_shard(nf, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == nf
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == nf
