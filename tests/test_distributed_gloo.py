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


def _worker_class(rank, world, port, nf_total, G, nsteps, q):
    """The real FieldShardedApply under gloo: HostOps for the stream plumbing, the oracle as the
    rank-local SpMM; buffer layout, group choreography, flush of a partial group and the collectives
    are the class's own."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from icebin_amd import synthetic as syn
        from icebin_amd.distributed import FieldShardedApply, HostOps, field_shard
        from oracle import oracle as orc
        g = syn.make_grids("g50")
        em = syn.dome_elevmask(g)
        o = orc.Regridder(g).matrix_d("AvI", em)
        f0, f1 = field_shard(nf_total, world, rank)
        calls = {"one": 0, "many": 0}

        def local_apply(x, ldx, y, fill, stream_h):
            calls["one"] += 1
            y[:, :o.nrow] = torch.from_numpy(o.apply(x.numpy()))

        def local_apply_many(xs, ldx, ys, fill, stream_h):
            calls["many"] += 1
            for x, y in zip(xs, ys):
                y[:, :o.nrow] = torch.from_numpy(o.apply(x.numpy()))

        sh = FieldShardedApply(None, nf_total, ops=HostOps(), steps_per_gather=G, local_apply=local_apply,
                               local_apply_many=local_apply_many, nrow=o.nrow, ncol=o.ncol)
        xs = [syn.fields(nf_total, o.ncol, seed=900 + k) for k in range(nsteps)]       # same on every rank
        refs = [o.apply(x) for x in xs]
        ok = True

        def check_group(grp, steps):
            nonlocal ok
            sh.wait()
            for slot, k in enumerate(steps):
                got = sh.result(grp, slot).numpy()
                ok = ok and got.shape == refs[k].shape and bool(np.array_equal(got, refs[k]))

        k = 0
        # first group through the batched entry when it is a whole group, the rest one apply at a time
        if nsteps >= G:
            grp, slot0 = sh.apply_many_ptr([torch.from_numpy(xs[j][f0:f1].copy()) for j in range(G)], o.ncol)
            assert slot0 == 0
            check_group(grp, list(range(G)))
            k = G
        pending = []
        while k < nsteps:
            grp, slot = sh.apply(torch.from_numpy(xs[k][f0:f1].copy()))
            pending.append(k)
            if slot == G - 1:
                check_group(grp, pending)
                pending = []
            k += 1
        if pending:
            grp = (sh._i // G) & 1
            sh.flush()
            check_group(grp, pending)
        q.put((rank, ok, calls["one"], calls["many"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nf_total,G,nsteps", [(2, 8, 1, 3), (2, 8, 4, 10), (2, 7, 1, 3), (3, 8, 4, 6), (3, 9, 2, 5),
                                                     (2, 1, 2, 3)])
def test_field_sharded_apply_class_under_gloo(world, nf_total, G, nsteps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_class, args=(r, world, port, nf_total, G, nsteps, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert all(ok for _, ok, _, _ in got), got
    # the whole first group went through the batched entry on every rank that owns fields
    assert all(many == (1 if nsteps >= G and rank < nf_total else 0) for rank, _, _, many in got)


def test_sheet_partition():
    from icebin_amd.distributed import sheet_partition
    sizes = {"greenland": 4_366_000, "antarctica": 38_800_000}
    p8 = sheet_partition(sizes, 8)
    assert p8 == {"antarctica": (0, 7), "greenland": (7, 1)}
    assert sheet_partition(sizes, 2) == {"antarctica": (0, 1), "greenland": (1, 1)}
    assert sheet_partition(sizes, 1) == {"antarctica": (0, 1), "greenland": (0, 1)}
    for world in range(1, 17):
        p = sheet_partition({"a": 5, "b": 3, "c": 1}, world)
        if world >= 3:      # disjoint contiguous rank sets covering every rank
            spans = sorted(p.values())
            assert spans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:])) and sum(n for _, n in spans) == world
        else:
            assert all(n == 1 and 0 <= r < world for r, n in p.values())


def test_field_shard_partition():
    from icebin_amd.distributed import field_shard
    for nf in (1, 7, 8, 64, 128):
        for world in (1, 2, 3, 8):
            parts = [field_shard(nf, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == nf
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == nf


@pytest.mark.parametrize("extra", [["--fields", "3"], ["--fields-total", "7"]])
def test_bench_self_launches_its_ranks(extra):
    """`python bench.py --gpus 2` with no torchrun around it starts its own ranks as a child process, relays rank 0's JSON
    line and the exit code (the command used to die on an assert).  --dry-run plays the N > 1 choreography of
    the bench -- rendezvous on 127.0.0.1, field sharding, grouped all-gathers through FieldShardedApply, the cross-rank
    gather check, ranks_seen -- on CPU over gloo."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "5", "--warmup", "2",
                        "--queue-depth", "2"] + extra, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                    # ONE JSON line, rank 0's
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["value"] is None
    assert out["gather_check"] == {"pass": True, "ranks": 2}
    assert sorted(s["rank"] for s in out["ranks_seen"]) == [0, 1]
    assert out["fields_total"] == (6 if extra[0] == "--fields" else 7)


def test_bench_refuses_a_world_size_mismatch():
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


# ---- the sharded apply behind the C-ABI (ibh_comm, ibh_weighted_apply_sharded_device) ------------------------------------
def _sharded_cases():
    return (("AvI", 6, 0), ("AvI", 6, 4), ("IvA", 5, 2), ("EvI", 8, 0))       # (matrix, fields per rank, block_fields)


@pytest.mark.gpu
def test_sharded_apply_world1_through_rccl():
    """World 1 on the one GPU of the box, with a real RCCL communicator (ibh_comm_unique_id -> ncclGetUniqueId,
    ibh_comm_create -> ncclCommInitRank: librccl is dlopen'ed by the library): the sharded apply is the plain apply."""
    sys.path.insert(0, ROOT)
    import icebin_amd
    from icebin_amd import synthetic as syn
    from icebin_amd.distributed import Communicator, apply_sharded
    g = syn.make_grids("g20")
    em = syn.dome_elevmask(g)
    rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
    for comm in (Communicator(1, 0), Communicator(1, 0, rccl=True)):      # without RCCL; a one-rank RCCL communicator
        for name, nl, bf in _sharded_cases():
            w = rm.matrix(name)
            x = torch.from_numpy(syn.fields(nl, w.ncol_d, seed=11)).cuda()
            x[0, 3] = float("nan")
            ref = w.apply_device(x, fill=-2.0, force_conservation=False)
            out = apply_sharded(w, comm, x, fill=-2.0, block_fields=bf)
            comm.wait()
            torch.cuda.synchronize()
            if bf == 0:
                assert np.array_equal(out.cpu().numpy().view(np.uint64), ref.cpu().numpy().view(np.uint64)), name
            else:       # a block of fewer fields may take another launch shape: same sums, rounding-level differences at most
                a, b = out.cpu().numpy(), ref.cpu().numpy()
                assert np.array_equal(np.isnan(a), np.isnan(b)) and np.nanmax(np.abs(a - b)) <= 1e-12 * np.nanmax(np.abs(b)), name


def _worker_sharded(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import icebin_amd
        from icebin_amd import synthetic as syn
        from icebin_amd.distributed import Communicator, apply_sharded
        torch.cuda.set_device(0)                        # both ranks share the box's one card: the transport is host-staged
        g = syn.make_grids("g20")
        em = syn.dome_elevmask(g)
        rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em, scale=True, correctA=True)
        state = {}

        def exchange(d_base, count, stride, w_, r_, stream):
            # every rank's `count` doubles at d_base + rank*stride -> the same place on every peer (gloo, through the host).
            # STREAM-ORDERED: the copies are asynchronous on the stream the library hands over (its exchange stream) and only
            # THAT stream is waited for, never the device -- if the library did not order its exchange stream behind the SpMM
            # that produces the block (ready event), the block read here would still hold the -9 the test filled it with
            flat = state["out"]
            off = (d_base - flat.data_ptr()) // 8
            xs = state.get("xs")
            if xs is None or xs.cuda_stream != stream:
                xs = state["xs"] = torch.cuda.ExternalStream(stream)
            with torch.cuda.stream(xs):
                def carve():          # from ONE pinned buffer allocated up front (a pinned allocation may synchronise the device)
                    a = state["pin"][state["pin_off"]: state["pin_off"] + count]
                    state["pin_off"] += count
                    assert a.numel() == count
                    return a
                mine = carve()
                mine.copy_(flat[off + r_ * stride: off + r_ * stride + count], non_blocking=True)
                xs.synchronize()
                parts = [carve() for _ in range(w_)]
                dist.all_gather(parts, mine)
                for p in range(w_):
                    if p != r_:
                        flat[off + p * stride: off + p * stride + count].copy_(parts[p], non_blocking=True)
            state["calls"] = state.get("calls", 0) + 1
            state.setdefault("counts", []).append(int(count))
        # the exchanges run on a stream THIS process owns (ibh_comm_set_stream): torch's pinned-host allocator remembers the streams
        # a block was used on, and a stream the communicator created would die in ibh_comm_destroy with such blocks still alive
        # (round 4: a segmentation fault in the allocator's garbage collection).  The planes of the result arrays below are
        # rounded up to 512 bytes and the gap is this test's: planes_padded.
        xstream = torch.cuda.Stream()
        comm = Communicator(world, rank, exchange=exchange, stream=xstream, planes_padded=True)
        state["pin"], state["pin_off"] = torch.empty(1 << 21, dtype=torch.float64).pin_memory(), 0
        ok = True
        ncalls = []
        for name, nl, bf in _sharded_cases():
            w = rm.matrix(name)
            x_all = syn.fields(world * nl, w.ncol_d, seed=21)        # the same on every rank
            x_all[1, 5] = np.nan
            ld = (w.nrow_d + 63) // 64 * 64
            store = torch.full((world * nl, ld), -9.0, dtype=torch.float64, device="cuda")
            state["out"], state["calls"], state["pin_off"] = store.view(-1), 0, 0
            x_loc = torch.from_numpy(x_all[rank * nl:(rank + 1) * nl].copy()).cuda()
            out = apply_sharded(w, comm, x_loc, out_all=store[:, : w.nrow_d], fill=-2.0, block_fields=bf)
            comm.wait()
            torch.cuda.synchronize()
            ncalls.append(state["calls"])
            for p in range(world):                                   # every rank's rows against that rank's own apply
                xp = torch.from_numpy(x_all[p * nl:(p + 1) * nl].copy()).cuda()
                if bf == 0:
                    ref = w.apply_device(xp, fill=-2.0, force_conservation=False).cpu().numpy()
                else:
                    ref = np.concatenate([w.apply_device(xp[f0:f0 + bf].contiguous(), fill=-2.0, force_conservation=False).cpu().numpy()
                                          for f0 in range(0, nl, bf)])
                got = out[p * nl:(p + 1) * nl].cpu().numpy()
                ok = ok and bool(np.array_equal(got.view(np.uint64), ref.view(np.uint64)))
        # ldb as a TRUE leading dimension: the result is a column view of a wider array whose other columns are live data
        # (ADVICE r03): every plane travels by itself (nrow doubles), nothing outside the view changes on any rank
        # -- the default of a communicator, planes_padded = 0, whatever the width of the gap: 200 live columns and 10 (a
        # threshold on the gap used to decide, and a gap below 64 was overwritten on every peer)
        comm.set_option("planes_padded", 0)
        w, nl = rm.matrix("AvI"), 3
        for left, right in ((100, 100), (4, 6)):
            wide = torch.full((world * nl, w.nrow_d + left + right), float(rank + 7), dtype=torch.float64, device="cuda")
            state["out"], state["calls"], state["counts"], state["pin_off"] = wide.view(-1), 0, [], 0
            x_all = syn.fields(world * nl, w.ncol_d, seed=41)
            x_loc = torch.from_numpy(x_all[rank * nl:(rank + 1) * nl].copy()).cuda()
            out = apply_sharded(w, comm, x_loc, out_all=wide[:, left: left + w.nrow_d], fill=-2.0)
            comm.wait()
            torch.cuda.synchronize()
            ok = ok and state["calls"] == nl and set(state["counts"]) == {w.nrow_d}
            hw = wide.cpu().numpy()
            ok = ok and bool(np.all(hw[:, :left] == rank + 7) and np.all(hw[:, left + w.nrow_d:] == rank + 7))
            for p in range(world):
                xp = torch.from_numpy(x_all[p * nl:(p + 1) * nl].copy()).cuda()
                ref = w.apply_device(xp, fill=-2.0, force_conservation=False).cpu().numpy()
                ok = ok and bool(np.array_equal(np.ascontiguousarray(hw[p * nl:(p + 1) * nl, left: left + w.nrow_d]).view(np.uint64), ref.view(np.uint64)))
        comm.set_option("planes_padded", 1)
        # several field batches: one SpMM launch, one grouped exchange (the transport is called once per result here)
        from icebin_amd.distributed import apply_many_sharded
        w, nl, nb = rm.matrix("AvI"), 4, 3
        ld = (w.nrow_d + 63) // 64 * 64
        big = torch.full((nb, world * nl, ld), -9.0, dtype=torch.float64, device="cuda")
        state["out"], state["calls"], state["pin_off"] = big.view(-1), 0, 0
        x_all = [syn.fields(world * nl, w.ncol_d, seed=31 + k) for k in range(nb)]
        xs = [torch.from_numpy(xa[rank * nl:(rank + 1) * nl].copy()).cuda() for xa in x_all]
        outs = apply_many_sharded(w, comm, xs, [big[k][:, : w.nrow_d] for k in range(nb)], fill=-2.0)
        comm.wait()
        torch.cuda.synchronize()
        ncalls.append(state["calls"])
        for k in range(nb):
            for p in range(world):
                xp = torch.from_numpy(x_all[k][p * nl:(p + 1) * nl].copy()).cuda()
                ref = w.apply_device(xp, fill=-2.0, force_conservation=False).cpu().numpy()
                ok = ok and bool(np.array_equal(outs[k][p * nl:(p + 1) * nl].cpu().numpy().view(np.uint64), ref.view(np.uint64)))
        q.put((rank, ok, ncalls))
        # the communicator goes FIRST, with the transport's pinned buffer still alive and recorded on the exchange stream: the
        # stream is this process's (ibh_comm_set_stream), so nothing the allocator remembers dies with the communicator
        comm.close()
        torch.cuda.synchronize()
        state.clear()
        import gc
        gc.collect()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_apply_two_ranks_over_a_custom_transport():
    """ibh_weighted_apply_sharded_device with world 2: two processes share the box's GPU (RCCL refuses two ranks on one
    device), the exchange goes through ibh_comm_create_custom -- a gloo all-gather staged through the host -- so the library's
    own choreography runs with a peer: the local SpMM straight into this rank's rows of the gathered array, the exchange per
    field block with the right offsets / counts / strides, ordering between SpMM and exchange (the transport is asynchronous on
    the exchange stream and never synchronises the device), a result that is a column view of a wider array, ibh_comm_wait."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert all(ok for _, ok, _ in got), got
    # one exchange per field block: 6 fields in one, 6 in blocks of 4, 5 in blocks of 2, 8 in one
    assert got[0][2] == [1, 2, 3, 1, 3] and got[1][2] == [1, 2, 3, 1, 3], got


# ---- the sharded ASSEMBLY behind the C-ABI (ibh_regrid_matrices_matrix_d_sharded; SURVEY.md 8e "Assembly (cfg 5)") -------------------
def _hip_runtime():
    import ctypes as C
    rt = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    rt.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    rt.hipStreamSynchronize.argtypes = [C.c_void_p]
    return rt


def _worker_asm_sharded(rank, world, port, q, configs=("g5", "g20"), stage_mb=64):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import icebin_amd
        from icebin_amd import _capi, synthetic as syn
        from icebin_amd.distributed import Communicator
        torch.cuda.set_device(0)                        # the ranks share the box's one card: the transport is host-staged
        hip = _hip_runtime()
        stage = torch.empty(stage_mb << 20, dtype=torch.uint8).pin_memory()
        calls = {"blocks": 0, "gatherv": 0, "bytes": 0}

        def move(d_base, pieces, r_, stream):
            """pieces[k] = (byte offset, length) of rank k's piece of the array at d_base: every rank ends up with all of them"""
            maxn = max(n for _, n in pieces)
            off, n = pieces[r_]
            assert maxn * (len(pieces) + 1) <= stage.numel()
            mine = stage[:maxn]
            if n:
                assert hip.hipMemcpyAsync(mine.data_ptr(), d_base + off, n, 2, stream) == 0
            assert hip.hipStreamSynchronize(stream) == 0
            parts = [torch.empty(maxn, dtype=torch.uint8) for _ in pieces]
            dist.all_gather(parts, mine.clone())
            for k, (offk, nk) in enumerate(pieces):
                if k != r_ and nk:
                    dst = stage[(k + 1) * maxn:(k + 1) * maxn + nk]
                    dst.copy_(parts[k][:nk])
                    assert hip.hipMemcpyAsync(d_base + offk, dst.data_ptr(), nk, 1, stream) == 0
            assert hip.hipStreamSynchronize(stream) == 0
            calls["bytes"] += sum(nk for _, nk in pieces)

        def exchange(d_base, count, stride, w_, r_, stream):
            calls["blocks"] += 1
            move(d_base, [(8 * k * stride, 8 * count) for k in range(w_)], r_, stream)

        def gatherv(d_base, offs, w_, r_, stream):
            calls["gatherv"] += 1
            move(d_base, [(offs[k], offs[k + 1] - offs[k]) for k in range(w_)], r_, stream)

        comm = Communicator(world, rank, exchange=exchange, gatherv=gatherv)
        ok, notes = True, []

        def same(a, b, what):
            nonlocal ok
            good = (a.nrow_d, a.ncol_d, a.nnz) == (b.nrow_d, b.ncol_d, b.nnz)
            if good:
                for x, y in zip(a.csr_dense() + (a.wM, a.Mw, a.dim(0), a.dim(1)), b.csr_dense() + (b.wM, b.Mw, b.dim(0), b.dim(1))):
                    x = x.view(np.uint64) if x.dtype == np.float64 else x
                    y = y.view(np.uint64) if y.dtype == np.float64 else y
                    good = good and bool(np.array_equal(x, y))
            if not good:
                notes.append(what)
            ok = ok and good

        strict = [True]

        def unshared(what):
            """the shared build declined (every rank then builds redundantly, which is allowed -- and expected of none of the fixed
            grids; on the random grids of scratch/fuzz_sharded.py it is only noted)"""
            nonlocal ok
            notes.append(what)
            ok = ok and not strict[0]

        for config in configs:
            random_case = isinstance(config, str) and config.startswith("rand:")
            strict[0] = not random_case
            if random_case:                             # scratch/fuzz_sharded.py: the random sorted grids of the assembly tests
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import test_gpu_parity
                g, em = test_gpu_parity._random_grid(int(config[5:]), force_sorted=True)
            else:
                g = syn.make_grids(config)
                em = syn.dome_elevmask(g)
            mm = icebin_amd.from_synthetic(g)
            for scale, correctA in ((True, True), (False, False)):
                rm = mm.regrid_matrices("greenland", em, scale=scale, correctA=correctA)
                for name in ("AvI", "EvI", "IvA", "IvE", "AvX", "EvX", "XvA", "XvE"):
                    n0 = calls["gatherv"]
                    ws = rm.matrix_d_sharded(comm, name, scale=scale, correctA=correctA)
                    w1 = rm.matrix_d(name, scale=scale, correctA=correctA)
                    if not ws.built_sharded() or (calls["gatherv"] - n0 != 4 + 7 and not random_case):     # (exchange 2: four arrays, exchange 3: seven)
                        unshared("%s %s not shared (%d gathers)" % (config, name, calls["gatherv"] - n0))
                    same(ws, w1, "%s %s scale=%d correctA=%d" % (config, name, scale, correctA))
            rm = mm.regrid_matrices("greenland", em, scale=True, correctA=True)
            # not served by the shared build: built redundantly by every rank, same result (EvA: another pipeline; an identity set)
            ws, w1 = rm.matrix_d_sharded(comm, "EvA"), rm.matrix_d("EvA")
            ok = ok and not ws.built_sharded()
            same(ws, w1, config + " EvA")
            # an A/E-row matrix on an identity P set IS shared (round 5): Mw travels as {position, value} pairs
            for name, n_id in (("AvI", g["nI"]), ("EvI", g["nI"]), ("AvX", len(g["ex_area"])), ("EvX", len(g["ex_area"]))):
                ident = [icebin_amd.SparseSet.identity(n_id) for _ in range(2)]
                ws, w1 = rm.matrix_d_sharded(comm, name, (None, ident[0])), rm.matrix_d(name, (None, ident[1]))
                if not ws.built_sharded():
                    unshared("%s %s on an identity set not shared" % (config, name))
                same(ws, w1, "%s %s identity P set" % (config, name))
            # X rows on the identity dimX with a column set of their own; I rows numbered by the build over a pre-populated dimE
            for name in ("XvA", "XvE"):
                ident = [icebin_amd.SparseSet.identity(len(g["ex_area"])) for _ in range(2)]
                ws, w1 = rm.matrix_d_sharded(comm, name, (ident[0], None)), rm.matrix_d(name, (ident[1], None))
                if not ws.built_sharded():
                    unshared("%s %s on the identity dimX not shared" % (config, name))
                same(ws, w1, "%s %s identity dimX" % (config, name))
            keysE = rm.matrix_d("EvI").dim(0)
            dE = [icebin_amd.SparseSet(g["nA"] * len(g["hcdefs"]), keysE[::-1].copy()) for _ in range(2)]      # (any order: here reversed)
            ws, w1 = rm.matrix_d_sharded(comm, "IvE", (None, dE[0])), rm.matrix_d("IvE", (None, dE[1]))
            if not ws.built_sharded():
                unshared("%s IvE over a pre-populated dimE not shared" % config)
            same(ws, w1, "%s IvE pre-populated dimE" % config)
            # I rows on the identity dimI (rows in ice-cell order: the row lengths are merged before the row pointer is scanned, the
            # rows travel in first-seen order), with a column set of their own
            for name in ("IvA", "IvE"):
                ident = [icebin_amd.SparseSet.identity(g["nI"]) for _ in range(2)]
                ws, w1 = rm.matrix_d_sharded(comm, name, (ident[0], None)), rm.matrix_d(name, (ident[1], None))
                if not ws.built_sharded():
                    unshared("%s %s on the identity dimI not shared" % (config, name))
                same(ws, w1, "%s %s identity dimI" % (config, name))
            # the coupler's step (IceCoupler.cpp:361-377, 462-467), all four calls shared: EvI {dimE1, identity dimI}, AvI {dimA1,
            # identity dimI} -- the CSR pieces of an A/E-row matrix are contiguous whatever the column numbering, Mw travels as pairs;
            # XvE {identity dimX, dimE1 as EvI left it} -- a block of cells is a block of rows, the column sums travel as pairs;
            # IvE {identity dimI, dimE1} -- rows in ice-cell order, copied to their places on every rank
            nE = g["nA"] * len(g["hcdefs"])
            sets = []
            for sharded in (True, False):
                dimI, dimX, dimE, dimA = icebin_amd.SparseSet.identity(g["nI"]), icebin_amd.SparseSet.identity(len(g["ex_area"])), icebin_amd.SparseSet(nE), icebin_amd.SparseSet(g["nA"])
                build = (lambda name, dims, **kw: rm.matrix_d_sharded(comm, name, dims, **kw)) if sharded else (lambda name, dims, **kw: rm.matrix_d(name, dims, **kw))
                sets.append([build("EvI", (dimE, dimI), scale=False, correctA=False), build("AvI", (dimA, dimI), scale=False, correctA=True),
                             build("IvE", (dimI, dimE), scale=True, correctA=True), build("XvE", (dimX, dimE), scale=False, correctA=True)])
            for a, b, name in zip(sets[0], sets[1], ("EvI", "AvI", "IvE", "XvE")):
                if not a.built_sharded():
                    unshared("%s coupler %s: built_sharded() %s" % (config, name, a.built_sharded()))
                same(a, b, "%s coupler %s" % (config, name))
        # an elevation above the last class: every rank raises the reference's error, whichever rank's block holds the cell
        g = syn.make_grids("g20")
        em = syn.dome_elevmask(g)
        em[np.flatnonzero(np.isfinite(em))[-3]] = 1e5
        rm = icebin_amd.from_synthetic(g).regrid_matrices("greenland", em)
        try:
            rm.matrix_d_sharded(comm, "EvI")
            ok = False
            notes.append("no range error")
        except _capi.IcebinHipError as e:
            ok = ok and e.code == _capi.IBH_ERANGE and "out of bounds" in str(e)
        q.put((rank, ok, notes, dict(calls)))
        torch.cuda.synchronize()
        del comm
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_assembly_at_1km_takes_the_streamed_builds_size_branches():
    """The shared build with two ranks on the 1 km grid (4.4 M exchange cells): a rank's block starts in the middle of the grid
    (sx0 != 0), the emit pass walks tiles, the byte passes run several waves, the kernel choices that depend on the number of
    ranges are taken from the WHOLE grid -- all eight matrices, both branches, bitwise the single-rank build; and the coupler's
    four calls on identity / pre-populated sets come out equal (redundantly built)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_asm_sharded, args=(r, 2, port, q, ("g1",), 512)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert all(ok for _, ok, _, _ in got), got


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_assembly_is_the_single_rank_build_bitwise(world):
    """ibh_regrid_matrices_matrix_d_sharded with 2 and 3 ranks (processes sharing the box's GPU, a host-staged gloo transport for
    the equal blocks and for the pieces of unequal size): every rank's matrix -- dims, CSR, wM, Mw -- is bitwise the single-rank
    build for the eight matrices the shared build serves, at 5 km (ranges of ~1 500 cells, ice cells straddling the ranks' blocks)
    and 20 km; builds it does not serve come out redundantly and equal; an out-of-range elevation raises on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_asm_sharded, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert all(ok for _, ok, _, _ in got), got
    assert all(c["gatherv"] > 0 and c["blocks"] > 0 for _, _, _, c in got), got
