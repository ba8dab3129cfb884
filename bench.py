#!/usr/bin/env python3
"""bench.py -- headline benchmark of the regrid hot path on MI355X.

Metric (BASELINE.json): regridded cells/sec + achieved HBM GB/s, AvI 5 km -> 2x2.5 deg, 64 fields.
A "step" is one Weighted::apply of the AvI matrix to one batch of 64 synthetic fields that
are already resident in HBM.  Steps cycle through enough distinct field batches (>= 512 MiB in
total) that no batch can be served from the 256 MiB Infinity Cache: every step streams its X
from HBM ("cold" numbers; `roofline.cold_check` SHOWS it: the same launch timed over
1 / 7 / 14 / 28 rotating batches).  Steps are submitted --queue-depth at a time through
ibh_weighted_apply_many_device (ONE launch serves up to 32 independent 64-field applies: a single
40 MB apply is latency-sized on this chip; `roofline.single_launch` is the one-apply-per-launch
figure of the same run).  Every step still reads its own 39 MB of X from HBM and writes its own
Y; the CSR is counted ONCE per launch in the roofline bytes.

Statistics: after W warm-up steps the K-step timed region (barrier + synchronize on both sides, max
over ranks) is repeated --repeats R (default 21) times; `value` / `ms_per_step` are the MEDIAN region,
`roofline.kernel_us` the median over all R x launches kernel durations (HIP events attached to each
SpMM dispatch), with min / max beside them.

N GPUs: `python bench.py --gpus N` starts its own `python -m torch.distributed.run` (one rank per GPU)
as a child process when it was not launched by one; two modes:
  * default (weak scaling): every rank applies the replicated CSR to its own 64 fields of a
    64*N-field regrid;
  * --fields-total F (strong scaling, BASELINE config 4: `--config g1 --fields-total 64`, AvI and
    `--matrix IvA`): ONE F-field regrid, F/N fields per rank.
Either way the SpMM needs no communication; RCCL all-gathers reassemble the [fields, nrow] results on
every rank, one collective per group of --queue-depth applies, issued on a second stream so it overlaps
the following SpMMs.  spmm_only_ms (no collectives) and spmm_plus_gather_ms are reported separately;
every rank checks the gathered blocks of two ranks against a local apply (`gather_check`).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def spmm_bytes(nnz, nrow, ncol, nf):
    """Algorithmic bytes of one apply: CSR once + every X element once + Y once."""
    return 12 * nnz + 4 * (nrow + 1) + 8 * nf * ncol + 8 * nf * nrow


def asm_bytes(nX, nI, nnz, nrow, ncol):
    """Algorithmic bytes of one matrix assembly."""
    return 16 * nX + 8 * nI + 12 * nnz + 4 * (nrow + 1) + 8 * (nrow + ncol)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1920)
    ap.add_argument("--warmup", type=int, default=160)
    ap.add_argument("--repeats", type=int, default=21, help="repetitions of the K-step timed region (median reported)")
    ap.add_argument("--config", default="g5", help="synthetic grid config (icebin_amd/synthetic.py)")
    ap.add_argument("--matrix", default="AvI")
    ap.add_argument("--fields", type=int, default=64, help="fields per GPU (weak scaling)")
    ap.add_argument("--fields-total", type=int, default=0, help="strong scaling: ONE regrid of this many fields, sharded over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-launch and cold-check side measurements (profiling runs: only the headline launches)")
    ap.add_argument("--variants", action="store_true", help="also time the peak-size (all-unmasked) variant; informational")
    ap.add_argument("--all-unmasked", action="store_true", help="every ice cell carries ice (peak-size variant, SURVEY.md 8d)")
    ap.add_argument("--warm", action="store_true", help="reuse ONE field batch (Infinity-Cache-resident numbers)")
    ap.add_argument("--queue-depth", type=int, default=32, help="applies submitted per launch (1..32); N>1: also applies per all-gather")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT", help="ibh_set_tuning override (experiments)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not attach HIP events to the SpMM dispatches (plain launches: what a rocprofv3 kernel trace of a product call sees; roofline figures then come from the region events)")
    ap.add_argument("--dry-run", action="store_true", help="rehearse launcher + rendezvous + sharding + gather check on CPU (gloo, scipy apply); NOT a measurement")
    return ap.parse_args(argv)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` not started by torchrun: start the N ranks ourselves, as a CHILD process (never
    exec: nothing here has touched the GPU, and nothing may replace a process that has), one rank per GPU,
    rendezvous on 127.0.0.1.  stdout / stderr are inherited, so rank 0's JSON line is relayed as it is printed;
    the exit code is the child's."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def median(v):
    v = sorted(v)
    n = len(v)
    return None if n == 0 else (v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2]))


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.dry_run:
        sys.exit(dry_run(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or under "
                         "torch.distributed.run with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    # one rank per GPU.  Rehearsal of the N>1 code path on a box with fewer GPUs than ranks (never a measurement):
    # ICEBIN_BENCH_BACKEND=gloo lets several ranks share a card (RCCL refuses two ranks on one device)
    backend = os.environ.get("ICEBIN_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also rehearsable at N=1)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import icebin_amd
    from icebin_amd import _capi
    from icebin_amd import synthetic as syn
    from icebin_amd.distributed import FieldShardedApply, field_shard
    _capi.check(_capi.lib().ibh_set_device(dev_index))      # handles bind to the device current at creation
    for kv in args.tune:
        k, v = kv.split("=")
        icebin_amd.set_tuning(k, int(v))

    # ---- the workload: assemble the matrix on this GPU (replicated on every rank) -----------------
    grids = syn.make_grids(args.config)
    em = syn.dome_elevmask(grids, all_unmasked=args.all_unmasked)
    mm = icebin_amd.from_synthetic(grids)
    t0 = time.perf_counter()
    rm = mm.regrid_matrices("greenland", em, scale=True, correctA=True)
    W = rm.matrix(args.matrix)
    torch.cuda.synchronize()
    t_asm_first = time.perf_counter() - t0
    t_asm = float("inf")
    for _ in range(5):                                  # (steady state: the best of five rebuilds -- the first ones still warm caches)
        t0 = time.perf_counter()
        W2 = rm.matrix(args.matrix)
        torch.cuda.synchronize()
        t_asm = min(t_asm, time.perf_counter() - t0)
        del W2
    nrow, ncol, nnz = W.nrow_d, W.ncol_d, W.nnz
    strong = args.fields_total > 0
    if strong:
        nf_total = args.fields_total
        assert nf_total % world == 0, "--fields-total must be a multiple of the number of ranks"
        f0, f1 = field_shard(nf_total, world, rank)
        nf = f1 - f0
    else:
        nf, nf_total = args.fields, args.fields * world
    n_unmasked = int(np.isfinite(em).sum())
    n_in_cells = n_unmasked if args.matrix[2] in "IX" else ncol      # input cells regridded per field

    # ---- field batches resident in HBM -----------------------------------------------------------
    xbytes = 8 * nf * ncol
    nbuf = 1 if args.warm else min(64, max(2, -(-(512 << 20) // xbytes)))      # (I-row matrices: X is KB-sized, Y is what streams)
    # the cold check rotates over up to 28 batches of an A-row matrix's (bandwidth-sized) input
    cold_sizes = [1, 7, 14, 28] if (not args.no_extras and not use_dist and not args.warm and (8 << 20) <= xbytes and 28 * xbytes <= (4 << 30)) else []
    nalloc = max([nbuf] + cold_sizes)
    x_host = syn.fields(nf, ncol, seed=syn.SEED + rank)
    x0 = torch.from_numpy(x_host).to(dev)
    X = [x0 if b == 0 else x0 + 1e-3 * b for b in range(nalloc)]
    depth = max(1, min(32, args.queue_depth))
    ldy = (nrow + 63) // 64 * 64 if nrow >= 4096 else nrow       # 512-byte planes for the I-row matrices (whole-line stores)
    if 8 * nf * ldy > (64 << 20):
        depth = min(depth, 4)                                    # results of ~1 GB each (I-row matrices at 1 km): a short queue is plenty
    Y = [torch.zeros((nf, ldy), dtype=torch.float64, device=dev) for _ in range(2 * depth)]

    L = _capi.lib()
    fn = L.ibh_weighted_apply_device
    fn_many = L.ibh_weighted_apply_many_device
    compute = torch.cuda.Stream(device=dev)
    cs = C.c_void_p(compute.cuda_stream)
    xp = [x.data_ptr() for x in X]
    yp = [y.data_ptr() for y in Y]
    nan = float("nan")
    W.prepare(nf, depth)          # apply structures + scratch up front: the timed applies only enqueue work
    # results of `depth` applies share ONE all-gather while they are small (a [64, 122] AvI result is 62 KB:
    # a collective of that size is latency-bound); a result of megabytes (I-row matrices) is gathered per apply
    gsteps = depth if 8 * nf * ldy < (4 << 20) else 1
    # The gathers go through torch.distributed's all_gather_into_tensor by default; ICEBIN_BENCH_SHARDED=cabi takes the library's
    # own RCCL calls (ibh_comm, ibh_weighted_apply_*_sharded_device: direct peer-to-peer exchange) -- the path a C++ host takes
    # (north_star: "host code stays C++"); `gather_via` in the JSON line says which one ran.  The communicator's bootstrap is
    # decided collectively (every rank probes RCCL, the flags are all-reduced before the id is broadcast); if the communicator
    # cannot be created, or its first gathered result is not bitwise the locally recomputed shards, EVERY rank falls back to
    # the torch path together (all-reduced again) and `gather_via_fallback` says why.
    sharded = None
    sharded_via = None
    sharded_fallback = None
    if use_dist:
        # (torch.distributed stays the default until a recorded run with more than one rank over RCCL has passed the
        # bitwise gather check through the library's own exchange; ICEBIN_BENCH_SHARDED=cabi asks for that path)
        want_cabi = os.environ.get("ICEBIN_BENCH_SHARDED", "torch") == "cabi" and backend == "nccl"
        if want_cabi:
            from icebin_amd.distributed import CabiFieldShardedApply, Communicator
            why = None
            try:
                sharded = CabiFieldShardedApply(W, nf_total, Communicator(world, rank, rccl=True), dev, steps_per_gather=gsteps)
            except Exception as e:      # noqa: BLE001 -- e.g. no librccl to dlopen
                why = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
            ok_t = torch.tensor([0 if why else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
            if ok_t.item() == 1:           # one gathered apply against the local recomputation, before anything is timed
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    x_probe = torch.from_numpy(syn.fields(nf, ncol, seed=syn.SEED + rank)).to(dev)
                    g_, slot_ = sharded.apply_ptr(x_probe.data_ptr(), ncol)
                    sharded.flush(); sharded.wait()
                    torch.cuda.synchronize(dev)
                    mine = sharded.result(g_, slot_)[rank * nf:(rank + 1) * nf] if not strong else None
                    if mine is not None:
                        ref = W.apply_device(x_probe, fill=nan, force_conservation=False)
                        torch.cuda.synchronize(dev)
                        same = torch.equal(mine.contiguous().view(torch.int64), ref[:, :nrow].contiguous().view(torch.int64))
                        ok_t = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev)
                        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
                        if ok_t.item() != 1:
                            why = "the first exchange through ibh_comm did not reproduce the local shard bitwise"
            else:
                why = why or "another rank could not create its ibh_comm communicator"
            if why is None:
                sharded_via = "C-ABI: ibh_weighted_apply_many_sharded_device (grouped ncclSend/ncclRecv to every peer)"
            else:
                sharded, sharded_fallback = None, why
        if sharded is None:
            sharded = FieldShardedApply(W, nf_total, None, dev, steps_per_gather=gsteps)
            sharded_via = "torch.distributed all_gather_into_tensor (%s)" % backend

    # ---- the assembly shared by the ranks (ibh_regrid_matrices_matrix_d_sharded; BASELINE config 5: "assembly + apply, 8 x MI355X")
    asm_sharded = None
    if use_dist and sharded_via and sharded_via.startswith("C-ABI"):
        try:
            ts = []
            Ws = None
            for _ in range(3):
                torch.cuda.synchronize(dev)
                dist.barrier()
                t0 = time.perf_counter()
                Ws = rm.matrix_d_sharded(sharded.comm, args.matrix, scale=True, correctA=True)
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t0)
            same = (Ws.nrow_d, Ws.ncol_d, Ws.nnz) == (nrow, ncol, nnz)
            if same:
                for a, b in zip(Ws.csr_dense() + (Ws.wM, Ws.Mw), W.csr_dense() + (W.wM, W.Mw)):
                    same = same and bool(np.array_equal(a.view(np.uint64) if a.dtype == np.float64 else a, b.view(np.uint64) if b.dtype == np.float64 else b))
            tt = torch.tensor([min(ts), 0.0 if same else 1.0, 1.0 if Ws.built_sharded() else 0.0], dtype=torch.float64, device=dev)
            tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tmin = tt.clone(); dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
            asm_sharded = {"ms_max_over_ranks": float(tmax[0]) * 1e3, "ms_min_over_ranks": float(tmin[0]) * 1e3, "ms_this_rank": min(ts) * 1e3,
                           "shared_by_the_ranks": bool(tmin[2] == 1.0), "bitwise_equal_to_the_replicated_build_on_every_rank": bool(tmax[1] == 0.0),
                           "what": "ibh_regrid_matrices_matrix_d_sharded: best of 3 collective builds, barrier before each, wall time until this rank holds the whole matrix"}
            del Ws
        except Exception as e:      # noqa: BLE001
            asm_sharded = {"error": "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")}

    def plan(i0, n, depth_=None, nbuf_=None):
        """Launch plan for steps i0 .. i0+n-1: (count, X pointers, X pointer table, Y pointer table),
        built before the timed region so the loop only makes the C calls."""
        d, nb = depth_ or depth, nbuf_ or nbuf
        out = []
        i, k = i0, 0
        nl = -(-n // d) if n else 0                           # launches, of equal size (20 steps at depth 16: 10 + 10)
        per = -(-n // nl) if nl else 0
        while i < i0 + n:
            m = min(per, i0 + n - i)
            xs = [xp[(i + j) % nb] for j in range(m)]
            xa = (C.c_void_p * m)(*xs)
            ya = (C.c_void_p * m)(*[yp[((k & 1) * depth + j)] for j in range(m)])
            out.append((m, xs, xa, ya))
            i += m
            k += 1
        return out

    def new_event():
        e = C.c_void_p()
        _capi.check(L.ibh_event_create(C.byref(e)))
        return e

    def new_pairs(launches):
        return [(new_event(), new_event()) for _ in launches]

    def read_pairs(pairs):
        """elapsed ms of every (start, stop) pair; the events are destroyed"""
        out = []
        for (a, b) in pairs:
            ms = C.c_float()
            _capi.check(L.ibh_event_elapsed_ms(a, b, C.byref(ms)))
            out.append(ms.value)
            L.ibh_event_destroy(a); L.ibh_event_destroy(b)
        return out

    def run(launches, gather=True, kev=None):
        """kev: one (start, stop) HIP event pair per launch, attached to the SpMM kernel's own dispatch
        (ibh_set_launch_events): the kernel's duration without submission gaps."""
        for li, (m, xs, xa, ya) in enumerate(launches):
            single_call = not (use_dist and gather) or m <= sharded.G
            if kev is not None and single_call:
                pair = kev[li]                      # created before the timed region (hipEventCreate is not free)
                L.ibh_set_launch_events(pair[0], pair[1])
            if use_dist and gather:       # field-sharded SpMM + grouped all-gather, icebin_amd/distributed.py
                if m <= sharded.G:
                    sharded.apply_many_ptr(xs, ncol)
                    sharded.flush()           # one all-gather per launch (a short last launch too)
                else:
                    for xptr in xs:
                        sharded.apply_ptr(xptr, ncol)
                continue
            if m == 1:
                rc = fn(W._h, xa[0], nf, ncol, ya[0], ldy, nan, 0, cs)
            else:
                rc = fn_many(W._h, m, xa, nf, ncol, ya, ldy, nan, 0, cs)
            if rc != 0:
                _capi.check(rc)

    def sync_all():
        if use_dist:
            sharded.flush()
            sharded.wait()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(launches, gather=True, kev=None):
        """EXACTLY the steps of `launches`, barrier + synchronize on both sides; wall time = max over ranks"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(compute)
        run(launches, gather, kev)
        e1.record(compute)
        sync_all()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, e0.elapsed_time(e1)

    R = max(1, args.repeats)
    warm_plan, timed_plan = plan(0, args.warmup), plan(args.warmup, args.steps)
    nlaunch = len(timed_plan)
    B = spmm_bytes(nnz, nrow, ncol, nf)
    csr_bytes = 12 * nnz + 4 * (nrow + 1)
    single_call_launches = (not use_dist or all(m <= gsteps for (m, _, _, _) in timed_plan)) and not args.no_kernel_events
    regions = []            # (wall s, region events ms, [kernel ms per launch]) per repetition
    spmm_only = None
    extras = {}
    with torch.cuda.stream(compute):
        run(warm_plan)
        sync_all()
        if use_dist:        # the same K steps without the collectives (diagnostic; not the reported value)
            so = []
            for _ in range(min(R, 5)):
                dt_s, _ = timed(timed_plan, gather=False)
                so.append(dt_s / max(args.steps, 1) * 1e3)
            spmm_only = median(so)
        for rep in range(R):
            kev = new_pairs(timed_plan) if single_call_launches else None
            dt, region_ms = timed(timed_plan, True, kev)
            regions.append((dt, region_ms, read_pairs(kev) if kev else []))
        if not args.no_extras and not use_dist and args.steps > 0:
            # one apply() per launch -- the reference's own call shape (merge_topo.cpp:65, icebin22m.cpp:153)
            p1 = plan(0, min(args.steps, 16), depth_=1)
            run(p1); sync_all()
            ks, ws = [], []
            for rep in range(R):
                kev = new_pairs(p1)
                dt, _ = timed(p1, True, kev)
                ks += read_pairs(kev)
                ws.append(dt / len(p1))
            extras["single_launch"] = (ks, ws)
            # cold or not: the headline launch shape over 1 / 7 / 14 / 28 rotating field batches
            cc = {}
            for nb in cold_sizes:
                pc = plan(0, 4 * depth, nbuf_=nb)
                run(pc); sync_all()
                ks = []
                for rep in range(5):
                    kev = new_pairs(pc)
                    timed(pc, True, kev)
                    ks += [ms / m for ms, (m, _, _, _) in zip(read_pairs(kev), pc)]
                cc[nb] = ks
            extras["cold_check"] = cc
            # what this box's HBM delivers to plain streams ("also report vs a measured device copy / triad bandwidth"):
            # torch's own kernels over 2 GiB, far beyond the 256 MB Infinity Cache -- a read-only sum, a write-only fill, a copy
            try:
                n64 = (2 << 30) // 8
                sa = torch.empty(n64, dtype=torch.float64, device=dev).fill_(1.0)
                sb_ = torch.empty(n64, dtype=torch.float64, device=dev)
                def stream_rate(fn, nbytes):
                    fn(); torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
                    return 3 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9
                extras["streams"] = {"read_GBs": stream_rate(lambda: sa.sum(), 8 * n64), "write_GBs": stream_rate(lambda: sb_.fill_(2.0), 8 * n64),
                                     "copy_GBs": stream_rate(lambda: sb_.copy_(sa), 16 * n64)}
                del sa, sb_
            except Exception as e:      # noqa: BLE001  (a side measurement: never fails the bench)
                extras["streams"] = {"error": str(e)[:120]}

    # ---- cross-rank correctness of the gathered results (N > 1 path; also at N = 1 under torchrun) -----------------
    gather_check = None
    ranks_seen = None
    if use_dist:
        with torch.cuda.stream(compute):
            g_, slot_ = sharded.apply_ptr(xp[0], ncol)
            sharded.flush(); sharded.wait()
            torch.cuda.synchronize(dev)
            res = sharded.result(g_, slot_)                       # [nf_total, nrow] as every rank sees it
            ok = True
            for r in sorted({rank, (rank + 1) % world}):
                r0, r1 = field_shard(nf_total, world, r) if strong else (r * nf, (r + 1) * nf)
                xr = x0 if r == rank else torch.from_numpy(syn.fields(r1 - r0, ncol, seed=syn.SEED + r)).to(dev)
                yr = torch.zeros((r1 - r0, ldy), dtype=torch.float64, device=dev)
                _capi.check(fn(W._h, C.c_void_p(xr.data_ptr()), r1 - r0, ncol, C.c_void_p(yr.data_ptr()), ldy, nan, 0, cs))
                torch.cuda.synchronize(dev)
                a, b = res[r0:r1].contiguous().view(torch.int64), yr[:, :nrow].contiguous().view(torch.int64)
                ok = ok and bool(torch.equal(a, b))               # same kernel, same field count: bitwise
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        gather_check = {"pass": bool(t.item() == 1), "ranks": world, "blocks_checked_per_rank": len({rank, (rank + 1) % world}),
                        "how": "every rank recomputes its own and its right neighbour's shard locally and compares the gathered blocks bitwise"}
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "device": dev_index, "host": socket.gethostname()})
        ranks_seen = seen

    result = None
    if rank == 0:
        walls = [r[0] for r in regions]
        dt = median(walls)
        region_ms = median([r[1] for r in regions])
        # per-launch kernel durations of all repetitions; full-size launches carry the roofline figure
        per = timed_plan[0][0] if timed_plan else 0
        all_ms = [ms for r in regions for ms in r[2]]
        full_ms = [ms for r in regions for ms, (m, _, _, _) in zip(r[2], timed_plan) if m == per]
        per_apply_us = [ms * 1e3 / m for r in regions for ms, (m, _, _, _) in zip(r[2], timed_plan)]
        have_k = len(full_ms) > 0
        B_launch = per * (B - csr_bytes) + csr_bytes              # CSR once per LAUNCH
        if have_k:
            launch_us = median(full_ms) * 1e3
            kernel_us = launch_us / per
            achieved = B_launch / (launch_us * 1e-6) / 1e9
        else:                                                     # no kernel-attached events (sharded per-apply path): region events
            launch_us = region_ms * 1e3 / max(nlaunch, 1)
            kernel_us = region_ms * 1e3 / max(args.steps, 1)
            achieved = (args.steps * (B - csr_bytes) + nlaunch * csr_bytes) / (region_ms * 1e-3) / 1e9
        cells = n_in_cells * nf_total * args.steps
        par = "1 GPU"
        if use_dist:
            par = ("field-shard x%d (%s: %d fields/rank) + all-gather every %d steps" %
                   (world, "strong" if strong else "weak", nf, gsteps))
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args, W.last_kernel(), depth, per, W.last_launch()),
                "launched": W.last_launch(),
                "kernel_us": kernel_us, "algorithmic_bytes": B,
                "launches": nlaunch, "steps_per_launch": per, "launch_us": launch_us,
                "statistic": "median over %d launches (%d repetitions of the timed region x %d launches)" % (len(full_ms), R, nlaunch) if have_k
                             else "median over %d repetitions of the timed region" % R,
                "repeats": R,
                "timing": "HIP events attached to each SpMM dispatch of the timed regions (kernel start -> end)" if have_k else "HIP events around the timed region on the launch stream",
                "region_us_per_step": region_ms * 1e3 / max(args.steps, 1),
                "algorithmic_bytes_per_launch": B_launch,
                "traffic_source": "profiles/*_pmc_traffic.json: separate rocprofv3 --pmc passes of this command, per launch; not measured in this run; null when that profile was taken on another kernel instantiation than `launched`"}
        if have_k:
            roof.update({"launch_us_min": min(full_ms) * 1e3, "launch_us_max": max(full_ms) * 1e3, "launch_samples": len(full_ms),
                         "kernel_us_min": min(per_apply_us), "kernel_us_max": max(per_apply_us),
                         "frac_min": B_launch / (max(full_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_max": B_launch / (min(full_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS})
        if "single_launch" in extras:
            ks, ws = extras["single_launch"]
            k_us = median(ks) * 1e3
            roof["single_launch"] = {"kernel_us": k_us, "frac": B / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                     "achieved": B / (k_us * 1e-6) / 1e9, "kernel_us_min": min(ks) * 1e3, "kernel_us_max": max(ks) * 1e3,
                                     "samples": len(ks), "wall_us_per_apply": median(ws) * 1e6, "kernel": W.last_kernel(),
                                     "what": "ONE apply() per launch (ibh_weighted_apply_device), same matrix, same rotating batches"}
        if extras.get("streams"):
            st_ = dict(extras["streams"])
            if "read_GBs" in st_:
                st_["achieved_over_measured_read"] = roof["achieved"] / st_["read_GBs"] if roof.get("achieved") else None
                st_["what"] = "plain streams on this box, torch kernels over 2 GiB (read-only sum / write-only fill / copy, bytes moved per second): the practical ceilings beside the 8 TB/s spec peak `frac` is priced against"
            roof["measured_streams"] = st_
        if extras.get("cold_check"):
            roof["cold_check"] = {"kernel_us_per_apply_by_rotating_batches": {str(nb): median(v) * 1e3 for nb, v in extras["cold_check"].items()},
                                  "batch_MB": xbytes / 1e6, "infinity_cache_MB": 256,
                                  "what": "the headline launch shape (%d applies per launch) over 1 (Infinity-Cache-resident) / 7 / 14 / 28 rotating field batches" % depth}
        result = {
            "metric": "regridded cells/sec (%s, %s, %d fields%s)" % (args.matrix, args.config, nf_total if strong else nf, "" if strong else "/GPU"),
            "value": cells / dt, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": {"R": R, "statistic": "median", "ms_per_step_min": min(walls) / args.steps * 1e3,
                        "ms_per_step_max": max(walls) / args.steps * 1e3, "ms_per_step_first": walls[0] / args.steps * 1e3},
            "config": {"workload": "searise 5 km Greenland -> ModelE 2x2.5 %s, %d fields" % (args.matrix, nf)
                       if args.config == "g5" else "%s %s %d fields" % (args.config, args.matrix, nf),
                       "nI": grids["nI"], "nX": int(len(grids["ex_area"])), "unmasked_cells": n_unmasked,
                       "mask": "all cells unmasked" if args.all_unmasked else "dome, ~45 % unmasked",
                       "nrow_d": nrow, "ncol_d": ncol, "nnz": nnz, "fields_per_gpu": nf, "fields_total": nf_total,
                       "field_batches": nbuf,
                       "cache": "warm" if args.warm else "cold (rotating batches > Infinity Cache)",
                       "kernel": W.last_kernel(), "queue_depth": depth, "parallelism": par},
            "roofline": roof,
            "assembly": {"first_call_ms": t_asm_first * 1e3, "steady_ms": t_asm * 1e3,
                         "algorithmic_bytes": asm_bytes(len(grids["ex_area"]), grids["nI"], nnz, nrow, ncol),
                         "GBps": asm_bytes(len(grids["ex_area"]), grids["nI"], nnz, nrow, ncol) / t_asm / 1e9},
        }
        if asm_sharded is not None:
            result["assembly"]["sharded"] = asm_sharded
        if use_dist:
            result["spmm_only_ms"] = spmm_only
            result["spmm_plus_gather_ms"] = dt / args.steps * 1e3
            result["gather_bytes_per_rank_per_step"] = 8 * nf * ((nrow + 63) // 64 * 64)
            result["gather_via"] = sharded_via
            if sharded_fallback:
                result["gather_via_fallback"] = sharded_fallback
            result["gather_check"] = gather_check
            result["ranks_seen"] = ranks_seen
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(grids, em, args, x_host, n_in_cells)
        if args.variants and not use_dist and not args.all_unmasked and not args.warm and args.config == "g5":
            # the same launch on the peak-size variant of the same grids (every ice cell unmasked:
            # X has all nI = 168 861 columns, the size BASELINE.md's byte table assumes); informational
            result["variants"] = {"all_unmasked": variant_all_unmasked(torch, icebin_amd, _capi, syn, grids, args, dev)}
        if use_dist:
            y = sharded.result(0, 0).cpu().numpy()
        else:
            y = torch.stack(Y[:min(2 * depth, args.steps, 4)])[:, :, :nrow].cpu().numpy() if args.steps > 0 else None
        result["finite_output"] = bool(y is not None and np.isfinite(y).all())
        print(json.dumps(result), flush=True)
    failed = bool(gather_check is not None and not gather_check["pass"])
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


def dry_run(args):
    """The N > 1 choreography of this file WITHOUT a GPU: torchrun rendezvous (gloo), field sharding, grouped all-gathers
    through the real FieldShardedApply, the cross-rank gather check, ranks_seen -- with a small random CSR applied by
    scipy on CPU tensors.  A rehearsal for tests/test_distributed_gloo.py; it measures nothing and says so."""
    import scipy.sparse
    import torch
    import torch.distributed as dist
    from icebin_amd.distributed import FieldShardedApply, HostOps, field_shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    use_dist = "RANK" in os.environ
    if not use_dist:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo")
    rng = np.random.default_rng(7)                      # the same matrix on every rank (replicated CSR)
    nrow, ncol = 37, 501
    M = scipy.sparse.random(nrow, ncol, density=0.05, random_state=rng, format="csr")
    strong = args.fields_total > 0
    nf_total = args.fields_total if strong else args.fields * world
    f0, f1 = field_shard(nf_total, world, rank)
    nf = f1 - f0

    def fields_of(r):
        r0, r1 = field_shard(nf_total, world, r)
        return np.random.default_rng(100 + r).standard_normal((r1 - r0, ncol))

    class Mat:
        nrow_d, ncol_d = nrow, ncol

    def local_apply(x, ldx, y, fill, stream_h):
        y[:, :nrow] = torch.from_numpy((M @ x.numpy().T).T)

    depth = max(1, min(32, args.queue_depth))
    sh = FieldShardedApply(Mat(), nf_total, None, ops=HostOps(), steps_per_gather=depth, local_apply=local_apply,
                           local_apply_many=None, nrow=nrow, ncol=ncol)
    x = torch.from_numpy(fields_of(rank))
    t0 = time.perf_counter()
    last = None
    for i in range(args.warmup + args.steps):
        last = sh.apply(x)
    sh.flush(); sh.wait()
    dist.barrier()
    dt = time.perf_counter() - t0
    # the cross-rank check of the real run: every rank recomputes its own and its right neighbour's shard
    g_, slot_ = last
    res = sh.result(g_, slot_).numpy()
    ok = True
    for r in sorted({rank, (rank + 1) % world}):
        r0, r1 = field_shard(nf_total, world, r)
        ok = ok and np.array_equal(res[r0:r1], (M @ fields_of(r).T).T)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    seen = [None] * world
    dist.all_gather_object(seen, {"rank": rank, "device": None, "host": socket.gethostname()})
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN of bench.py's N>1 path on CPU (gloo, scipy apply) -- not a measurement", "dry_run": True,
                          "value": None, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / max(args.steps + args.warmup, 1) * 1e3, "scaling": "strong" if strong else "weak",
                          "fields_per_rank": nf, "fields_total": nf_total,
                          "gather_check": {"pass": bool(t.item() == 1), "ranks": world}, "ranks_seen": seen}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if t.item() == 1 else 3


def variant_all_unmasked(torch, icebin_amd, _capi, syn, grids, args, dev, steps=480):
    em = syn.dome_elevmask(grids, all_unmasked=True)
    W = icebin_amd.from_synthetic(grids).regrid_matrices("greenland", em, scale=True, correctA=True).matrix(args.matrix)
    nf, nrow, ncol, nnz = args.fields, W.nrow_d, W.ncol_d, W.nnz
    depth = max(1, min(32, args.queue_depth))
    nbuf = max(2, -(-(512 << 20) // (8 * nf * ncol)))
    x0 = torch.from_numpy(syn.fields(nf, ncol)).to(dev)
    X = [x0 + 1e-3 * b for b in range(nbuf)]
    Y = [torch.empty((nf, nrow), dtype=torch.float64, device=dev) for _ in range(depth)]
    fn = _capi.lib().ibh_weighted_apply_many_device
    st = torch.cuda.current_stream(dev)
    cs = C.c_void_p(st.cuda_stream)
    ya = (C.c_void_p * depth)(*[y.data_ptr() for y in Y])
    plans = [(C.c_void_p * depth)(*[X[(i + j) % nbuf].data_ptr() for j in range(depth)]) for i in range(0, steps, depth)]

    def run():
        for xa in plans:
            fn(W._h, depth, xa, nf, ncol, ya, nrow, float("nan"), 0, cs)
    run()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    run()
    e1.record(st)
    torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) / (len(plans) * depth) * 1e3
    B = spmm_bytes(nnz, nrow, ncol, nf)
    return {"nrow_d": nrow, "ncol_d": ncol, "nnz": nnz, "algorithmic_bytes": B, "kernel_us": us, "queue_depth": depth,
            "achieved_GBps": B / us / 1e3, "frac": B / us / 1e3 / HBM_PEAK_GBS, "cells_per_s": ncol * nf / us * 1e6}


def pmc_traffic(args, kernel, depth, applies_per_launch, launched=None):
    """HBM bytes per LAUNCH of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs of this very command,
    FETCH doubled per the gfx950 correction), scaled from the profiled launch (32 applies) to this
    run's applies per launch.  None when no profile matches this workload."""
    import glob
    key = "spmm_%s_%s_%s_%df_d%d%s" % (kernel, args.config, args.matrix, args.fields, depth, "_allunmasked" if args.all_unmasked else "")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if key in d and not args.warm:
            # the profile is quoted only for the kernel instantiation it was taken on (a retuned launch makes it stale: null then)
            if launched and d[key].get("kernel") and d[key]["kernel"] != launched:
                return None
            return d[key]["traffic_bytes"] / d[key].get("applies_per_launch", 1) * applies_per_launch
    return None


def cpu_baseline(grids, em, args, x_host, n_in_cells):
    """The oracle (reference-faithful port: Eigen's ColMajor sparse x dense loop, 1 thread, as the
    reference is single-threaded, CMakeLists.txt:135-137) timed on this box's host cores on the
    same matrix and the same 64 fields; also a row-partitioned CSR SpMM on all cores."""
    from oracle import oracle as orc
    rg = orc.Regridder(grids)
    o = rg.matrix_d(args.matrix, em, scale=True, correctA=True)
    nf = x_host.shape[0]
    t1 = o.time_apply(x_host, reps=1, threads=1)
    reps = max(3, int(8.0 / max(t1, 1e-6)))
    reps = min(reps, 20000)
    t1 = min(o.time_apply(x_host, reps=max(1, reps // 4), threads=1) for _ in range(4))
    ncpu = min(len(os.sched_getaffinity(0)), 16)      # the GPU box's CPU share for one GPU
    tn = min(o.time_apply(x_host, reps=max(1, reps // 4), threads=ncpu) for _ in range(4))
    return {"value": n_in_cells * nf / t1, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "same matrix and %d fields, %d applies, best of 4 batches" % (nf, reps),
            "ms_per_apply": t1 * 1e3,
            "all_cores": {"value": n_in_cells * nf / tn, "cores": ncpu, "ms_per_apply": tn * 1e3}}


if __name__ == "__main__":
    main()
