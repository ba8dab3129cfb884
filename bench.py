#!/usr/bin/env python3
"""bench.py -- headline benchmark of the regrid hot path on MI355X.

Metric (BASELINE.json): regridded cells/sec + achieved HBM GB/s, AvI 5 km -> 2x2.5 deg, 64 fields.
A "step" is one Weighted::apply of the AvI matrix to one batch of 64 synthetic fields that
are already resident in HBM.  Steps cycle through enough distinct field batches (>= 512 MiB in
total) that no batch can be served from the 256 MiB Infinity Cache: every step streams its X
from HBM ("cold" numbers).  Steps are submitted --queue-depth at a time through
ibh_weighted_apply_many_device (ONE launch serves up to 32 independent 64-field applies: a single
40 MB apply is latency-sized on this chip, launch + dependent loads are a third of its 11 us;
--queue-depth 1 is the one-launch-per-apply figure).  Every step still reads its own 39 MB of X
from HBM and writes its own Y; the CSR is counted ONCE per launch in the roofline bytes.

N GPUs (torchrun, one rank per GPU), two modes:
  * default (weak scaling): every rank applies the replicated CSR to its own 64 fields of a
    64*N-field regrid;
  * --fields-total F (strong scaling, BASELINE config 4: `--config g1 --fields-total 64`, AvI and
    `--matrix IvA`): ONE F-field regrid, F/N fields per rank.
Either way the SpMM needs no communication; RCCL all-gathers reassemble the [fields, nrow] results on
every rank, one collective per group of --queue-depth applies, issued on a second stream so it overlaps
the following SpMMs.  spmm_only_ms (no collectives) and spmm_plus_gather_ms are reported separately.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1920)
    ap.add_argument("--warmup", type=int, default=160)
    ap.add_argument("--config", default="g5", help="synthetic grid config (icebin_amd/synthetic.py)")
    ap.add_argument("--matrix", default="AvI")
    ap.add_argument("--fields", type=int, default=64, help="fields per GPU (weak scaling)")
    ap.add_argument("--fields-total", type=int, default=0, help="strong scaling: ONE regrid of this many fields, sharded over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variants", action="store_true", help="also time the peak-size (all-unmasked) variant; informational")
    ap.add_argument("--all-unmasked", action="store_true", help="every ice cell carries ice (peak-size variant, SURVEY.md 8d)")
    ap.add_argument("--warm", action="store_true", help="reuse ONE field batch (Infinity-Cache-resident numbers)")
    ap.add_argument("--queue-depth", type=int, default=32, help="applies submitted per launch (1..32); N>1: also applies per all-gather")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT", help="ibh_set_tuning override (experiments)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.gpus > 1:
        assert world == args.gpus, "launch with torchrun --nproc-per-node %d" % args.gpus
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
    t0 = time.perf_counter()
    W2 = rm.matrix(args.matrix)
    torch.cuda.synchronize()
    t_asm = time.perf_counter() - t0
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
    x_host = syn.fields(nf, ncol, seed=syn.SEED + rank)
    x0 = torch.from_numpy(x_host).to(dev)
    X = [x0 if b == 0 else x0 + 1e-3 * b for b in range(nbuf)]
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
    W.reserve(nf)
    # results of `depth` applies share ONE all-gather while they are small (a [64, 122] AvI result is 62 KB:
    # a collective of that size is latency-bound); a result of megabytes (I-row matrices) is gathered per apply
    gsteps = depth if 8 * nf * ldy < (4 << 20) else 1
    sharded = FieldShardedApply(W, nf_total, None, dev, steps_per_gather=gsteps) if use_dist else None

    def plan(i0, n):
        """Launch plan for steps i0 .. i0+n-1: (count, X pointers, X pointer table, Y pointer table),
        built before the timed region so the loop only makes the C calls."""
        out = []
        i, k = i0, 0
        nl = -(-n // depth) if n else 0                       # launches, of equal size (20 steps at depth 16: 10 + 10)
        per = -(-n // nl) if nl else 0
        while i < i0 + n:
            m = min(per, i0 + n - i)
            xs = [xp[(i + j) % nbuf] for j in range(m)]
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

    def run(launches, gather=True, kev=None):
        """kev: list that receives one (start, stop) HIP event pair per launch, attached to the SpMM
        kernel's own dispatch (ibh_set_launch_events): the kernel's duration without submission gaps."""
        for li, (m, xs, xa, ya) in enumerate(launches):
            if kev is not None:
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

    kev = None          # set below: one pre-created (start, stop) event pair per timed launch (N = 1)

    def timed(launches, gather=True):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(compute)
        run(launches, gather, kev if gather else None)
        e1.record(compute)
        sync_all()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, e0.elapsed_time(e1)

    warm_plan, timed_plan = plan(0, args.warmup), plan(args.warmup, args.steps)
    if not use_dist:
        kev = [(new_event(), new_event()) for _ in timed_plan]
    with torch.cuda.stream(compute):
        run(warm_plan)
        sync_all()
        spmm_only = None
        if use_dist:        # the same K steps without the collectives (diagnostic; not the reported value)
            dt_s, _ = timed(timed_plan, gather=False)
            spmm_only = dt_s / max(args.steps, 1) * 1e3
        dt, region_ms = timed(timed_plan)                     # EXACTLY K steps, barrier + synchronize on both sides
    nlaunch = len(timed_plan)
    # the SpMM kernel's own launch durations (HIP events attached to each dispatch of the timed region);
    # region_ms (events around the whole region on the launch stream) also holds the host's submission
    # latency in front of the first launch and the gaps between launches
    kern_ms = None
    if kev:
        kern_ms = 0.0
        for (a, b) in kev:
            ms = C.c_float()
            _capi.check(L.ibh_event_elapsed_ms(a, b, C.byref(ms)))
            kern_ms += ms.value
            L.ibh_event_destroy(a); L.ibh_event_destroy(b)
    meas_ms = kern_ms if kern_ms else region_ms
    kernel_ms = meas_ms / max(args.steps, 1)                  # per step (= per 64-field apply)

    result = None
    if rank == 0:
        cells = n_in_cells * nf_total * args.steps
        B = spmm_bytes(nnz, nrow, ncol, nf)
        csr_bytes = 12 * nnz + 4 * (nrow + 1)
        # algorithmic bytes of the timed region: every step's X and Y once, the CSR once per LAUNCH
        B_region = args.steps * (B - csr_bytes) + nlaunch * csr_bytes
        achieved = B_region / (meas_ms * 1e-3) / 1e9
        par = "1 GPU"
        if use_dist:
            par = ("field-shard x%d (%s: %d fields/rank) + all-gather every %d steps" %
                   (world, "strong" if strong else "weak", nf, gsteps))
        result = {
            "metric": "regridded cells/sec (%s, %s, %d fields%s)" % (args.matrix, args.config, nf_total if strong else nf, "" if strong else "/GPU"),
            "value": cells / dt, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "searise 5 km Greenland -> ModelE 2x2.5 %s, %d fields" % (args.matrix, nf)
                       if args.config == "g5" else "%s %s %d fields" % (args.config, args.matrix, nf),
                       "nI": grids["nI"], "nX": int(len(grids["ex_area"])), "unmasked_cells": n_unmasked,
                       "mask": "all cells unmasked" if args.all_unmasked else "dome, ~45 % unmasked",
                       "nrow_d": nrow, "ncol_d": ncol, "nnz": nnz, "fields_per_gpu": nf, "fields_total": nf_total,
                       "field_batches": nbuf,
                       "cache": "warm" if args.warm else "cold (rotating batches > Infinity Cache)",
                       "kernel": W.last_kernel(), "queue_depth": depth, "parallelism": par},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args, W.last_kernel(), depth, args.steps / max(nlaunch, 1)),
                         "kernel_us": kernel_ms * 1e3, "algorithmic_bytes": B,
                         "launches": nlaunch, "steps_per_launch": args.steps / max(nlaunch, 1),
                         "launch_us": meas_ms * 1e3 / max(nlaunch, 1),
                         "timing": "HIP events attached to each SpMM dispatch of the timed region (kernel start -> end)" if kern_ms else "HIP events around the timed region on the launch stream",
                         "region_us_per_step": region_ms * 1e3 / max(args.steps, 1),
                         "algorithmic_bytes_per_launch": B_region / max(nlaunch, 1),
                         "traffic_source": "profiles/*_pmc_traffic.json: separate rocprofv3 --pmc passes of this command, per launch; not measured in this run"},
            "assembly": {"first_call_ms": t_asm_first * 1e3, "steady_ms": t_asm * 1e3,
                         "algorithmic_bytes": asm_bytes(len(grids["ex_area"]), grids["nI"], nnz, nrow, ncol),
                         "GBps": asm_bytes(len(grids["ex_area"]), grids["nI"], nnz, nrow, ncol) / t_asm / 1e9},
        }
        if use_dist:
            result["spmm_only_ms"] = spmm_only
            result["spmm_plus_gather_ms"] = dt / args.steps * 1e3
            result["gather_bytes_per_rank_per_step"] = 8 * nf * ((nrow + 63) // 64 * 64)
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
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


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


def pmc_traffic(args, kernel, depth, applies_per_launch):
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
