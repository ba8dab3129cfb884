"""Field-column sharding of a multi-field regrid across the GPUs of one node.

Every field (one row of A_b(nvar, n), icebin22m.cpp:142) is an independent SpMV with the same
matrix, so rank g owns the contiguous fields [f0, f1) of the field-major arrays: its shard is
contiguous in memory, the CSR is replicated, and the SpMM needs no communication.  One RCCL
all-gather (torch.distributed, backend "nccl" on ROCm; "gloo" in the CPU tests) reassembles
the [nf_total, nrow] result on every rank.  The reference has no counterpart: it gathers
everything to MPI rank 0 and regrids there (modele/GCMCoupler_ModelE.cpp:764-792).

Assembly (config 5: several ice sheets): sheets are independent regridders
(GCMRegridder.hpp:249, regrid_matrices(sheet_index, ...)), so `sheet_partition` deals whole sheets
to disjoint rank sets by size; the ranks of one set build the same matrices redundantly (a build is
1-6 ms, cheaper than broadcasting a 0.5 GB CSR over one xGMI link) and shard the sheet's fields.
"""
import torch
import torch.distributed as dist


def field_shard(nf_total, world, rank):
    """Contiguous block partition of nf_total fields: the first (nf_total % world) ranks get one more."""
    base, rem = divmod(nf_total, world)
    f0 = rank * base + min(rank, rem)
    return f0, f0 + base + (1 if rank < rem else 0)


def sheet_partition(sheet_sizes, world):
    """Deal ice sheets to disjoint, contiguous rank sets in proportion to their size (exchange cells).

    sheet_sizes: {name: nX}.  Returns {name: (first_rank, n_ranks)}.  With fewer ranks than sheets
    the sheets are dealt round-robin to single ranks (largest first).  Deterministic on every rank:
    no communication."""
    names = sorted(sheet_sizes, key=lambda k: (-sheet_sizes[k], k))
    if world <= len(names):
        load = [0] * world
        out = {}
        for n in names:
            r = min(range(world), key=lambda k: (load[k], k))
            out[n] = (r, 1)
            load[r] += sheet_sizes[n]
        return out
    total = float(sum(sheet_sizes.values())) or 1.0
    quota = {n: world * sheet_sizes[n] / total for n in names}
    share = {n: max(1, int(quota[n])) for n in names}
    while sum(share.values()) < world:          # largest remainder first
        n = max(names, key=lambda k: (quota[k] - share[k], sheet_sizes[k]))
        share[n] += 1
    while sum(share.values()) > world:          # the minimum of one rank per sheet overshot: trim the most over-served
        n = min((k for k in names if share[k] > 1), key=lambda k: quota[k] - share[k])
        share[n] -= 1
    out, r = {}, 0
    for n in names:
        out[n] = (r, share[n])
        r += share[n]
    return out


def all_gather_fields(y_local, nf_total, group=None, out=None):
    """Reassemble the field-sharded result: y_local [nf_local, nrow] -> [nf_total, nrow] on every
    rank.  Equal shards go through one all_gather_into_tensor straight into the output (rank g's
    block lands at rows [g*nf_local, (g+1)*nf_local): no packing); ragged shards are padded to the
    largest shard and trimmed."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nrow = y_local.shape[1]
    f0, f1 = field_shard(nf_total, world, rank)
    assert y_local.shape[0] == f1 - f0, "rank %d holds %d fields, expected %d" % (rank, y_local.shape[0], f1 - f0)
    if out is None:
        out = torch.empty((nf_total, nrow), dtype=y_local.dtype, device=y_local.device)
    if nf_total % world == 0:
        dist.all_gather_into_tensor(out, y_local.contiguous(), group=group)
        return out
    nmax = -(-nf_total // world)
    pad = torch.zeros((nmax, nrow), dtype=y_local.dtype, device=y_local.device)
    pad[: f1 - f0] = y_local
    buf = torch.empty((world * nmax, nrow), dtype=y_local.dtype, device=y_local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    for g in range(world):
        g0, g1 = field_shard(nf_total, world, g)
        out[g0:g1] = buf[g * nmax: g * nmax + (g1 - g0)]
    return out


class CudaOps:
    """Stream / event plumbing of the sharded apply on a HIP device (torch.cuda == HIP on ROCm)."""

    def __init__(self, device):
        self.device = device
        self.comm = torch.cuda.Stream(device=device)

    def current(self):
        return torch.cuda.current_stream(self.device)

    def event(self):
        return torch.cuda.Event()

    def record(self, ev, stream):
        ev.record(stream)

    def wait_event(self, stream, ev):
        stream.wait_event(ev)

    def on_comm(self):
        return torch.cuda.stream(self.comm)

    def join_comm(self):
        torch.cuda.current_stream(self.device).wait_stream(self.comm)

    def stream_handle(self, stream):
        return stream.cuda_stream


class HostOps:
    """The same plumbing for CPU tensors (gloo rehearsals): everything is synchronous."""

    device = torch.device("cpu")
    comm = None

    class _Ctx:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def current(self):
        return None

    def event(self):
        return None

    def record(self, ev, stream):
        pass

    def wait_event(self, stream, ev):
        pass

    def on_comm(self):
        return self._Ctx()

    def join_comm(self):
        pass

    def stream_handle(self, stream):
        return 0


class FieldShardedApply:
    """Weighted::apply of nf_total fields, sharded by field over the ranks of `group`.

    apply(x_local) enqueues the local SpMM on the calling stream.  Results are reassembled with
    one all-gather per `steps_per_gather` applies (fewer, larger collectives: a [nf_local, nrow]
    AvI result is only tens of KB, far below the size at which an RCCL call is bandwidth- rather
    than latency-bound), issued on a second stream so it overlaps the following SpMMs.  Buffers
    are double-buffered by group; result(g, slot) is valid after wait().

    `ops` carries the stream/event plumbing (CudaOps on a GPU, HostOps under gloo) and
    `local_apply(x, y)` / `local_apply_many(xs, ys)` the rank-local SpMM writing into y: by default
    the C-ABI device apply of `weighted`; the CPU tests inject the oracle, so the buffer layout, the
    group choreography and the collectives of THIS class run with world > 1 on gloo."""

    def __init__(self, weighted, nf_total, group=None, device=None, steps_per_gather=1, ops=None,
                 local_apply=None, local_apply_many=None, nrow=None, ncol=None):
        self.w, self.nf_total, self.group = weighted, nf_total, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.f0, self.f1 = field_shard(nf_total, self.world, self.rank)
        self.G = int(steps_per_gather)
        assert self.G >= 1
        self.ops = ops if ops is not None else CudaOps(device)
        self.device = self.ops.device
        nl = self.f1 - self.f0
        self.nl = nl
        self.nrow = nrow = weighted.nrow_d if nrow is None else nrow
        self.ncol = weighted.ncol_d if ncol is None else ncol
        self.equal = nf_total % self.world == 0
        self.nmax = -(-nf_total // self.world)                # fields of the largest shard
        # field planes padded to 512 bytes: whole-line wave stores (spmm.hip, shortrow); ragged shards
        # are padded to nmax fields (zeros) so that ONE all_gather_into_tensor serves every layout
        ld = self.ld = (nrow + 63) // 64 * 64
        dev = self.device
        self._y = [torch.zeros((self.G, self.nmax, ld), dtype=torch.float64, device=dev) for _ in range(2)]
        self._out = [torch.empty((self.world, self.G, self.nmax, ld), dtype=torch.float64, device=dev) for _ in range(2)]
        self._done = [self.ops.event() for _ in range(2)]
        self._free = [self.ops.event() for _ in range(2)]
        self._used = [False, False]
        self._i = 0
        self._cur = None
        self._cur_h = 0
        if local_apply is None:
            # raw C-ABI calls, arguments prepared once: the per-apply host cost must stay below the ~10 us kernel
            import ctypes as C
            from . import _capi
            L = _capi.lib()
            h, check, nrow_, ld_ = weighted._h, _capi.check, nrow, ld

            def local_apply(x_ptr, ldx, y, fill, stream_h):
                rc = L.ibh_weighted_apply_device(h, C.c_void_p(x_ptr), nl, ldx, C.c_void_p(y.data_ptr()), ld_, fill, 0,
                                                 C.c_void_p(stream_h))
                if rc != 0:
                    check(rc)

            def local_apply_many(x_ptrs, ldx, ys, fill, stream_h):
                m = len(x_ptrs)
                xa = (C.c_void_p * m)(*x_ptrs)
                ya = (C.c_void_p * m)(*[y.data_ptr() for y in ys])
                rc = L.ibh_weighted_apply_many_device(h, m, xa, nl, ldx, ya, ld_, fill, 0, C.c_void_p(stream_h))
                if rc != 0:
                    check(rc)
        self._apply, self._apply_many = local_apply, local_apply_many

    # ---- submission ----------------------------------------------------------------------------
    def _open_group(self, g):
        cur = self._cur = self.ops.current()
        self._cur_h = self.ops.stream_handle(cur)
        if self._used[g]:
            self.ops.wait_event(cur, self._free[g])       # this group's buffers are still being gathered
        return cur

    def apply(self, x_local, fill=float("nan")):
        """x_local: float64 tensor [nf_local, ncol_d] on the ops' device, row-contiguous.  Returns (group, slot)."""
        assert x_local.shape[0] == self.nl
        return self.apply_ptr(x_local if self._is_host() else x_local.data_ptr(), x_local.stride(0) if self.nl > 1 else self.ncol, fill)

    def _is_host(self):
        return isinstance(self.ops, HostOps)

    def apply_ptr(self, x_ptr, ldx, fill=float("nan")):
        """Same with a raw device pointer (int) and leading dimension: the lean path for callers that
        keep their field batches resident and call this every few microseconds.  (Under HostOps the
        "pointer" is the tensor itself.)"""
        i = self._i
        g, slot = (i // self.G) & 1, i % self.G
        cur = self._cur
        if cur is None or slot == 0:
            cur = self._open_group(g)
        if self.nl:
            self._apply(x_ptr, ldx, self._y[g][slot][: self.nl], fill, self._cur_h)
        self._i = i + 1
        if slot == self.G - 1:
            self._gather(g, cur)
        return g, slot

    def apply_many_ptr(self, x_ptrs, ldx, fill=float("nan")):
        """len(x_ptrs) applies through ONE batched launch (ibh_weighted_apply_many_device) into the next
        slots of the current group (they must fit: len <= steps_per_gather - slot); the group is
        gathered when it fills up (or by flush()).  Returns (group, first slot)."""
        m, i = len(x_ptrs), self._i
        g, slot = (i // self.G) & 1, i % self.G
        assert 1 <= m <= self.G - slot, "apply_many_ptr: %d applies do not fit the %d free slots of the group" % (m, self.G - slot)
        cur = self._cur
        if cur is None or slot == 0:
            cur = self._open_group(g)
        if self.nl:
            ys = [self._y[g][slot + s][: self.nl] for s in range(m)]
            if self._apply_many is not None:
                self._apply_many(list(x_ptrs), ldx, ys, fill, self._cur_h)
            else:
                for xp, y in zip(x_ptrs, ys):
                    self._apply(xp, ldx, y, fill, self._cur_h)
        self._i = i + m
        if slot + m == self.G:
            self._gather(g, cur)
        return g, slot

    def _gather(self, g, cur):
        self.ops.record(self._done[g], cur)
        with self.ops.on_comm():
            if self.ops.comm is not None:
                self.ops.wait_event(self.ops.comm, self._done[g])
            # flat views: rank r's [G, nmax, ld] block lands at _out[g][r] on both RCCL and gloo
            dist.all_gather_into_tensor(self._out[g].view(-1), self._y[g].view(-1), group=self.group)
            self.ops.record(self._free[g], self.ops.comm)
        self._used[g] = True

    def flush(self):
        """Gather a partially filled group (end of a run)."""
        if self._i % self.G != 0:
            g = (self._i // self.G) & 1
            self._gather(g, self.ops.current())
            self._i += self.G - self._i % self.G
        self._cur = None

    def wait(self):
        self.ops.join_comm()

    def result(self, g, slot):
        """[nf_total, nrow] field-major gathered fields of one apply (a view for equal shards)."""
        o = self._out[g][:, slot]                         # [world, nmax, ld]
        if self.equal:
            return o.reshape(self.world * self.nmax, self.ld)[:, : self.nrow]
        parts = []
        for r in range(self.world):
            r0, r1 = field_shard(self.nf_total, self.world, r)
            parts.append(o[r, : r1 - r0, : self.nrow])
        return torch.cat(parts, dim=0)


# ---- the same behind the C-ABI: ibh_comm + ibh_weighted_apply_sharded_device (RCCL called from the library) ----------------
class Communicator:
    """ibh_comm: the ranks of a field-sharded regrid, one process per GPU.  `bootstrap` ships rank 0's 128-byte RCCL unique id
    to the other ranks -- by default through torch.distributed.broadcast_object_list on whatever process group is up (any
    backend: it only carries 128 bytes); a C++ host would use its own channel (ModelE: MPI_Bcast).  world == 1 needs nothing.
    `exchange` (a Python callable, tests only) replaces RCCL by a custom transport (ibh_comm_create_custom)."""

    def __init__(self, world=1, rank=0, bootstrap=None, exchange=None, rccl=None, gatherv=None, stream=None, planes_padded=False):
        """rccl: None = only when world > 1; True = also at world 1 (a one-rank RCCL communicator: exercises the library's
        RCCL binding on a single GPU).  gatherv (custom transports, tests): callable(d_base, offsets, world, rank, stream) for
        pieces of unequal size -- offsets is a list of world + 1 byte offsets (ibh_comm_set_custom_gatherv: the sharded assembly).
        stream: a torch.cuda.Stream the exchanges run on instead of a stream the communicator creates and destroys
        (ibh_comm_set_stream: the caller's stream outlives the communicator -- what a transport whose allocator keeps
        per-stream state wants); planes_padded: the gap between nrow_d and the row stride of the result arrays is padding
        (ibh_comm_set_option)."""
        import ctypes as C
        from . import _capi
        self._capi, self._C = _capi, C
        L = _capi.lib()
        h = C.c_void_p()
        self._cb = self._cbv = None
        self._stream = stream
        if exchange is not None:
            proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p)

            def _cb(user, d_base, count, stride, w, r, stream):
                try:
                    exchange(d_base, count, stride, w, r, stream)
                    return 0
                except Exception:      # noqa: BLE001 -- must not unwind through the C frame
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = proto(_cb)
            _capi.check(L.ibh_comm_create_custom(world, rank, C.cast(self._cb, C.c_void_p), None, C.byref(h)))
            if gatherv is not None:
                proto_v = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_void_p)

                def _cbv(user, d_base, offs, w, r, stream):
                    try:
                        gatherv(d_base, [int(offs[k]) for k in range(w + 1)], w, r, stream)
                        return 0
                    except Exception:      # noqa: BLE001 -- must not unwind through the C frame
                        import traceback
                        traceback.print_exc()
                        return 1
                self._cbv = proto_v(_cbv)
                _capi.check(L.ibh_comm_set_custom_gatherv(h, C.cast(self._cbv, C.c_void_p)))
        elif world == 1 and not rccl:
            _capi.check(L.ibh_comm_create(1, 0, None, C.byref(h)))
        else:
            buf = C.create_string_buffer(128)
            # EVERY rank asks for an id first (rank 0's is the one that counts): a rank that cannot load RCCL finds out here,
            # before any collective -- and with the default bootstrap the ranks agree on that (all-reduce) before the broadcast,
            # so one failing rank makes all of them raise instead of leaving the others inside a broadcast it never joins
            err = None
            try:
                _capi.check(L.ibh_comm_unique_id(buf))
            except Exception as e:      # noqa: BLE001
                err = e
            if world == 1:
                if err is not None:
                    raise err
                uid = buf.raw
            elif bootstrap is None:
                dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
                ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) != 1:
                    raise RuntimeError("ibh_comm: RCCL is not usable on every rank (this rank: %s)" % (err if err is not None else "fine"))
                box = [buf.raw]
                dist.broadcast_object_list(box, src=0)
                uid = box[0]
            else:
                if err is not None:
                    raise err
                uid = bootstrap(buf.raw if rank == 0 else None)
            _capi.check(L.ibh_comm_create(world, rank, uid, C.byref(h)))
        self._h, self.world, self.rank = h, world, rank
        if stream is not None:
            _capi.check(L.ibh_comm_set_stream(h, C.c_void_p(stream.cuda_stream)))
        if planes_padded:
            _capi.check(L.ibh_comm_set_option(h, b"planes_padded", 1))

    def set_option(self, key, value):
        """ibh_comm_set_option, e.g. ("planes_padded", 1)"""
        self._capi.check(self._capi.lib().ibh_comm_set_option(self._h, key.encode(), int(value)))

    def close(self):
        """Synchronise the exchanges and destroy the communicator now (idempotent; `del` does the same)."""
        h, self._h = getattr(self, "_h", None), None
        if h is not None:
            self._capi.destroy("ibh_comm_destroy", h)

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown
            pass

    def wait(self, stream=None):
        """Make `stream` (a raw stream handle; default torch's current stream) wait for the exchanges enqueued so far."""
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        self._capi.check(self._capi.lib().ibh_comm_wait(self._h, self._C.c_void_p(s)))


def apply_sharded(weighted, comm, x_local, out_all=None, fill=float("nan"), block_fields=0, stream=None):
    """ibh_weighted_apply_sharded_device: x_local [nvar_local, ncol_d] (torch.float64, CUDA) -> out_all
    [world * nvar_local, nrow_d] on every rank (rows padded to 512 bytes when allocated here); complete after comm.wait()."""
    import ctypes as C
    from . import _capi
    assert x_local.is_cuda and x_local.dtype == torch.float64 and x_local.dim() == 2 and x_local.stride(1) == 1
    nl = x_local.shape[0]
    if out_all is None:
        ld = (weighted.nrow_d + 63) // 64 * 64
        out_all = torch.zeros((comm.world * nl, ld), dtype=torch.float64, device=x_local.device)[:, : weighted.nrow_d]
    assert out_all.shape == (comm.world * nl, weighted.nrow_d) and out_all.stride(1) == 1
    s = torch.cuda.current_stream(x_local.device).cuda_stream if stream is None else stream
    _capi.check(_capi.lib().ibh_weighted_apply_sharded_device(
        weighted._h, comm._h, C.c_void_p(x_local.data_ptr()), nl, max(x_local.stride(0), weighted.ncol_d), C.c_void_p(out_all.data_ptr()),
        max(out_all.stride(0), weighted.nrow_d), float(fill), int(block_fields), C.c_void_p(s)))
    return out_all


def apply_many_sharded(weighted, comm, xs_local, outs_all, fill=float("nan"), stream=None):
    """ibh_weighted_apply_many_sharded_device: len(xs_local) field batches through ONE SpMM launch and ONE grouped exchange;
    outs_all: list of [world * nvar_local, nrow_d] result arrays (all with the same row stride)."""
    import ctypes as C
    from . import _capi
    nb = len(xs_local)
    assert nb == len(outs_all) and nb >= 1
    nl, lda = xs_local[0].shape[0], max(xs_local[0].stride(0), weighted.ncol_d)
    ldb = max(outs_all[0].stride(0), weighted.nrow_d)
    for x, o in zip(xs_local, outs_all):
        assert x.is_cuda and x.dtype == torch.float64 and x.shape == (nl, weighted.ncol_d) and max(x.stride(0), weighted.ncol_d) == lda
        assert o.shape == (comm.world * nl, weighted.nrow_d) and max(o.stride(0), weighted.nrow_d) == ldb and o.stride(1) == 1
    xa = (C.c_void_p * nb)(*[x.data_ptr() for x in xs_local])
    oa = (C.c_void_p * nb)(*[o.data_ptr() for o in outs_all])
    s = torch.cuda.current_stream(xs_local[0].device).cuda_stream if stream is None else stream
    _capi.check(_capi.lib().ibh_weighted_apply_many_sharded_device(weighted._h, comm._h, nb, xa, nl, lda, oa, ldb, float(fill), C.c_void_p(s)))
    return outs_all


class CabiFieldShardedApply:
    """The interface bench.py drives (FieldShardedApply: apply_ptr / apply_many_ptr / flush / wait / result) over the C-ABI
    path: ibh_weighted_apply_(many_)sharded_device on a Communicator -- the library calls RCCL itself, as a C++ host would use
    it.  Equal shards only; two alternating result arrays per group slot, so the exchange of group g overlaps the SpMMs of
    group g+1 (the library orders work by the address ranges in flight)."""

    def __init__(self, weighted, nf_total, comm, device, steps_per_gather=1):
        import ctypes as C
        from . import _capi
        self._C, self._capi, self._L = C, _capi, _capi.lib()
        self.w, self.comm, self.G = weighted, comm, int(steps_per_gather)
        assert nf_total % comm.world == 0, "the C-ABI sharded apply takes equal shards"
        self.nl, self.nf_total = nf_total // comm.world, nf_total
        self.nrow, self.ncol = weighted.nrow_d, weighted.ncol_d
        self.ld = (self.nrow + 63) // 64 * 64
        self.device = device
        self._out = [torch.zeros((self.G, nf_total, self.ld), dtype=torch.float64, device=device) for _ in range(2)]
        comm.set_option("planes_padded", 1)        # (the planes of these arrays are rounded up to 512 bytes: the gap is ours)
        self._i = 0

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def apply_many_ptr(self, x_ptrs, ldx, fill=float("nan")):
        C, m, i = self._C, len(x_ptrs), self._i
        g, slot = (i // self.G) & 1, i % self.G
        assert 1 <= m <= self.G - slot
        xa = (C.c_void_p * m)(*x_ptrs)
        oa = (C.c_void_p * m)(*[self._out[g][slot + s].data_ptr() for s in range(m)])
        self._capi.check(self._L.ibh_weighted_apply_many_sharded_device(self.w._h, self.comm._h, m, xa, self.nl, ldx, oa, self.ld, fill, self._stream()))
        self._i = i + m
        return g, slot

    def apply_ptr(self, x_ptr, ldx, fill=float("nan")):
        C, i = self._C, self._i
        g, slot = (i // self.G) & 1, i % self.G
        self._capi.check(self._L.ibh_weighted_apply_sharded_device(self.w._h, self.comm._h, C.c_void_p(x_ptr), self.nl, ldx,
                                                                   C.c_void_p(self._out[g][slot].data_ptr()), self.ld, fill, 0, self._stream()))
        self._i = i + 1
        return g, slot

    def flush(self):
        if self._i % self.G:
            self._i += self.G - self._i % self.G

    def wait(self):
        self.comm.wait()

    def result(self, g, slot):
        return self._out[g][slot][:, : self.nrow]
