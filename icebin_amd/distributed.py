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
