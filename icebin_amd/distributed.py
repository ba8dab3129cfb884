"""Field-column sharding of a multi-field regrid across the GPUs of one node.

Every field (one row of A_b(nvar, n), icebin22m.cpp:142) is an independent SpMV with the same
matrix, so rank g owns the contiguous fields [f0, f1) of the field-major arrays: its shard is
contiguous in memory, the CSR is replicated, and the SpMM needs no communication.  One RCCL
all-gather (torch.distributed, backend "nccl" on ROCm; "gloo" in the CPU tests) reassembles
the [nf_total, nrow] result on every rank.  The reference has no counterpart: it gathers
everything to MPI rank 0 and regrids there (modele/GCMCoupler_ModelE.cpp:764-792).
"""
import torch
import torch.distributed as dist


def field_shard(nf_total, world, rank):
    """Contiguous block partition of nf_total fields: the first (nf_total % world) ranks get one more."""
    base, rem = divmod(nf_total, world)
    f0 = rank * base + min(rank, rem)
    return f0, f0 + base + (1 if rank < rem else 0)


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


class FieldShardedApply:
    """Weighted::apply of nf_total fields, sharded by field over the ranks of `group`.

    apply(x_local) enqueues the local SpMM on the calling stream.  Results are reassembled with
    one all-gather per `steps_per_gather` applies (fewer, larger collectives: a [nf_local, nrow]
    AvI result is only tens of KB, far below the size at which an RCCL call is bandwidth- rather
    than latency-bound), issued on a second stream so it overlaps the following SpMMs.  Buffers
    are double-buffered by group; result(g, slot) is valid after wait()."""

    def __init__(self, weighted, nf_total, group=None, device=None, steps_per_gather=1):
        import ctypes as C
        from . import _capi
        self.w, self.nf_total, self.group = weighted, nf_total, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        assert nf_total % self.world == 0 or steps_per_gather == 1, "grouped gathers need equal shards"
        self.f0, self.f1 = field_shard(nf_total, self.world, self.rank)
        self.device, self.G = device, int(steps_per_gather)
        self.comm = torch.cuda.Stream(device=device)
        nl, nrow = self.f1 - self.f0, weighted.nrow_d
        self.nl, self.nrow = nl, nrow
        # field planes padded to 512 bytes: whole-line wave stores (spmm.hip, shortrow) and the padded
        # buffers stay contiguous, which is what all_gather_into_tensor needs
        ld = self.ld = (nrow + 63) // 64 * 64 if self.world == 1 or nf_total % self.world == 0 else nrow
        self._y = [torch.empty((self.G, nl, ld), dtype=torch.float64, device=device) for _ in range(2)]
        if self.G == 1:
            self._out = [torch.empty((nf_total, ld), dtype=torch.float64, device=device) for _ in range(2)]
        else:
            self._out = [torch.empty((self.world, self.G, nl, ld), dtype=torch.float64, device=device) for _ in range(2)]
        self._done = [torch.cuda.Event() for _ in range(2)]
        self._free = [torch.cuda.Event() for _ in range(2)]
        self._used = [False, False]
        self._i = 0
        self._cur = None
        self._cur_ptr = None
        # raw C-ABI call, arguments prepared once: the per-apply host cost must stay below the ~10 us kernel
        self._fn = _capi.lib().ibh_weighted_apply_device
        self._check = _capi.check
        self._h = weighted._h
        self._yp = [[C.c_void_p(self._y[g][s].data_ptr()) for s in range(self.G)] for g in range(2)]
        self._C = C

    def apply(self, x_local, fill=float("nan")):
        """x_local: torch.float64 CUDA tensor [nf_local, ncol_d], contiguous.  Returns (group, slot)."""
        return self.apply_ptr(x_local.data_ptr(), x_local.stride(0), fill)

    def apply_ptr(self, x_ptr, ldx, fill=float("nan")):
        """Same with a raw device pointer (int) and leading dimension: the lean path for callers that
        keep their field batches resident and call this every few microseconds."""
        i = self._i
        g, slot = (i // self.G) & 1, i % self.G
        cur = self._cur
        if cur is None or slot == 0:
            cur = self._cur = torch.cuda.current_stream(self.device)
            self._cur_ptr = self._C.c_void_p(cur.cuda_stream)
            if self._used[g]:
                cur.wait_event(self._free[g])       # this group's buffers are still being gathered
        rc = self._fn(self._h, x_ptr, self.nl, ldx, self._yp[g][slot], self.ld, fill, 0, self._cur_ptr)
        if rc != 0:
            self._check(rc)
        self._i = i + 1
        if slot == self.G - 1:
            self._gather(g, cur)
        return g, slot

    def _gather(self, g, cur):
        self._done[g].record(cur)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self._done[g])
            if self.G == 1:
                all_gather_fields(self._y[g][0], self.nf_total, self.group, out=self._out[g])      # planes of width ld
            else:
                dist.all_gather_into_tensor(self._out[g], self._y[g], group=self.group)
            self._free[g].record(self.comm)
        self._used[g] = True

    def flush(self):
        """Gather a partially filled group (end of a run)."""
        if self._i % self.G != 0:
            g = (self._i // self.G) & 1
            self._gather(g, torch.cuda.current_stream(self.device))
            self._i += self.G - self._i % self.G
        self._cur = None

    def wait(self):
        torch.cuda.current_stream(self.device).wait_stream(self.comm)

    def result(self, g, slot):
        """[world, nf_local, nrow] view (rank-major == field-major) of the gathered fields of one apply."""
        if self.G == 1:
            return self._out[g].view(self.world, self.nl, self.ld)[:, :, :self.nrow]
        return self._out[g][:, slot, :, :self.nrow]
