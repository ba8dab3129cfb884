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

    apply(x_local) runs the local SpMM on the calling stream and the all-gather on a second
    stream, so the gather of one call overlaps the SpMM of the next; results are double-buffered
    and wait() (or the returned event) orders consumers."""

    def __init__(self, weighted, nf_total, group=None, device=None):
        self.w, self.nf_total, self.group = weighted, nf_total, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.f0, self.f1 = field_shard(nf_total, self.world, self.rank)
        self.device = device
        self.comm = torch.cuda.Stream(device=device)
        nl = self.f1 - self.f0
        self._y = [torch.empty((nl, weighted.nrow_d), dtype=torch.float64, device=device) for _ in range(2)]
        self._out = [torch.empty((nf_total, weighted.nrow_d), dtype=torch.float64, device=device) for _ in range(2)]
        self._done = [torch.cuda.Event() for _ in range(2)]
        self._free = [torch.cuda.Event() for _ in range(2)]
        self._i = 0

    def apply(self, x_local, fill=float("nan")):
        s = self._i & 1
        cur = torch.cuda.current_stream(self.device)
        if self._i >= 2:
            cur.wait_event(self._free[s])        # y[s] is still being gathered from two calls ago
        self.w.apply_device(x_local, out=self._y[s], fill=fill, force_conservation=False)
        self._done[s].record(cur)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self._done[s])
            all_gather_fields(self._y[s], self.nf_total, self.group, out=self._out[s])
            self._free[s].record(self.comm)
        self._i += 1
        return self._out[s], self._free[s]

    def wait(self):
        torch.cuda.current_stream(self.device).wait_stream(self.comm)
