"""Sharding of the additive kernel MVM across ranks.  Two splits of the same work, both ending in
ONE all-reduce of the N x T partial result per MVM (RCCL over xGMI on MI355X: `torch.distributed` backend "nccl"; gloo
on CPU in tests); the noise term is added once (by rank 0's kernel), so every rank holds the identical result and runs
the identical CG recurrences:
  * mode "j"     — K = sum_j K_j: rank r owns a contiguous slice of the J projections and sweeps the full N x N index
                   space (north_star's split; the only one available to backends without `supports_pair_shard`);
  * mode "pairs" — rank r owns a contiguous 1/world range of the (row block, column chunk) workgroups of the symmetric
                   tile decomposition, i.e. an equal share of the (i, i') pairs, and evaluates ALL J projections on them.  Same message, but every rank runs the
                   JT = 20 kernel at full per-term efficiency (thin J-slices pay the per-pair overhead J/J_r times):
                   measured per-rank kernel time at N = 50k: 8 ranks 0.66 ms (j) vs ~0.33 ms (pairs).
Gradients (bilinear derivative) are always J-sharded.

Replaces the row-sharded `MultiDeviceKernel(kernel, devices, devices[0])` of training_routines.py:407-408.
"""
import torch
import torch.distributed as dist


def j_partition(J, world_size):
    """Contiguous, balanced split of range(J): the first J % world_size ranks get one extra projection.
    Returns a list of (j0, j1); ranks beyond J get empty ranges (j0 == j1)."""
    if J <= 0 or world_size <= 0:
        raise ValueError("J and world_size must be positive")
    base, extra = divmod(J, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def is_distributed(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def all_reduce_sum_(t, group=None):
    """In-place SUM all-reduce; identity when not running distributed."""
    if is_distributed(group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class JShard:
    """This rank's slice of the projections."""

    def __init__(self, J, group=None, mode="pairs"):
        self.group = group
        self.mode = mode
        if is_distributed(group):
            self.world_size = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        else:
            self.world_size, self.rank = 1, 0
        self.J = J
        self.j0, self.j1 = j_partition(J, self.world_size)[self.rank]

    @property
    def empty(self):
        return self.j1 <= self.j0

    def pair_shard(self, backend):
        """(world, rank) for pair-sharding, or None when mode is "j" or the backend cannot do it (then J-slices)."""
        if self.mode != "pairs" or not getattr(backend, "supports_pair_shard", False):
            return None
        return (self.world_size, self.rank)

    def sharded_mvm(self, local_mvm, V, noise):
        """local_mvm(j0, j1, noise) -> partial product of what this rank owns (a J-slice, or its share of the tile pairs in
        "pairs" mode) + noise * V.  The noise term is handed to rank 0's kernel only (fused into its slab reduce), so
        the single all-reduce yields sum over ranks + noise * V, identical on every rank, with no extra pass over V."""
        partial = local_mvm(self.j0, self.j1, noise if self.rank == 0 else 0.0)
        all_reduce_sum_(partial, self.group)
        return partial


# ---- row sharding (the SKI operator; replaces MultiDeviceKernel around GridInterpolationKernel, -----------------------
# ---- training_routines.py:407-408 with :157-158) ---------------------------------------------------------------
def row_partition(N, world_size):
    """Contiguous, balanced split of range(N): list of (r0, r1), the first N % world_size ranks one row longer."""
    if N < 0 or world_size <= 0:
        raise ValueError("N must be non-negative and world_size positive")
    base, extra = divmod(N, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


class RowShard:
    """This rank's contiguous block [r0, r1) of the N data rows, and the collectives the row-sharded solve needs
    (all SUM / MIN / MAX all-reduces of tiny tensors: the J x G x T grid histogram, k x T preconditioner projections and
    the per-column scalars of CG)."""

    def __init__(self, N, group=None):
        self.group = group
        self.N = int(N)
        # (a one-rank process group still routes its collectives through the backend: the world-size-1 RCCL test)
        self.active = dist.is_available() and dist.is_initialized()
        if self.active:
            self.world_size = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        else:
            self.world_size, self.rank = 1, 0
        self.bounds = row_partition(self.N, self.world_size)
        self.r0, self.r1 = self.bounds[self.rank]

    @property
    def local_rows(self):
        return self.r1 - self.r0

    def owner(self, row):
        for r, (a, b) in enumerate(self.bounds):
            if a <= row < b:
                return r
        raise IndexError(row)

    def all_reduce_(self, t, op="sum"):
        if self.active:
            dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op],
                            group=self.group)
        return t

    def broadcast_(self, t, src):
        if self.active:
            dist.broadcast(t, src=src, group=self.group)
        return t
