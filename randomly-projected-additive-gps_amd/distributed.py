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
    def reducer(self):
        """The all-reduce the sharded MVM / solve runs on (RCCL by default, rpgp_comm with RPGP_COMM=ipc)."""
        return get_reducer(self.group)

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
        if self.world_size > 1:
            self.reducer.all_reduce_(partial)
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

    @property
    def reducer(self):
        return get_reducer(self.group)

    def owner(self, row):
        for r, (a, b) in enumerate(self.bounds):
            if a <= row < b:
                return r
        raise IndexError(row)

    def all_reduce_(self, t, op="sum"):
        if self.active and op == "sum" and t.is_cuda and self.world_size > 1:
            return self.reducer.all_reduce_(t)          # SUM of device tensors: the configured backend (RCCL / IPC)
        if self.active:
            dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op],
                            group=self.group)
        return t

    def broadcast_(self, t, src):
        if self.active:
            dist.broadcast(t, src=src, group=self.group)
        return t


# ---- the all-reduce the sharded solve runs on --------------------------------------------------------------------
class Reducer:
    """In-place SUM all-reduce of device tensors for the sharded solve, in the two forms the stack needs:
      * `all_reduce_(t)`         — a torch tensor (float32 / float64), enqueued on the current stream;
      * `c_hook()`               — `(fn, ctx)` for the native mBCG executor (`rpgp_allreduce_fn` of include/rpgp.h): the
                                   executor calls it between its enqueue-only phases, on its launch stream.
    Backends:
      * "rccl" (default)         — torch.distributed (backend "nccl" = RCCL over xGMI; gloo in CPU tests).  The C hook is a
                                   ctypes callback that wraps the raw device pointer in a tensor view and issues the
                                   collective; RCCL orders it after the work already queued on the current stream, so
                                   there is no host synchronisation.
      * "ipc"                    — rpgp_comm (csrc/rpgp_comm.hip): one kernel launch per call over IPC-mapped peer
                                   buffers; the hook is the library's own function pointer (no Python in the loop).
                                   Opt-in: `RPGP_COMM=ipc` or `settings.comm_backend("ipc")`.
    `world_size == 1` makes every call the identity (and the hook None)."""

    def __init__(self, group=None, backend=None, max_bytes=1 << 24, device=None, force=False):
        import os
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.world_size = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        if backend is None:
            from . import settings
            backend = os.environ.get("RPGP_COMM") or settings.comm_backend.value()
        # `force`: keep the collective path at world size 1 (the one-rank RCCL test exercises the hook that way)
        self.backend = backend if (self.world_size > 1 or (force and self.active)) else "none"
        self._comm = None
        self._cb = None
        self._ws = None
        if self.backend == "ipc":
            self._init_ipc(max_bytes, device)
        elif self.backend not in ("rccl", "none"):
            raise ValueError("unknown comm backend %r (rccl | ipc)" % (backend,))

    # -- rpgp_comm bootstrap: create, exchange the IPC handles through the process group, connect
    def _init_ipc(self, max_bytes, device):
        import ctypes
        from . import _lib
        lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        comm = ctypes.c_void_p()
        handle = ctypes.create_string_buffer(_lib.RPGP_COMM_HANDLE_BYTES)
        with torch.cuda.device(self.device):
            _lib.check(lib.rpgp_comm_create(self.world_size, self.rank, int(max_bytes), ctypes.byref(comm), handle),
                       "rpgp_comm_create")
            handles = [None] * self.world_size
            dist.all_gather_object(handles, bytes(handle.raw), group=self.group)
            blob = b"".join(handles)
            _lib.check(lib.rpgp_comm_connect(comm, blob), "rpgp_comm_connect")
        self._comm = comm
        self._lib = lib
        self.max_bytes = int(lib.rpgp_comm_capacity(comm))
        dist.barrier(group=self.group)          # nobody starts publishing before every peer has mapped everybody

    def all_reduce_(self, t):
        if self.backend == "none":
            return t
        if self.backend == "ipc" and t.is_cuda and t.dtype in (torch.float32, torch.float64) and t.is_contiguous() and \
                t.numel() * t.element_size() <= self.max_bytes:
            from . import _lib
            with torch.cuda.device(t.device):
                _lib.check(self._lib.rpgp_comm_allreduce(self._comm, t.data_ptr(), t.numel(),
                                                         _lib.RPGP_F64 if t.dtype == torch.float64 else _lib.RPGP_F32,
                                                         torch.cuda.current_stream().cuda_stream), "rpgp_comm_allreduce")
            return t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def check(self):
        """Raise if a bounded wait of the IPC backend timed out (a peer was lost)."""
        if self._comm is not None:
            import ctypes
            from . import _lib
            err = ctypes.c_int(0)
            _lib.check(self._lib.rpgp_comm_error(self._comm, ctypes.byref(err)), "rpgp_comm_error")
            if err.value:
                raise RuntimeError("rpgp_comm: a peer did not arrive within the wait bound (rank %d)" % self.rank)

    def c_hook(self, workspace=None):
        """(function pointer, context pointer) for `struct rpgp_reducer`, or (None, None) for a single rank.
        `workspace`: the torch uint8 tensor the executor's buffers live in (RCCL backend: the callback turns the raw
        pointers it is handed into views of this tensor)."""
        import ctypes
        from . import _lib
        if self.backend == "none":
            return None, None
        if self.backend == "ipc":
            fn = ctypes.cast(self._lib.rpgp_comm_allreduce, ctypes.c_void_p).value
            return fn, self._comm.value
        # ONE callback per Reducer (created on first use); the workspace it may touch is mutable state set before every
        # solve — a fresh closure per solve would pin every superseded grow-only workspace for the life of the process
        self._ws = workspace
        if self._cb is None:
            group = self.group

            def _cb(ctx, buf, count, dtype, stream):
                try:
                    ws = self._ws
                    esz = 8 if dtype == _lib.RPGP_F64 else 4
                    off = int(buf) - ws.data_ptr()
                    if off < 0 or off + count * esz > ws.numel():
                        return _lib.RPGP_EINVAL
                    cur = torch.cuda.current_stream(ws.device).cuda_stream if ws.is_cuda else 0
                    if ws.is_cuda and int(stream or 0) != int(cur):
                        return _lib.RPGP_EINVAL         # the collective is ordered on torch's current stream only
                    view = ws[off:off + count * esz].view(torch.float64 if dtype == _lib.RPGP_F64 else torch.float32)
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
                    return 0
                except Exception:                       # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return _lib.RPGP_EINVAL
            self._cb = _lib.ALLREDUCE_FN(_cb)
        return ctypes.cast(self._cb, ctypes.c_void_p).value, None

    def release_workspace(self):
        """Drop the reference to the executor's workspace once its solve has returned (the executor has synchronised)."""
        self._ws = None

    def close(self):
        if self._comm is not None:
            self._lib.rpgp_comm_destroy(self._comm)
            self._comm = None


_reducers = {}


def get_reducer(group=None):
    """One Reducer per process group (the IPC backend owns mapped buffers: created once, reused by every solve)."""
    key = id(group)
    r = _reducers.get(key)
    if r is None:
        r = _reducers[key] = Reducer(group)
    return r
