"""Tensor-level wrappers over the C-ABI (include/rpgp.h): torch supplies device memory and streams only.

All functions require CUDA(=HIP) float32 contiguous tensors and launch on the caller's current stream.
"""
import torch

from . import _lib

_workspaces = {}


def _require(t, name, ndim=None, allow64=False):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on a HIP device (got %s): the rpgp kernels have no CPU fallback"
                           % (name, t.device))
    if t.dtype != torch.float32 and not (allow64 and t.dtype == torch.float64):
        raise TypeError("%s must be float32%s (got %s)" % (name, " or float64" if allow64 else "", t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must be %d-dimensional (got shape %s)" % (name, ndim, tuple(t.shape)))
    return t.contiguous()


# (private torch entry points, with the public ones as the fallback of a build that lacks them)
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEV = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _stream():
    """Raw handle of the current stream of the current device.  (`torch.cuda.current_stream().cuda_stream` is the same value
    through ~10 us of Python — 13 calls per optimiser step; the step is host-bound between its launches)"""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(_CUR_DEV())
    return torch.cuda.current_stream().cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_GUARD = _NoGuard()


def _on(device):
    """`torch.cuda.device(device)` — entered only when `device` is not the current device already (the context manager reads and
    restores the device through several Python layers: ~8 us per entry, ~20 entries per optimiser step)."""
    idx = device.index
    if idx is None or idx == _CUR_DEV():
        return _NO_GUARD
    return torch.cuda.device(device)


def _workspace(device, nbytes):
    """Grow-only per-(device, stream) scratch buffer; kernels on one stream are ordered so reuse is safe."""
    key = (device.index, _stream())
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def init():
    lib = _lib.load()
    _lib.check(lib.rpgp_init(), "rpgp_init")


def space_equally(P, lr, niter):
    """DPA-GP diversification on the device (rpgp_space_equally): returns (P_new [J x d, unit rows], final_loss [1 x 1])."""
    lib = _lib.load()
    Q = _require(P.detach().clone(), "P", 2)
    J, d = Q.shape
    loss = torch.empty(1, dtype=torch.float32, device=Q.device)
    with _on(Q.device):
        _lib.check(lib.rpgp_space_equally(Q.data_ptr(), J, d, float(lr), int(niter), loss.data_ptr(), _stream()),
                   "rpgp_space_equally")
    return Q, loss.reshape(1, 1)


def project(X, Peff):
    """Z = X @ Peff  (N x d)(d x J) -> N x J."""
    lib = _lib.load()
    X = _require(X, "X", 2, allow64=True)
    Peff = _require(Peff, "Peff", 2, allow64=True)
    N, d = X.shape
    if Peff.shape[0] != d or Peff.dtype != X.dtype:
        raise ValueError("Peff must be d x J with d=%d and the dtype of X (got %s)" % (d, tuple(Peff.shape)))
    J = Peff.shape[1]
    Z = torch.empty((N, J), dtype=X.dtype, device=X.device)
    fn = lib.rpgp_project_f64 if X.dtype == torch.float64 else lib.rpgp_project
    with _on(X.device):
        _lib.check(fn(X.data_ptr(), Peff.data_ptr(), Z.data_ptr(), N, d, J, _stream()), "rpgp_project")
    return Z


def project_grad(X, G):
    """dPeff = X^T @ G  (d x J)."""
    lib = _lib.load()
    X = _require(X, "X", 2, allow64=True)
    G = _require(G, "G", 2, allow64=True)
    N, d = X.shape
    if G.shape[0] != N or G.dtype != X.dtype:
        raise ValueError("G must have N=%d rows and the dtype of X" % N)
    J = G.shape[1]
    out = torch.empty((d, J), dtype=X.dtype, device=X.device)
    fn = lib.rpgp_project_grad_f64 if X.dtype == torch.float64 else lib.rpgp_project_grad
    with _on(X.device):
        _lib.check(fn(X.data_ptr(), G.data_ptr(), out.data_ptr(), N, d, J, _stream()), "rpgp_project_grad")
    return out


def _as_matrix(V, N, name, allow64=False):
    squeeze = V.dim() == 1
    V2 = V.unsqueeze(1) if squeeze else V
    V2 = _require(V2, name, 2, allow64=allow64)
    if V2.shape[0] != N:
        raise ValueError("%s must have %d rows (got %s)" % (name, N, tuple(V.shape)))
    return V2, squeeze


def mvm_sym(Z, V, scale, noise=0.0, j0=0, j1=None, out=None, shard=None):
    """out = scale * sum_{j in [j0,j1)} K_j(Z,Z) @ V + noise * V.
    shard = (world, rank): only this rank's 1/world share of the (i,i') tile pairs (pair-sharding; partial output)."""
    world, rank = shard if shard is not None else (1, 0)
    lib = _lib.load()
    Z = _require(Z, "Z", 2, allow64=True)
    N, J = Z.shape
    j1 = J if j1 is None else j1
    V2, squeeze = _as_matrix(V, N, "V", allow64=True)
    T = V2.shape[1]
    if out is None:
        out = torch.empty_like(V2)
    if Z.dtype == torch.float64:
        if V2.dtype != torch.float64 or world != 1:
            raise TypeError("float64 MVM needs float64 V (and does not support pair-sharding)")
        with _on(Z.device):
            _lib.check(lib.rpgp_mvm_f64(Z.data_ptr(), Z.data_ptr(), V2.data_ptr(), out.data_ptr(), N, N, J, J, T, j0, j1,
                                        float(scale), float(noise), _stream()), "rpgp_mvm_f64")
        return out.squeeze(1) if squeeze else out
    with _on(Z.device):
        nbytes = lib.rpgp_mvm_sym_range_workspace_bytes(N, T, world, rank)
        ws = _workspace(Z.device, nbytes)
        _lib.check(lib.rpgp_mvm_sym_range(Z.data_ptr(), V2.data_ptr(), out.data_ptr(), N, J, T, j0, j1, world, rank,
                                          float(scale), float(noise), ws.data_ptr(), ws.numel(), _stream()),
                   "rpgp_mvm_sym")
    return out.squeeze(1) if squeeze else out


class _NullRange:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL_RANGE = _NullRange()
_RANGES_ON = None


class _Range:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        _lib.load().rpgp_range_push(self.name)
        return self

    def __exit__(self, *exc):
        _lib.load().rpgp_range_pop()
        return False


def trace_range(name):
    """Context manager marking a phase of a training step as a roctx range (rpgp_range_push / rpgp_range_pop; SURVEY.md §5
    "Tracing / profiling").  Live only when a marker library is loaded in the process (`rocprofv3 --marker-trace`); otherwise
    the shared no-op object is returned — an unprofiled step pays one attribute test per phase."""
    global _RANGES_ON
    if _RANGES_ON is None:
        try:
            _RANGES_ON = bool(_lib.load().rpgp_range_available())
        except Exception:
            _RANGES_ON = False
    return _Range(name.encode()) if _RANGES_ON else _NULL_RANGE


class Prepared:
    """Device-side tables of rpgp_prepare for one Z (centred, pre-scaled coordinates for the factorised fast path)."""

    def __init__(self, Z):
        import ctypes
        lib = _lib.load()
        Z = _require(Z, "Z", 2)
        self.N, self.J = Z.shape
        self.device = Z.device
        self.fast_ok = False
        self.max_abs = float("inf")
        self.buf = None
        if self.J > 64:
            return
        with _on(Z.device):
            nbytes = lib.rpgp_prepare_bytes(self.N, self.J)
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=Z.device)
            _lib.check(lib.rpgp_prepare(Z.data_ptr(), self.N, self.J, self.J, self.buf.data_ptr(), self.buf.numel(),
                                        _stream()), "rpgp_prepare")
            ok, mx = ctypes.c_int(0), ctypes.c_float(0)
            _lib.check(lib.rpgp_prepare_status(self.buf.data_ptr(), ctypes.byref(ok), ctypes.byref(mx), _stream()),
                       "rpgp_prepare_status")
        self.fast_ok = bool(ok.value)
        self.max_abs = float(mx.value)


def mvm_sym_prepared(prep, V, scale, noise=0.0, j0=0, j1=None, out=None, shard=None):
    """Factorised fast path: same result contract as mvm_sym for the Z that `prep` was built from."""
    world, rank = shard if shard is not None else (1, 0)
    lib = _lib.load()
    if not prep.fast_ok:
        raise RuntimeError("rpgp_prepare flagged the coordinate range as unsafe for the factorised path "
                           "(max|a| = %g): use mvm_sym" % prep.max_abs)
    N, J = prep.N, prep.J
    j1 = J if j1 is None else j1
    V2, squeeze = _as_matrix(V, N, "V")
    T = V2.shape[1]
    if out is None:
        out = torch.empty_like(V2)
    with _on(prep.device):
        nbytes = lib.rpgp_mvm_sym_range_workspace_bytes(N, T, world, rank)
        ws = _workspace(prep.device, nbytes)
        _lib.check(lib.rpgp_mvm_sym_prepared_range(prep.buf.data_ptr(), V2.data_ptr(), out.data_ptr(), N, J, T, j0, j1,
                                                   world, rank, float(scale), float(noise), ws.data_ptr(), ws.numel(),
                                                   _stream()), "rpgp_mvm_sym_prepared")
    return out.squeeze(1) if squeeze else out


def mvm_rect(Z1, Z2, V, scale, j0=0, j1=None):
    """out = scale * sum_j K_j(Z1,Z2) @ V   (M x T)."""
    lib = _lib.load()
    Z1 = _require(Z1, "Z1", 2, allow64=True)
    Z2 = _require(Z2, "Z2", 2, allow64=True)
    M, J = Z1.shape
    N = Z2.shape[0]
    if Z2.shape[1] != J:
        raise ValueError("Z1 and Z2 must have the same number of projections")
    j1 = J if j1 is None else j1
    V2, squeeze = _as_matrix(V, N, "V", allow64=True)
    T = V2.shape[1]
    out = torch.empty((M, T), dtype=Z1.dtype, device=Z1.device)
    if Z1.dtype == torch.float64:
        with _on(Z1.device):
            _lib.check(lib.rpgp_mvm_f64(Z1.data_ptr(), Z2.data_ptr(), V2.data_ptr(), out.data_ptr(), M, N, J, J, T, j0,
                                        j1, float(scale), 0.0, _stream()), "rpgp_mvm_f64")
        return out.squeeze(1) if squeeze else out
    with _on(Z1.device):
        nbytes = lib.rpgp_mvm_rect_workspace_bytes(M, N, T)
        ws = _workspace(Z1.device, nbytes)
        _lib.check(lib.rpgp_mvm_rect(Z1.data_ptr(), Z2.data_ptr(), V2.data_ptr(), out.data_ptr(), M, N, J, J, T,
                                     j0, j1, float(scale), ws.data_ptr(), ws.numel(), _stream()), "rpgp_mvm_rect")
    return out.squeeze(1) if squeeze else out


def dense(Z1, Z2, scale, j0=0, j1=None, pad=False):
    """Dense block K(Z1,Z2) (M x N).  pad=True (float32): the rows are laid out with a leading dimension rounded up to
    64 floats (256-byte aligned rows) and an M x N view of that buffer is returned — the cached-K stream
    (rpgp_dense_mvm) loads 16 B per lane and needs 16-byte aligned rows, which a bare N x N array only has when 4 | N
    (measured at N = 14 939: 3.6 TB/s on the element-wise edge path against 5.5+ with padded rows)."""
    lib = _lib.load()
    Z1 = _require(Z1, "Z1", 2, allow64=True)
    Z2 = _require(Z2, "Z2", 2, allow64=True)
    M, J = Z1.shape
    N = Z2.shape[0]
    if Z2.shape[1] != J:
        raise ValueError("Z1 and Z2 must have the same number of projections")
    j1 = J if j1 is None else j1
    if Z1.dtype == torch.float64:
        out = torch.empty((M, N), dtype=Z1.dtype, device=Z1.device)
        with _on(Z1.device):
            _lib.check(lib.rpgp_dense_f64(Z1.data_ptr(), Z2.data_ptr(), out.data_ptr(), M, N, J, J, N, j0, j1,
                                          float(scale), _stream()), "rpgp_dense_f64")
        return out
    ld = (N + 63) // 64 * 64 if (pad and N % 64) else N        # decided before allocating: ONE buffer
    out = torch.empty((M, ld), dtype=Z1.dtype, device=Z1.device)
    if ld != N:
        out = out[:, :N]
    with _on(Z1.device):
        _lib.check(lib.rpgp_dense(Z1.data_ptr(), Z2.data_ptr(), out.data_ptr(), M, N, J, J, ld, j0, j1, float(scale),
                                  _stream()), "rpgp_dense")
    return out


def bilinear_grad(Z, L, R, scale, j0=0, j1=None):
    """(gZ [N x J], gscale [scalar tensor]) = d/dZ, d/dscale of sum((L R^T) * K(Z,Z))."""
    lib = _lib.load()
    Z = _require(Z, "Z", 2, allow64=True)
    N, J = Z.shape
    j1 = J if j1 is None else j1
    L2, _ = _as_matrix(L, N, "L", allow64=True)
    R2, _ = _as_matrix(R, N, "R", allow64=True)
    if L2.shape != R2.shape:
        raise ValueError("L and R must have the same shape")
    T = L2.shape[1]
    # (the kernels WRITE gZ[:, j0:j1] and gs: the zero fill is only needed for the columns outside a J-slice and for the
    #  accumulation over 12-column pieces)
    full = j0 == 0 and j1 == J and (T <= 12 or Z.dtype == torch.float64)
    gZ = (torch.empty if full else torch.zeros)((N, J), dtype=Z.dtype, device=Z.device)
    gs = (torch.empty if full else torch.zeros)((), dtype=Z.dtype, device=Z.device)
    if Z.dtype == torch.float64:
        scratch = torch.empty(N, dtype=torch.float64, device=Z.device)
        with _on(Z.device):
            _lib.check(lib.rpgp_bilinear_grad_f64(Z.data_ptr(), L2.data_ptr(), R2.data_ptr(), gZ.data_ptr(), gs.data_ptr(),
                                                  N, J, J, T, j0, j1, float(scale), scratch.data_ptr(), _stream()),
                       "rpgp_bilinear_grad_f64")
        return gZ, gs
    with _on(Z.device):
        if T <= 12:
            nbytes = lib.rpgp_bilinear_grad_workspace_bytes(N, j1 - j0)
            ws = _workspace(Z.device, nbytes)
            _lib.check(lib.rpgp_bilinear_grad(Z.data_ptr(), L2.data_ptr(), R2.data_ptr(), gZ.data_ptr(), gs.data_ptr(),
                                              N, J, J, T, j0, j1, float(scale), ws.data_ptr(), ws.numel(), _stream()),
                       "rpgp_bilinear_grad")
        else:
            # wide blocks are processed 12 columns at a time (the derivative is additive over columns)
            nbytes = lib.rpgp_bilinear_grad_workspace_bytes(N, j1 - j0)
            ws = _workspace(Z.device, nbytes)
            gZp = torch.empty_like(gZ)
            gsp = torch.empty_like(gs)
            for t0 in range(0, T, 12):
                Lc = L2[:, t0:t0 + 12].contiguous()
                Rc = R2[:, t0:t0 + 12].contiguous()
                _lib.check(lib.rpgp_bilinear_grad(Z.data_ptr(), Lc.data_ptr(), Rc.data_ptr(), gZp.data_ptr(),
                                                  gsp.data_ptr(), N, J, J, Lc.shape[1], j0, j1, float(scale),
                                                  ws.data_ptr(), ws.numel(), _stream()), "rpgp_bilinear_grad")
                gZ[:, j0:j1] += gZp[:, j0:j1]
                gs += gsp
    return gZ, gs


def bilinear_grad_dense(Z, S, scale, j0=0, j1=None):
    """(gZ, gscale) = d/dZ, d/dscale of 0.5 * sum(S * K(Z,Z)) for an explicit symmetric N x N weight matrix S."""
    lib = _lib.load()
    Z = _require(Z, "Z", 2, allow64=True)
    S = _require(S, "S", 2, allow64=True)
    N, J = Z.shape
    if S.shape != (N, N):
        raise ValueError("S must be %d x %d" % (N, N))
    j1 = J if j1 is None else j1
    gZ = torch.zeros((N, J), dtype=Z.dtype, device=Z.device)
    gs = torch.zeros((), dtype=Z.dtype, device=Z.device)
    if Z.dtype == torch.float64:
        scratch = torch.empty(N, dtype=torch.float64, device=Z.device)
        with _on(Z.device):
            _lib.check(lib.rpgp_bilinear_grad_dense_f64(Z.data_ptr(), S.data_ptr(), gZ.data_ptr(), gs.data_ptr(), N, J, J,
                                                        N, j0, j1, float(scale), scratch.data_ptr(), _stream()),
                       "rpgp_bilinear_grad_dense_f64")
        return gZ, gs
    with _on(Z.device):
        nbytes = lib.rpgp_bilinear_grad_workspace_bytes(N, j1 - j0)
        ws = _workspace(Z.device, nbytes)
        _lib.check(lib.rpgp_bilinear_grad_dense(Z.data_ptr(), S.data_ptr(), gZ.data_ptr(), gs.data_ptr(), N, J, J, N,
                                                j0, j1, float(scale), ws.data_ptr(), ws.numel(), _stream()),
                   "rpgp_bilinear_grad_dense")
    return gZ, gs


def pivoted_cholesky(Z, scale, rank):
    """L (N x rank) with K(Z,Z) ~= L L^T (greedy pivots), one launch."""
    lib = _lib.load()
    Z = _require(Z, "Z", 2)
    N, J = Z.shape
    L = torch.empty((N, rank), dtype=torch.float32, device=Z.device)
    work = torch.empty(N + _lib.RPGP_PIVCHOL_SCRATCH, dtype=torch.float32, device=Z.device)
    with _on(Z.device):
        _lib.check(lib.rpgp_pivoted_cholesky(Z.data_ptr(), L.data_ptr(), work.data_ptr(), N, J, J, int(rank),
                                             float(scale), _stream()), "rpgp_pivoted_cholesky")
    return L


def family_pivoted_cholesky(fam, Z, scale, rank, weight_sum):
    """Pivoted Cholesky of a family member's kernel matrix (diagonal = scale * weight_sum)."""
    lib = _lib.load()
    Z = _family_cols(fam, Z, "Z")
    N, J = Z.shape
    L = torch.empty((N, rank), dtype=torch.float32, device=Z.device)
    work = torch.empty(N + _lib.RPGP_PIVCHOL_SCRATCH, dtype=torch.float32, device=Z.device)
    with _on(Z.device):
        _lib.check(lib.rpgp_family_pivoted_cholesky(fam.ref, Z.data_ptr(), L.data_ptr(), work.data_ptr(), N, J, int(rank),
                                                    float(scale), float(weight_sum), _stream()),
                   "rpgp_family_pivoted_cholesky")
    return L


def dense_mvm(Kd, V, noise=0.0):
    """out = Kd @ V + noise * V for a cached dense symmetric kernel matrix."""
    lib = _lib.load()
    if isinstance(Kd, torch.Tensor) and Kd.dim() == 2 and Kd.stride(1) == 1 and Kd.stride(0) >= Kd.shape[1]:
        _require(Kd[:1], "Kd", 2)               # device / dtype checks; a row-padded view is used as it is
    else:
        Kd = _require(Kd, "Kd", 2)
    N = Kd.shape[0]
    V2, squeeze = _as_matrix(V, N, "V")
    T = V2.shape[1]
    out = torch.empty_like(V2)
    with _on(Kd.device):
        _lib.check(lib.rpgp_dense_mvm(Kd.data_ptr(), V2.data_ptr(), out.data_ptr(), N, Kd.stride(0), T, float(noise),
                                      _stream()), "rpgp_dense_mvm")
    return out.squeeze(1) if squeeze else out


class SymCache:
    """Packed symmetric cache of the additive kernel on Z (rpgp_symcache_build): every unordered pair once, in the order
    the symmetric sweep consumes it — half the bytes of the dense N x N matrix.  `shard` = (world, rank) keeps only
    this rank's share of the pairs (the product is then a partial result, summed by one all-reduce like the pair-sharded
    fused MVM).  Values are unscaled (sum over the projections); `scale` and `noise` are applied by the product.
    `wide`: False = rotation order (blocks of 1..4 right-hand sides stream at the HBM rate), True = 16 x 16 matrix-core
    tiles (blocks of up to 16 right-hand sides per pass: the training block)."""

    def __init__(self, Z, j0=0, j1=None, shard=None, wide=False):
        lib = _lib.load()
        Z = _require(Z, "Z", 2)
        self.N, J = Z.shape
        j1 = J if j1 is None else j1
        self.world, self.rank = (1, 0) if shard is None else (int(shard[0]), int(shard[1]))
        nbytes = lib.rpgp_symcache_bytes(self.N, self.world, self.rank)
        self.buf = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=Z.device)
        self.nbytes = nbytes
        self.layout = _lib.RPGP_SYMCACHE_WIDE if wide else _lib.RPGP_SYMCACHE_THIN
        with _on(Z.device):
            _lib.check(lib.rpgp_symcache_build(Z.data_ptr(), self.buf.data_ptr(), nbytes, self.N, Z.stride(0), j0, j1,
                                               self.layout, self.world, self.rank, _stream()), "rpgp_symcache_build")

    @property
    def device(self):
        return self.buf.device


def symcache_mvm(cache, V, scale, noise=0.0):
    """out = scale * K V + noise * V from a SymCache (a partial product when the cache is one rank's shard)."""
    lib = _lib.load()
    V2, squeeze = _as_matrix(V, cache.N, "V")
    T = V2.shape[1]
    out = torch.empty_like(V2)
    with _on(cache.device):
        ws = _workspace(cache.device, lib.rpgp_symcache_workspace_bytes(cache.N, T, cache.world, cache.rank))
        _lib.check(lib.rpgp_symcache_mvm(cache.buf.data_ptr(), cache.nbytes, cache.layout, V2.data_ptr(), out.data_ptr(), cache.N, T,
                                         float(scale), float(noise), cache.world, cache.rank, ws.data_ptr(), ws.numel(),
                                         _stream()), "rpgp_symcache_mvm")
    return out.squeeze(1) if squeeze else out


# ------------------------------------------------------------------------------------------------ SKI path

_SKI_KINDS = {"RBF": 0, "Matern": 1, "InverseMQ": 2, "Cosine": 3}          # RPGP_KIND_* of include/rpgp.h


def _ski_set_kind(gp, kind):
    """The 1-D sub-kernel the grid's Toeplitz matrix is built from rides in the flags word of the grid block (bits 2-3:
    csrc/rpgp_ski_common.h) — `GridInterpolationKernel` wraps whatever `_map_to_kernel` returned (training_routines.py:157-158)."""
    if kind not in _SKI_KINDS:
        raise ValueError("Unknown kernel type")
    if _SKI_KINDS[kind]:
        gp[3] += 4.0 * _SKI_KINDS[kind]
    return gp


def ski_grid(Z1, Z2=None, grid_size=1024, weights=None, rule="shared", kind="RBF"):
    """Device grid-parameter block covering Z1 (and Z2).
    rule "shared" (this build's default for the additive_rp kinds): ONE regular grid for all projections,
        [g0, h, 1/h, flags, (w_0 .. w_{J-1})];
    rule "reference" (polynomial_projection_kernels.py:54-63, the rp_poly / strictly_additive / additive kinds): a grid
        per projection, [., ., ., flags, w_0 .. w_{J-1}, (g0_j, h_j, 1/h_j) x J].
    `weights` (J per-projection output scales) switches every SKI entry point to the weighted sum (flags |= 1).
    `kind` (RBF | Matern | InverseMQ | Cosine): the wrapped 1-D sub-kernel (flags |= 4 * RPGP_KIND_*).
    float64 coordinates get the float64 twin of the block (the float64 parity path, rpgp_ski_f64.hip)."""
    if rule not in ("shared", "reference"):
        raise ValueError("unknown SKI grid rule %r (shared | reference)" % (rule,))
    if Z1.dtype == torch.float64:
        return _ski_set_kind(_ski64_grid(Z1, Z2, grid_size, weights, rule), kind)
    lib = _lib.load()
    Z1 = _require(Z1, "Z1", 2)
    N1, J = Z1.shape
    if rule not in ("shared", "reference"):
        raise ValueError("unknown SKI grid rule %r (shared | reference)" % (rule,))
    per_proj = rule == "reference"
    gp = torch.empty(4 + 4 * J if per_proj else (4 if weights is None else 4 + J), dtype=torch.float32, device=Z1.device)
    fn = lib.rpgp_ski_grid_per_projection if per_proj else lib.rpgp_ski_grid
    with _on(Z1.device):
        ws = _workspace(Z1.device, lib.rpgp_ski_workspace_bytes(J, grid_size, 1))
        if Z2 is None:
            rc = fn(Z1.data_ptr(), N1, J, None, 0, 0, J, grid_size, gp.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        else:
            Z2 = _require(Z2, "Z2", 2)
            rc = fn(Z1.data_ptr(), N1, J, Z2.data_ptr(), Z2.shape[0], Z2.shape[1], J, grid_size, gp.data_ptr(), ws.data_ptr(),
                    ws.numel(), _stream())
        _lib.check(rc, "rpgp_ski_grid")
    if weights is not None:
        w = weights.detach().to(device=Z1.device, dtype=torch.float32).reshape(-1)
        if w.numel() != J:
            raise ValueError("weights must have one entry per projection (%d)" % J)
        gp[3] = 3.0 if per_proj else 1.0
        gp[4:4 + J] = w
    return _ski_set_kind(gp, kind)


class SkiPlan:
    """rpgp_ski_plan of one Z (and grid block): every projection's points sorted by interpolation cell, built once per
    hyper-parameter step; the products of the CG solve then scatter without atomics (rpgp_ski_mvm_planned)."""

    def __init__(self, Z, gp, grid_size=1024):
        lib = _lib.load()
        Z = _require(Z, "Z", 2)
        self.N, self.J = Z.shape
        self.G = int(grid_size)
        self.device = Z.device
        self.buf = None
        nbytes = lib.rpgp_ski_plan_bytes(self.N, self.J, self.G) if self.N > 0 else 0
        if nbytes == 0:
            return
        self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=Z.device)
        with _on(Z.device):
            ws = _workspace(Z.device, lib.rpgp_ski_plan_workspace_bytes(self.N, self.J, self.G))
            _lib.check(lib.rpgp_ski_plan(Z.data_ptr(), gp.data_ptr(), self.N, Z.stride(0), self.J, self.G, self.buf.data_ptr(),
                                         self.buf.numel(), ws.data_ptr(), ws.numel(), _stream()), "rpgp_ski_plan")

    @property
    def ok(self):
        return self.buf is not None


def ski_plan(Z, gp, grid_size=1024):
    return SkiPlan(Z, gp, grid_size)


def ski_mvm(Z1, Z2, gp, V, scale, noise=0.0, grid_size=1024, plan=None):
    """out = scale * sum_j W1_j Tm W2_j^T V (+ noise V);  Z2 may be Z1 (square operator).  `plan` (SkiPlan of Z1, square
    operator only): the planned product for blocks of up to 12 columns."""
    if Z1.dtype == torch.float64:
        return _ski64_mvm(Z1, Z2, gp, V, scale, noise, grid_size)
    lib = _lib.load()
    Z1 = _require(Z1, "Z1", 2)
    Z2 = _require(Z2, "Z2", 2)
    M, J = Z1.shape
    N = Z2.shape[0]
    V2, squeeze = _as_matrix(V, N, "V")
    T = V2.shape[1]
    out = torch.empty((M, T), dtype=torch.float32, device=Z1.device)
    if plan is not None and plan.ok and T <= 12 and M == N and plan.N == N and plan.J == J and plan.G == grid_size:
        with _on(Z1.device):
            ws = _workspace(Z1.device, lib.rpgp_ski_workspace_bytes(J, grid_size, T))
            rc = lib.rpgp_ski_mvm_planned(plan.buf.data_ptr(), Z1.data_ptr(), gp.data_ptr(), V2.data_ptr(), out.data_ptr(), N,
                                          J, J, grid_size, T, float(scale), float(noise), ws.data_ptr(), ws.numel(), _stream())
        if rc != _lib.RPGP_EWORKSPACE:              # (too many points for the planned form: the plain product below)
            _lib.check(rc, "rpgp_ski_mvm_planned")
            return out.squeeze(1) if squeeze else out
    with _on(Z1.device):
        ws = _workspace(Z1.device, lib.rpgp_ski_workspace_bytes(J, grid_size, T))
        _lib.check(lib.rpgp_ski_mvm(Z1.data_ptr(), Z2.data_ptr(), gp.data_ptr(), V2.data_ptr(), out.data_ptr(), M, N,
                                    J, J, J, grid_size, T, float(scale), float(noise), ws.data_ptr(), ws.numel(),
                                    _stream()), "rpgp_ski_mvm")
    return out.squeeze(1) if squeeze else out


def ski_grid_from_range(zmin, zmax, grid_size, device, weights=None, kind="RBF", dtype=torch.float32):
    """The grid-parameter block of rpgp_ski_grid for a KNOWN coordinate range (row-sharded SKI: the range is all-reduced
    over the ranks first): h = range / (G - 5), g0 = zmin - 2 h, so that every 4-tap stencil is interior.  `dtype`
    float64: the float64 twin of the block (`--double`, the rule of _ski64_grid)."""
    import numpy as np
    ft = np.float64 if dtype == torch.float64 else np.float32
    mn, mx = ft(zmin), ft(zmax)
    rng = ft(mx - mn)
    if not rng > ft(1e-12):
        rng = ft(1e-12)
    h = ft(rng / ft(grid_size - 5))
    head = [float(ft(mn - ft(2.0) * h)), float(h), float(ft(1.0) / h), 0.0 if weights is None else 1.0]
    gp = torch.tensor(head, dtype=dtype, device=device)
    if weights is not None:
        gp = torch.cat([gp, weights.detach().reshape(-1).to(device=device, dtype=dtype)])
    return _ski_set_kind(gp, kind)


def ski_scatter(Z, gp, V, grid_size=1024, plan=None):
    """Stage 1 of the SKI MVM: hist[j][g][t] (float64, J x G x T) = sum over the rows of Z of w(z_ij)[g] V[i][t]."""
    lib = _lib.load()
    if Z.dtype == torch.float64:
        return _ski64_scatter(Z, gp, V, grid_size)
    Z = _require(Z, "Z", 2)
    N, J = Z.shape
    V2, _ = _as_matrix(V, N, "V")
    T = V2.shape[1]
    hist = torch.empty((J, grid_size, T), dtype=torch.float64, device=Z.device)
    if plan is not None and plan.ok and T <= 12 and plan.N == N and plan.J == J and plan.G == grid_size:
        with _on(Z.device):
            ws = _workspace(Z.device, lib.rpgp_ski_workspace_bytes(J, grid_size, T))
            rc = lib.rpgp_ski_scatter_planned(plan.buf.data_ptr(), V2.data_ptr(), hist.data_ptr(), N, J, grid_size, T,
                                              ws.data_ptr(), ws.numel(), _stream())
        if rc != _lib.RPGP_EWORKSPACE:
            _lib.check(rc, "rpgp_ski_scatter_planned")
            return hist
    with _on(Z.device):
        ws = _workspace(Z.device, lib.rpgp_ski_workspace_bytes(J, grid_size, T))
        _lib.check(lib.rpgp_ski_scatter(Z.data_ptr(), gp.data_ptr(), V2.data_ptr(), hist.data_ptr(), N, J, J, grid_size,
                                        T, ws.data_ptr(), ws.numel(), _stream()), "rpgp_ski_scatter")
    return hist


def ski_grid_product(hist, gp, grid_size=1024):
    """Stage 2: H[j][m][t] (float32) = w_j * sum_m' Tm[m, m'] hist[j][m'][t]  (float64 Toeplitz product on the matrix cores)."""
    lib = _lib.load()
    if hist.dtype != torch.float64 or not hist.is_cuda or hist.dim() != 3:
        raise TypeError("hist must be a float64 J x G x T tensor on a HIP device")
    hist = hist.contiguous()
    J, G, T = hist.shape
    if gp.dtype == torch.float64:                 # the float64 twin of the grid block: the float64 parity path
        H = torch.empty((J, G, T), dtype=torch.float64, device=hist.device)
        with _on(hist.device):
            _lib.check(lib.rpgp_ski_f64_grid_product(hist.data_ptr(), gp.data_ptr(), H.data_ptr(), J, G, T, 1, _stream()),
                       "rpgp_ski_f64_grid_product")
        return H
    H = torch.empty((J, G, T), dtype=torch.float32, device=hist.device)
    with _on(hist.device):
        _lib.check(lib.rpgp_ski_grid_product(hist.data_ptr(), gp.data_ptr(), H.data_ptr(), J, G, T, _stream()),
                   "rpgp_ski_grid_product")
    return H


def ski_gather(Z, gp, H, V, scale, noise=0.0, grid_size=1024, plan=None):
    """Stage 3: out[i][t] = scale * sum_j sum_k w_k(z_ij) H[j][idx0 + k][t] + noise * V[i][t] for the rows of Z."""
    lib = _lib.load()
    if Z.dtype == torch.float64:
        return _ski64_gather(Z, gp, H, V, scale, noise, grid_size)
    Z = _require(Z, "Z", 2)
    M, J = Z.shape
    H = _require(H, "H", 3)
    T = H.shape[2]
    V2 = None
    if noise:
        V2, _ = _as_matrix(V, M, "V")
    out = torch.empty((M, T), dtype=torch.float32, device=Z.device)
    with _on(Z.device):
        pl = plan.buf.data_ptr() if (plan is not None and plan.ok and plan.N == M and plan.J == J and plan.G == grid_size) else None
        _lib.check(lib.rpgp_ski_gather_fast(pl, Z.data_ptr(), gp.data_ptr(), H.data_ptr(), None if V2 is None else V2.data_ptr(),
                                            out.data_ptr(), M, J, J, grid_size, T, float(scale), float(noise), _stream()),
                   "rpgp_ski_gather_fast")
    return out


def ski_pivoted_cholesky(Z, gp, scale, rank, grid_size=1024):
    """Pivoted Cholesky of the SKI operator: L (N x rank)."""
    lib = _lib.load()
    Z = _require(Z, "Z", 2)
    N, J = Z.shape
    L = torch.empty((N, rank), dtype=torch.float32, device=Z.device)
    # (residual diagonal + argmax partials and, for large N, the factor in column-major order while it is built)
    work = torch.empty(lib.rpgp_ski_pivoted_cholesky_work_floats(N, int(rank)), dtype=torch.float32, device=Z.device)
    with _on(Z.device):
        _lib.check(lib.rpgp_ski_pivoted_cholesky(Z.data_ptr(), gp.data_ptr(), L.data_ptr(), work.data_ptr(), work.numel(), N, J, J,
                                                 int(grid_size), int(rank), float(scale), _stream()),
                   "rpgp_ski_pivoted_cholesky")
    return L


def ski_dense(Z1, Z2, gp, scale, grid_size=1024):
    """Dense block K_ski(Z1, Z2) (M x N) of the SKI operator."""
    if Z1.dtype == torch.float64:
        return _ski64_dense(Z1, Z2, gp, scale, grid_size)
    lib = _lib.load()
    Z1 = _require(Z1, "Z1", 2)
    Z2 = _require(Z2, "Z2", 2)
    M, J = Z1.shape
    N = Z2.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=Z1.device)
    with _on(Z1.device):
        _lib.check(lib.rpgp_ski_dense(Z1.data_ptr(), Z2.data_ptr(), gp.data_ptr(), out.data_ptr(), M, N, J, J, N, J,
                                      int(grid_size), float(scale), _stream()), "rpgp_ski_dense")
    return out


def ski_diag(Z, gp, scale, grid_size=1024):
    if Z.dtype == torch.float64:
        return _ski64_diag(Z, gp, scale, grid_size)
    lib = _lib.load()
    Z = _require(Z, "Z", 2)
    N, J = Z.shape
    out = torch.empty(N, dtype=torch.float32, device=Z.device)
    with _on(Z.device):
        _lib.check(lib.rpgp_ski_diag(Z.data_ptr(), gp.data_ptr(), out.data_ptr(), N, J, J, grid_size, float(scale),
                                     _stream()), "rpgp_ski_diag")
    return out


def ski_bilinear_grad(Z, gp, L, R, scale, grid_size=1024):
    """(gZ, gscale) of sum((L R^T) * K_ski(Z,Z)); wide blocks are processed 12 columns at a time."""
    if Z.dtype == torch.float64:
        return _ski64_bilinear(Z, gp, L, R, scale, grid_size)[:2]
    lib = _lib.load()
    Z = _require(Z, "Z", 2)
    N, J = Z.shape
    L2, _ = _as_matrix(L, N, "L")
    R2, _ = _as_matrix(R, N, "R")
    T = L2.shape[1]
    gZ = torch.zeros((N, J), dtype=torch.float32, device=Z.device)
    gs = torch.zeros((), dtype=torch.float32, device=Z.device)
    scratch = torch.empty(N, dtype=torch.float32, device=Z.device)
    gZp = torch.empty_like(gZ)
    gsp = torch.empty_like(gs)
    with _on(Z.device):
        ws = _workspace(Z.device, lib.rpgp_ski_workspace_bytes(J, grid_size, min(T, 12)))
        for t0 in range(0, T, 12):
            Lc = L2[:, t0:t0 + 12].contiguous()
            Rc = R2[:, t0:t0 + 12].contiguous()
            _lib.check(lib.rpgp_ski_bilinear_grad(Z.data_ptr(), gp.data_ptr(), Lc.data_ptr(), Rc.data_ptr(),
                                                  gZp.data_ptr(), gsp.data_ptr(), N, J, J, J, grid_size, Lc.shape[1],
                                                  float(scale), ws.data_ptr(), ws.numel(), scratch.data_ptr(),
                                                  _stream()), "rpgp_ski_bilinear_grad")
            gZ += gZp
            gs += gsp
    return gZ, gs


def ski_bilinear_grad_comp(Z, gp, L, R, scale, grid_size=1024):
    """(gZ, gscale, gcomp [J]) for the weighted SKI operator: gcomp[j] is projection j's part of gscale (it carries w_j)."""
    if Z.dtype == torch.float64:
        return _ski64_bilinear(Z, gp, L, R, scale, grid_size)
    lib = _lib.load()
    Z = _require(Z, "Z", 2)
    N, J = Z.shape
    L2, _ = _as_matrix(L, N, "L")
    R2, _ = _as_matrix(R, N, "R")
    T = L2.shape[1]
    gZ = torch.zeros((N, J), dtype=torch.float32, device=Z.device)
    gs = torch.zeros((), dtype=torch.float32, device=Z.device)
    gc = torch.zeros(J, dtype=torch.float32, device=Z.device)
    scratch = torch.empty(N * (J + 1), dtype=torch.float32, device=Z.device)
    gZp, gsp, gcp = torch.empty_like(gZ), torch.empty_like(gs), torch.empty_like(gc)
    with _on(Z.device):
        ws = _workspace(Z.device, lib.rpgp_ski_workspace_bytes(J, grid_size, min(T, 12)))
        for t0 in range(0, T, 12):
            Lc = L2[:, t0:t0 + 12].contiguous()
            Rc = R2[:, t0:t0 + 12].contiguous()
            _lib.check(lib.rpgp_ski_bilinear_grad_comp(Z.data_ptr(), gp.data_ptr(), Lc.data_ptr(), Rc.data_ptr(),
                                                       gZp.data_ptr(), gsp.data_ptr(), gcp.data_ptr(), N, J, J, J,
                                                       grid_size, Lc.shape[1], float(scale), ws.data_ptr(), ws.numel(),
                                                       scratch.data_ptr(), _stream()), "rpgp_ski_bilinear_grad_comp")
            gZ += gZp
            gs += gsp
            gc += gcp
    return gZ, gs, gc


def ski_bilinear_scatter(Z, gp, L, R, grid_size=1024, plan=None):
    """Stage 1 of the SKI derivative: hist2 (J x G x 2T float64) = [W^T L | W^T R] over the rows of Z (T <= 12; zeros for
    an empty row block).  With the plan of Z the scatters are the cell-sorted, atomics-free ones of the planned product."""
    lib = _lib.load()
    N, J = Z.shape
    T = L.shape[1]
    if T > 12:
        raise ValueError("ski_bilinear_scatter takes at most 12 columns per call")
    hist = torch.zeros((J, grid_size, 2 * T), dtype=torch.float64, device=Z.device)
    if N == 0:
        return hist
    if Z.dtype == torch.float64:
        Zc, Lc, Rc = (_require(t, n, 2, allow64=True) for t, n in ((Z, "Z"), (L, "L"), (R, "R")))
        with _on(Z.device):
            for src, off in ((Lc, 0), (Rc, T)):
                _lib.check(lib.rpgp_ski_f64_scatter(Zc.data_ptr(), gp.data_ptr(), src.data_ptr(), hist.data_ptr(), N, J, J,
                                                    grid_size, T, 2 * T, off, 0, _stream()), "rpgp_ski_f64_scatter")
        return hist
    Z = _require(Z, "Z", 2)
    L2, _ = _as_matrix(L, N, "L")
    R2, _ = _as_matrix(R, N, "R")
    if plan is not None and plan.ok and plan.N == N and plan.J == J and plan.G == grid_size:
        with _on(Z.device):
            ws = _workspace(Z.device, lib.rpgp_ski_workspace_bytes(J, grid_size, T))
            rc = lib.rpgp_ski_bilinear_scatter_planned(plan.buf.data_ptr(), L2.data_ptr(), R2.data_ptr(), hist.data_ptr(), N,
                                                       J, grid_size, T, ws.data_ptr(), ws.numel(), _stream())
        if rc != _lib.RPGP_EWORKSPACE:
            _lib.check(rc, "rpgp_ski_bilinear_scatter_planned")
            return hist
    with _on(Z.device):
        ws = _workspace(Z.device, lib.rpgp_ski_workspace_bytes(J, grid_size, T))
        _lib.check(lib.rpgp_ski_bilinear_scatter(Z.data_ptr(), gp.data_ptr(), L2.data_ptr(), R2.data_ptr(), hist.data_ptr(),
                                                 N, J, J, grid_size, T, ws.data_ptr(), ws.numel(), _stream()),
                   "rpgp_ski_bilinear_scatter")
    return hist


def ski_bilinear_finish(Z, gp, hist2, L, R, scale, grid_size=1024, comp=False):
    """Stage 2: (gZ [local rows], gscale partial, gcomp partial or None) from the ALL-REDUCED histogram."""
    lib = _lib.load()
    N, J = Z.shape
    T = L.shape[1]
    gZ = torch.zeros((N, J), dtype=Z.dtype, device=Z.device)
    gs = torch.zeros((), dtype=Z.dtype, device=Z.device)
    gc = torch.zeros(J, dtype=Z.dtype, device=Z.device) if comp else None
    if N == 0:
        return gZ, gs, gc
    if Z.dtype == torch.float64:
        Zc, Lc, Rc = (_require(t, n, 2, allow64=True) for t, n in ((Z, "Z"), (L, "L"), (R, "R")))
        hist2 = hist2.contiguous()
        with _on(Z.device):
            ws = _workspace(Z.device, 8 * J * grid_size * 2 * T + 8 * J + 1024)
            _lib.check(lib.rpgp_ski_f64_bilinear_finish(Zc.data_ptr(), gp.data_ptr(), hist2.data_ptr(), Lc.data_ptr(), Rc.data_ptr(),
                                                        gZ.data_ptr(), gs.data_ptr(), None if gc is None else gc.data_ptr(), N, J,
                                                        J, J, grid_size, T, float(scale), ws.data_ptr(), ws.numel(), _stream()),
                       "rpgp_ski_f64_bilinear_finish")
        return gZ, gs, gc
    Z = _require(Z, "Z", 2)
    L2, _ = _as_matrix(L, N, "L")
    R2, _ = _as_matrix(R, N, "R")
    hist2 = hist2.contiguous()
    scratch = torch.empty(N * (J + 1) if comp else N, dtype=torch.float32, device=Z.device)
    with _on(Z.device):
        ws = _workspace(Z.device, lib.rpgp_ski_workspace_bytes(J, grid_size, T))
        _lib.check(lib.rpgp_ski_bilinear_finish(Z.data_ptr(), gp.data_ptr(), hist2.data_ptr(), L2.data_ptr(), R2.data_ptr(),
                                                gZ.data_ptr(), gs.data_ptr(), None if gc is None else gc.data_ptr(), N, J, J,
                                                J, grid_size, T, float(scale), ws.data_ptr(), ws.numel(),
                                                scratch.data_ptr(), _stream()), "rpgp_ski_bilinear_finish")
    return gZ, gs, gc


# ------------------------------------------------------------------------------- Woodbury preconditioner pieces

def _rows_fp32(A, name):
    """fp32 device matrix with unit column stride (any row stride >= columns): (tensor, leading dimension)."""
    if A.dtype != torch.float32 or not A.is_cuda or A.dim() != 2:
        raise TypeError("%s must be a 2-D float32 HIP tensor" % name)
    if A.shape[1] > 1 and A.stride(1) != 1 or (A.shape[0] > 1 and A.stride(0) < A.shape[1]):
        A = A.contiguous()
    return A, (A.stride(0) if A.shape[0] > 1 else A.shape[1])


def gram_f64(A, B):
    """A^T B (K x T, float64) for tall fp32 A (N x K), B (N x T): exact products, float64 sums, fixed order
    (rpgp_gram_f64).  K, T <= 64."""
    lib = _lib.load()
    A, lda = _rows_fp32(A, "A")
    B, ldb = _rows_fp32(B, "B")
    N, K = A.shape
    T = B.shape[1]
    if B.shape[0] != N or K > 64 or T > 64 or K == 0 or T == 0:
        raise ValueError("gram_f64: N x K and N x T with 1 <= K, T <= 64 required")
    out = torch.empty((K, T), dtype=torch.float64, device=A.device)
    with _on(A.device):
        ws = _workspace(A.device, lib.rpgp_gram_f64_workspace_bytes(K, T))
        _lib.check(lib.rpgp_gram_f64(A.data_ptr(), lda, B.data_ptr(), ldb, N, K, T, out.data_ptr(), ws.data_ptr(),
                                     ws.numel(), _stream()), "rpgp_gram_f64")
    return out


def woodbury_apply(L, R, Tm, noise):
    """(R - L Tm) / noise with float64 arithmetic, fp32 in and out (rpgp_woodbury_apply).  L: N x K, R: N x T, Tm: K x T
    float64."""
    lib = _lib.load()
    L, ldl = _rows_fp32(L, "L")
    R, ldr = _rows_fp32(R, "R")
    N, K = L.shape
    T = R.shape[1]
    if R.shape[0] != N or tuple(Tm.shape) != (K, T) or Tm.dtype != torch.float64 or K > 64 or T > 64:
        raise ValueError("woodbury_apply: L N x K, R N x T, Tm K x T float64 with K, T <= 64 required")
    Tm = Tm.contiguous()
    out = torch.empty((N, T), dtype=torch.float32, device=L.device)
    with _on(L.device):
        _lib.check(lib.rpgp_woodbury_apply(L.data_ptr(), ldl, R.data_ptr(), ldr, Tm.data_ptr(), float(noise),
                                           out.data_ptr(), T, N, K, T, _stream()), "rpgp_woodbury_apply")
    return out


def woodbury_solve(L, R, cinv, noise):
    """M^-1 R = (R - L (C^-1 (L^T R))) / noise for M = L L^T + noise I in two launches: the float64 Gram product L^T R
    (rpgp_gram_f64) and rpgp_woodbury_apply_cinv, which forms C^-1 (L^T R) itself."""
    lib = _lib.load()
    g = gram_f64(L, R)
    L, ldl = _rows_fp32(L, "L")
    R, ldr = _rows_fp32(R, "R")
    N, K = L.shape
    T = R.shape[1]
    if tuple(cinv.shape) != (K, K) or cinv.dtype != torch.float64 or not cinv.is_contiguous():
        raise ValueError("woodbury_solve: cinv must be a contiguous K x K float64 matrix")
    out = torch.empty((N, T), dtype=torch.float32, device=L.device)
    with _on(L.device):
        _lib.check(lib.rpgp_woodbury_apply_cinv(L.data_ptr(), ldl, R.data_ptr(), ldr, g.data_ptr(), cinv.data_ptr(), float(noise),
                                                out.data_ptr(), T, N, K, T, _stream()), "rpgp_woodbury_apply_cinv")
    return out


def woodbury_setup(gram, noise, logdet_pinned=None):
    """(chol, cinv, logdet) of C = gram + noise I for a K x K float64 device matrix, K <= 64, in one launch
    (rpgp_woodbury_setup); logdet is a 1-element device tensor (no synchronisation here).  `logdet_pinned`: a 1-element pinned
    float64 HOST tensor the kernel writes log|C| to as well (read it once the stream has passed this launch)."""
    lib = _lib.load()
    if gram.dtype != torch.float64 or not gram.is_cuda or gram.dim() != 2 or gram.shape[0] != gram.shape[1] or \
            gram.shape[0] > 64:
        raise TypeError("gram must be a square float64 HIP matrix with at most 64 rows")
    gram = gram.contiguous()
    K = gram.shape[0]
    out = torch.empty((2 * K * K + 1,), dtype=torch.float64, device=gram.device)
    chol, cinv, logdet = out[:K * K].view(K, K), out[K * K:2 * K * K].view(K, K), out[2 * K * K:]
    with _on(gram.device):
        _lib.check(lib.rpgp_woodbury_setup_pinned(gram.data_ptr(), float(noise), K, chol.data_ptr(), cinv.data_ptr(),
                                                  logdet.data_ptr(),
                                                  logdet_pinned.data_ptr() if logdet_pinned is not None else None, _stream()),
                   "rpgp_woodbury_setup")
    return chol, cinv, logdet


# ------------------------------------------------------------------------------------------------ native mBCG

# ------------------------------------------------------------------------------------ generalised family

KINDS = {"RBF": _lib.RPGP_KIND_RBF, "Matern": _lib.RPGP_KIND_MATERN15, "InverseMQ": _lib.RPGP_KIND_IMQ,
         "Cosine": _lib.RPGP_KIND_COSINE}
FAMILY_GROUPS = (1, 2, 3, 4, 5, 8, 10, 20)


class Family:
    """`struct rpgp_family`: kind (name of training_routines.py:47-88 or RPGP_KIND_*), group size, per-component
    weights (device tensor, kept alive here).  `generic`: served by the runtime-(kind, group) kernels of
    csrc/rpgp_family_generic.hip instead of the templated fast kernels — float64 weights (`--double`), k > 1 sub-kernels of
    the non-RBF types, group sizes that are not instantiated.  `product`: a k > 1 group is the product of k 1-D sub-kernels
    (the ProductKernel groups of polynomial_projection_kernels.py:70-86) instead of one radial k-dimensional sub-kernel
    (training_routines.py:172-174); the same function for the RBF, so the flag only changes the other kinds."""

    def __init__(self, kind, group, weights, product=False):
        import ctypes
        self.kind = KINDS[kind] if isinstance(kind, str) else int(kind)
        self.group = int(group)
        self.product = bool(product) and self.group > 1 and self.kind != _lib.RPGP_KIND_RBF
        w = weights.detach().reshape(-1)
        self.dtype = torch.float64 if w.dtype == torch.float64 else torch.float32
        if not w.is_cuda:
            raise TypeError("weights must live on a HIP device (there is no CPU fallback)")
        self.weights = w.to(self.dtype).contiguous()
        self.ncomp = self.weights.numel()
        fast = self.group in FAMILY_GROUPS and not (self.group > 1 and self.kind != _lib.RPGP_KIND_RBF)
        self.generic = self.dtype == torch.float64 or not fast
        if self.generic and (self.group < 1 or self.group > 32 or self.group * self.ncomp > 64):
            raise ValueError("unsupported family member: %d components of %d-dimensional sub-kernels (the generic kernels take "
                             "groups of at most 32 and 64 columns in all)" % (self.ncomp, self.group))
        if not self.generic:
            self.struct = _lib.RpgpFamily(self.kind, self.group, self.ncomp, self.weights.data_ptr())
            self.ref = ctypes.byref(self.struct)
        else:
            self.struct = self.ref = None
        self.code = _lib.RPGP_F64 if self.dtype == torch.float64 else _lib.RPGP_F32
        # what the runtime-(kind, group) entry points take as `kind`
        self.generic_kind = self.kind | (_lib.RPGP_KIND_PRODUCT if self.product else 0)

    @property
    def ncols(self):
        return self.ncomp * self.group


def _family_cols(fam, Z, name):
    if fam.generic:
        if not (torch.is_tensor(Z) and Z.is_cuda and Z.dim() == 2 and Z.dtype == fam.dtype):
            raise TypeError("%s must be a 2-D %s tensor on a HIP device" % (name, fam.dtype))
        Z = Z.contiguous()
    else:
        Z = _require(Z, name, 2)
    if Z.shape[1] != fam.ncols:
        raise ValueError("%s must have ncomp * group = %d columns (got %d)" % (name, fam.ncols, Z.shape[1]))
    return Z


def _generic_matrix(fam, V, rows, name):
    squeeze = V.dim() == 1
    V2 = (V.reshape(-1, 1) if squeeze else V).to(fam.dtype).contiguous()
    if V2.shape[0] != rows or not V2.is_cuda:
        raise ValueError("%s must have %d rows on the device" % (name, rows))
    return V2, squeeze


def _generic_mvm(fam, Z1, Z2, V, scale, noise):
    lib = _lib.load()
    M, J = Z1.shape
    N = M if Z2 is None else Z2.shape[0]
    V2, squeeze = _generic_matrix(fam, V, N, "V")
    out = torch.empty((M, V2.shape[1]), dtype=fam.dtype, device=Z1.device)
    with _on(Z1.device):
        for t0 in range(0, V2.shape[1], 16):
            Vc = V2[:, t0:t0 + 16].contiguous()
            oc = torch.empty((M, Vc.shape[1]), dtype=fam.dtype, device=Z1.device)
            _lib.check(lib.rpgp_family_generic_mvm(fam.code, fam.generic_kind, fam.group, fam.ncomp, fam.weights.data_ptr(), Z1.data_ptr(),
                                                   None if Z2 is None else Z2.data_ptr(), Vc.data_ptr(), oc.data_ptr(), M, N, J, J,
                                                   Vc.shape[1], float(scale), float(noise), _stream()),
                       "rpgp_family_generic_mvm")
            out[:, t0:t0 + 16] = oc
    return out.squeeze(1) if squeeze else out


def family_mvm_sym(fam, Z, V, scale, noise=0.0):
    """out = scale * sum_c w_c K_c(Z,Z) @ V + noise * V for a member of the generalised family."""
    lib = _lib.load()
    Z = _family_cols(fam, Z, "Z")
    if fam.generic:
        return _generic_mvm(fam, Z, None, V, scale, noise)
    N, J = Z.shape
    V2, squeeze = _as_matrix(V, N, "V")
    T = V2.shape[1]
    out = torch.empty_like(V2)
    with _on(Z.device):
        ws = _workspace(Z.device, lib.rpgp_family_mvm_workspace_bytes(N, N, T, 1))
        _lib.check(lib.rpgp_family_mvm_sym(fam.ref, Z.data_ptr(), V2.data_ptr(), out.data_ptr(), N, J, T, float(scale),
                                           float(noise), ws.data_ptr(), ws.numel(), _stream()), "rpgp_family_mvm_sym")
    return out.squeeze(1) if squeeze else out


def family_mvm_rect(fam, Z1, Z2, V, scale):
    """out (M x T) = scale * sum_c w_c K_c(Z1, Z2) @ V."""
    lib = _lib.load()
    Z1 = _family_cols(fam, Z1, "Z1")
    Z2 = _family_cols(fam, Z2, "Z2")
    if fam.generic:
        return _generic_mvm(fam, Z1, Z2, V, scale, 0.0)
    M, J = Z1.shape
    N = Z2.shape[0]
    V2, squeeze = _as_matrix(V, N, "V")
    T = V2.shape[1]
    out = torch.empty((M, T), dtype=torch.float32, device=Z1.device)
    with _on(Z1.device):
        ws = _workspace(Z1.device, lib.rpgp_family_mvm_workspace_bytes(M, N, T, 0))
        _lib.check(lib.rpgp_family_mvm_rect(fam.ref, Z1.data_ptr(), Z2.data_ptr(), V2.data_ptr(), out.data_ptr(), M, N,
                                            J, J, T, float(scale), ws.data_ptr(), ws.numel(), _stream()),
                   "rpgp_family_mvm_rect")
    return out.squeeze(1) if squeeze else out


def family_dense(fam, Z1, Z2, scale):
    """Dense block K(Z1, Z2) (M x N) of a family member."""
    lib = _lib.load()
    Z1 = _family_cols(fam, Z1, "Z1")
    Z2 = _family_cols(fam, Z2, "Z2")
    M, J = Z1.shape
    N = Z2.shape[0]
    out = torch.empty((M, N), dtype=fam.dtype, device=Z1.device)
    with _on(Z1.device):
        if fam.generic:
            _lib.check(lib.rpgp_family_generic_dense(fam.code, fam.generic_kind, fam.group, fam.ncomp, fam.weights.data_ptr(),
                                                     Z1.data_ptr(), Z2.data_ptr(), out.data_ptr(), M, N, J, J, N, float(scale),
                                                     _stream()), "rpgp_family_generic_dense")
        else:
            _lib.check(lib.rpgp_family_dense(fam.ref, Z1.data_ptr(), Z2.data_ptr(), out.data_ptr(), M, N, J, J, N,
                                             float(scale), _stream()), "rpgp_family_dense")
    return out


def _generic_bilinear(fam, Z, L, R, S, scale):
    lib = _lib.load()
    N, J = Z.shape
    gZ = torch.zeros((N, J), dtype=fam.dtype, device=Z.device)
    gc = torch.zeros(fam.ncomp, dtype=fam.dtype, device=Z.device)
    with _on(Z.device):
        ws = _workspace(Z.device, lib.rpgp_family_generic_bilinear_workspace_bytes(fam.code, N, fam.ncomp))
        if S is not None:
            S = S.to(fam.dtype).contiguous()
            _lib.check(lib.rpgp_family_generic_bilinear(fam.code, fam.generic_kind, fam.group, fam.ncomp, fam.weights.data_ptr(),
                                                        Z.data_ptr(), None, None, S.data_ptr(), gZ.data_ptr(), gc.data_ptr(), N, J,
                                                        J, 0, N, float(scale), ws.data_ptr(), ws.numel(), _stream()),
                       "rpgp_family_generic_bilinear")
            return gZ, gc
        gZp, gcp = torch.empty_like(gZ), torch.empty_like(gc)
        for t0 in range(0, L.shape[1], 16):      # the derivative is additive over the columns of L, R
            Lc, Rc = L[:, t0:t0 + 16].contiguous(), R[:, t0:t0 + 16].contiguous()
            _lib.check(lib.rpgp_family_generic_bilinear(fam.code, fam.generic_kind, fam.group, fam.ncomp, fam.weights.data_ptr(),
                                                        Z.data_ptr(), Lc.data_ptr(), Rc.data_ptr(), None, gZp.data_ptr(),
                                                        gcp.data_ptr(), N, J, J, Lc.shape[1], 0, float(scale), ws.data_ptr(),
                                                        ws.numel(), _stream()), "rpgp_family_generic_bilinear")
            gZ += gZp
            gc += gcp
    return gZ, gc


def family_bilinear_grad(fam, Z, L, R, scale):
    """(gZ [N x cols], gcomp [ncomp]) for sum((L R^T) * K): gcomp are the unweighted per-component sums."""
    lib = _lib.load()
    Z = _family_cols(fam, Z, "Z")
    N, J = Z.shape
    if fam.generic:
        L2, _ = _generic_matrix(fam, L, N, "L")
        R2, _ = _generic_matrix(fam, R, N, "R")
        if L2.shape != R2.shape:
            raise ValueError("L and R must have the same shape")
        return _generic_bilinear(fam, Z, L2, R2, None, scale)
    L2, _ = _as_matrix(L, N, "L")
    R2, _ = _as_matrix(R, N, "R")
    if L2.shape != R2.shape:
        raise ValueError("L and R must have the same shape")
    T = L2.shape[1]
    gZ = torch.zeros((N, J), dtype=torch.float32, device=Z.device)
    gc = torch.zeros(fam.ncomp, dtype=torch.float32, device=Z.device)
    with _on(Z.device):
        ws = _workspace(Z.device, lib.rpgp_family_bilinear_grad_workspace_bytes(N, J, fam.ncomp))
        gZp, gcp = torch.empty_like(gZ), torch.empty_like(gc)
        for t0 in range(0, T, 12):      # the derivative is additive over the columns of L, R
            Lc, Rc = L2[:, t0:t0 + 12].contiguous(), R2[:, t0:t0 + 12].contiguous()
            _lib.check(lib.rpgp_family_bilinear_grad(fam.ref, Z.data_ptr(), Lc.data_ptr(), Rc.data_ptr(), gZp.data_ptr(),
                                                     gcp.data_ptr(), N, J, J, Lc.shape[1], float(scale), ws.data_ptr(),
                                                     ws.numel(), _stream()), "rpgp_family_bilinear_grad")
            gZ += gZp
            gc += gcp
    return gZ, gc


def family_bilinear_grad_dense(fam, Z, S, scale):
    """(gZ, gcomp) for 0.5 * sum(S * K) with an explicit symmetric N x N weight matrix S."""
    lib = _lib.load()
    Z = _family_cols(fam, Z, "Z")
    N, J = Z.shape
    if S.shape != (N, N):
        raise ValueError("S must be %d x %d" % (N, N))
    if fam.generic:
        return _generic_bilinear(fam, Z, None, None, S, scale)
    S = _require(S, "S", 2)
    gZ = torch.zeros((N, J), dtype=torch.float32, device=Z.device)
    gc = torch.zeros(fam.ncomp, dtype=torch.float32, device=Z.device)
    with _on(Z.device):
        ws = _workspace(Z.device, lib.rpgp_family_bilinear_grad_workspace_bytes(N, J, fam.ncomp))
        _lib.check(lib.rpgp_family_bilinear_grad_dense(fam.ref, Z.data_ptr(), S.data_ptr(), gZ.data_ptr(), gc.data_ptr(),
                                                       N, J, J, N, float(scale), ws.data_ptr(), ws.numel(), _stream()),
                   "rpgp_family_bilinear_grad_dense")
    return gZ, gc


def make_operator_desc(kind, N, J, scale, noise, Z=None, prep=None, gp=None, j0=0, j1=None, G=0, Kd=None, family=None,
                       symcache=None, world=1, rank=0):
    """Fill a `struct rpgp_operator`; returns (struct, keepalive) — keep both referenced while the solve runs.
    (world, rank): this rank's pair-shard of the fused / prepared operator (a symcache carries its own)."""
    import ctypes
    d = _lib.RpgpOperator()
    d.kind, d.N, d.J, d.ldz = kind, N, J, J
    d.j0, d.j1, d.G = j0, J if j1 is None else j1, G
    d.scale, d.noise = float(scale), float(noise)
    d.Z = Z.data_ptr() if Z is not None else None
    d.prep = prep.buf.data_ptr() if (prep is not None and prep.buf is not None) else None
    d.grid_params = gp.data_ptr() if gp is not None else None
    d.Kd = Kd.data_ptr() if Kd is not None else None
    d.ldk = Kd.stride(0) if Kd is not None else 0
    d.family = ctypes.addressof(family.struct) if family is not None else None
    d.world, d.rank = int(world), int(rank)
    if symcache is not None:                       # RPGP_OP_SYMCACHE: the cache travels in (Kd, ldk = bytes, G = layout)
        d.Kd, d.ldk, d.G = symcache.buf.data_ptr(), symcache.nbytes, symcache.layout
        d.world, d.rank = symcache.world, symcache.rank
    return d, (Z, prep, gp, Kd, family, symcache)


def make_sum_operator_desc(parts):
    """RPGP_OP_SUM: the sum of the given (struct, keepalive) descriptors of unsharded operators on the same rows, as one
    descriptor for the native executor; returns (struct, keepalive)."""
    import ctypes
    if not parts:
        raise ValueError("a sum operator needs at least one part")
    arr = (_lib.RpgpOperator * len(parts))()
    for i, (d, _) in enumerate(parts):
        if d.kind == _lib.RPGP_OP_SUM or d.N != parts[0][0].N:
            raise ValueError("the parts of a sum operator are plain operators on the same rows")
        ctypes.memmove(ctypes.addressof(arr[i]), ctypes.addressof(d), ctypes.sizeof(_lib.RpgpOperator))
    s = _lib.RpgpOperator()
    s.kind, s.N, s.J, s.ldz, s.G = _lib.RPGP_OP_SUM, parts[0][0].N, 0, 0, len(parts)
    s.scale, s.noise = 1.0, 0.0
    s.prep = ctypes.cast(arr, ctypes.c_void_p).value
    return s, (arr, [keep for _, keep in parts])


def mbcg_solve(desc, rhs, tolerance, max_iter, min_iter=10, hist_len=0, check_every=1, L=None, Cinv=None, sigma2=1.0,
               stagnation_window=0, sharding=None):
    """Native preconditioned batched CG (rpgp_mbcg_solve).  rhs: N x T (T <= 16).
    `sharding` = (mode, reducer, global_N): mode "partial" (replicated vectors, the operator's product is this rank's
    partial) or "rows" (SKI, `rhs` / `L` are this rank's rows); reducer: rpgp_amd.distributed.Reducer.
    Returns (x, alpha_hist [h x T], beta_hist [h x T], iterations, mean_residual)."""
    import ctypes
    import numpy as np
    lib = _lib.load()
    if rhs.shape[0] > 0:
        rhs = _require(rhs, "rhs", 2)
    N, T = rhs.shape
    if T > 16:
        raise ValueError("the native mBCG executor handles at most 16 right-hand sides")
    k = 0
    Lp = Cp = None
    if L is not None:
        if N > 0:
            L = _require(L, "L", 2)
        if Cinv.dtype != torch.float64 or not Cinv.is_cuda:
            raise TypeError("Cinv must be a float64 device tensor")
        Cinv = Cinv.contiguous()
        k = L.shape[1]
        if N == 0:
            L = torch.zeros((1, k), dtype=torch.float32, device=Cinv.device)
        Lp, Cp = L.data_ptr(), Cinv.data_ptr()
    x = torch.empty_like(rhs) if N > 0 else torch.empty((0, T), dtype=torch.float32, device=rhs.device)
    rhs_buf, x_buf = (rhs, x) if N > 0 else (torch.zeros((1, T), dtype=torch.float32, device=rhs.device),) * 2
    ah = np.zeros((max(hist_len, 1), 16), dtype=np.float32)
    bh = np.zeros((max(hist_len, 1), 16), dtype=np.float32)
    iters, mres = ctypes.c_int(0), ctypes.c_float(0)
    with _on(rhs.device):
        nbytes = lib.rpgp_mbcg_workspace_bytes(ctypes.byref(desc), T, k)
        ws = _workspace(rhs.device, nbytes)
        red_ref = None
        fn = None
        if sharding is not None:
            mode, reducer, global_N = sharding
            fn, ctx = reducer.c_hook(ws)
        if fn is not None:
            red = _lib.RpgpReducer()
            red.mode = {"partial": _lib.RPGP_SHARD_PARTIAL, "rows": _lib.RPGP_SHARD_ROWS}[mode]
            red.world, red.rank, red.global_N = reducer.world_size, reducer.rank, int(global_N)
            red.fn, red.ctx = fn, ctx
            red_ref = ctypes.byref(red)
        rc = lib.rpgp_mbcg_solve(ctypes.byref(desc), rhs_buf.data_ptr(), x_buf.data_ptr(), T, int(max_iter), int(min_iter),
                                 int(hist_len), int(check_every), int(stagnation_window), float(tolerance), k, Lp, Cp,
                                 float(sigma2), red_ref,
                                 ah.ctypes.data, bh.ctypes.data, ctypes.byref(iters), ctypes.byref(mres),
                                 ws.data_ptr(), ws.numel(), _stream())
    if sharding is not None:
        # the executor has synchronised: the reducer's workspace reference can go, and a bounded wait of the IPC backend
        # that ran out (a peer stalled on the host for longer than the bound) surfaces HERE, not as silently wrong numbers
        sharding[1].release_workspace()
        sharding[1].check()
    if rc == _lib.RPGP_ENUMERIC:
        raise RuntimeError("NaNs encountered when trying to perform matrix-vector multiplication")
    _lib.check(rc, "rpgp_mbcg_solve")
    h = min(hist_len, iters.value)
    return x, ah[:h, :T], bh[:h, :T], iters.value, mres.value        # (views of the [hist x 16] landing arrays)


def slq_logdet_history(alpha_hist, beta_hist, num_probes, n):
    """log|A| estimate from the coefficient histories `mbcg_solve` returns ([iters x 16] float32 host arrays; the first
    `num_probes` columns are unit-norm probes): rpgp_slq_logdet (host arithmetic in the library)."""
    import ctypes
    lib = _lib.load()
    out = ctypes.c_double(0.0)
    _lib.check(lib.rpgp_slq_logdet(alpha_hist.ctypes.data, beta_hist.ctypes.data, int(alpha_hist.shape[0]),
                                   int(alpha_hist.strides[0] // 4), int(num_probes), float(n), ctypes.byref(out)),
               "rpgp_slq_logdet")
    return out.value


# ---- the small kernels around the solve of one optimiser step (csrc/rpgp_step.hip; caller: fused_mll.py) -------------------
def step_hyper(raw_ls, raw_os, raw_noise, mean, W, prescale, min_noise):
    """Raw parameters -> (Peff [d x J], dev [8 + 2 n_ls floats: outputscale, noise, mean, sigmoid(raw_os), sigmoid(raw_noise),
    ..., ls, sigmoid(raw_ls)], outputscale, noise, mean as host floats): rpgp_step_hyper, one launch; the three floats come
    back through pinned memory (no copy, no stream synchronisation)."""
    import ctypes
    lib = _lib.load()
    W = _require(W, "W", 2)
    J, d = W.shape
    n_ls = raw_ls.numel()
    for t in (raw_ls, raw_os, raw_noise, mean):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise TypeError("step_hyper: parameters must be contiguous float32 device tensors")
    Peff = torch.empty((d, J), dtype=torch.float32, device=W.device)
    dev = torch.empty(8 + 2 * n_ls, dtype=torch.float32, device=W.device)
    os_h, nz_h, mu_h = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(0)
    with _on(W.device):
        _lib.check(lib.rpgp_step_hyper(raw_ls.data_ptr(), n_ls, raw_os.data_ptr(), raw_noise.data_ptr(), mean.data_ptr(),
                                       W.data_ptr(), d, J, 1 if prescale else 0, float(min_noise), Peff.data_ptr(),
                                       dev.data_ptr(), ctypes.byref(os_h), ctypes.byref(nz_h), ctypes.byref(mu_h), _stream()),
                   "rpgp_step_hyper")
    return Peff, dev, os_h.value, nz_h.value, mu_h.value


def step_probes(L, e1, e2, sqrt_noise, y, mean_dev):
    """full_rhs [N x (p + 1)] = [L e1 + sqrt_noise e2 | y - mean]: rpgp_step_probes, one launch (the probes stay unnormalised:
    the executor normalises its right-hand sides itself)."""
    lib = _lib.load()
    L = _require(L, "L", 2)
    e1 = _require(e1, "e1", 2)
    e2 = _require(e2, "e2", 2)
    N, k = L.shape
    p = e1.shape[1]
    if e1.shape[0] != k or e2.shape != (N, p) or y.shape != (N,) or y.dtype != torch.float32 or not y.is_contiguous():
        raise ValueError("step_probes: e1 is k x p, e2 is N x p, y has N float32 entries")
    full_rhs = torch.empty((N, p + 1), dtype=torch.float32, device=L.device)
    with _on(L.device):
        _lib.check(lib.rpgp_step_probes(L.data_ptr(), k, e1.data_ptr(), e2.data_ptr(), float(sqrt_noise), y.data_ptr(),
                                        mean_dev.data_ptr(), N, p, full_rhs.data_ptr(), _stream()), "rpgp_step_probes")
    return full_rhs


_value_ws = {}


def step_value(full_rhs, solves, col, logdet, c1, c2, post=False):
    """out[0] = (sum_i full_rhs[i][col] solves[i][col] + logdet) c1 + c2, out[1] = the inner product: rpgp_step_value.
    `post`: returns (out, ticket) — the kernel also posts out[0] to pinned host memory, `step_value_wait(ticket)` reads it."""
    import ctypes
    lib = _lib.load()
    N, T = full_rhs.shape
    out = torch.empty(2, dtype=torch.float32, device=full_rhs.device)
    key = (full_rhs.device.index, int(_stream() or 0))
    ws = _value_ws.get(key)
    if ws is None:              # (zeroed ONCE: the kernel leaves its arrival counter at zero; one buffer per stream)
        ws = _value_ws[key] = torch.zeros(lib.rpgp_step_value_workspace_bytes(), dtype=torch.uint8, device=full_rhs.device)
    with _on(full_rhs.device):
        ticket = ctypes.c_int(-1)
        _lib.check(lib.rpgp_step_value(full_rhs.data_ptr(), solves.data_ptr(), N, T, int(col), float(logdet), float(c1), float(c2),
                                       out.data_ptr(), ws.data_ptr(), ws.numel(), ctypes.byref(ticket) if post else None,
                                       _stream()), "rpgp_step_value")
    return (out, ticket.value) if post else out


def step_value_wait(ticket):
    """The value a `step_value(..., post=True)` call posted to the host, or None when the ticket is stale (16 later posts),
    from another thread, or not there within 2 s — the caller then reads the device tensor."""
    import ctypes
    if ticket is None or ticket < 0:
        return None
    v = ctypes.c_float(0)
    if _lib.load().rpgp_step_value_wait(int(ticket), ctypes.byref(v)) != 0:
        return None
    return v.value


def step_lr(solves, pre_probes, g, gscale):
    """The two sides of the bilinear derivative (rpgp_step_lr): (left, right, partials, nparts).  `pre_probes`: N x p with unit
    column stride (any row stride)."""
    import ctypes
    lib = _lib.load()
    N, T = solves.shape
    p = T - 1
    if pre_probes.shape != (N, p) or pre_probes.stride(1) != 1 or pre_probes.dtype != torch.float32:
        raise ValueError("step_lr: pre_probes must be N x p float32 with unit column stride")
    left = torch.empty_like(solves)
    right = torch.empty_like(solves)
    part = torch.empty(lib.rpgp_step_lr_workspace_bytes() // 4, dtype=torch.float32, device=solves.device)
    nparts = ctypes.c_int(0)
    with _on(solves.device):
        _lib.check(lib.rpgp_step_lr(solves.data_ptr(), pre_probes.data_ptr(), pre_probes.stride(0), g.data_ptr(), float(gscale), N,
                                    p, left.data_ptr(), right.data_ptr(), part.data_ptr(), ctypes.byref(nparts), _stream()),
                   "rpgp_step_lr")
    return left, right, part, nparts.value


def step_hyper_backward(dPeff, W, n_ls, prescale, zfac, hyper_dev, gs, partials, nparts, g, gscale, dlp_over_n, gs_scale=1.0):
    """Gradients of the raw parameters (rpgp_step_hyper_backward): (g_raw_ls [n_ls], g_raw_os, g_raw_noise, g_mean) as views of
    one buffer."""
    lib = _lib.load()
    J, d = W.shape
    out = torch.empty(n_ls + 3, dtype=torch.float32, device=W.device)
    base = out.data_ptr()
    with _on(W.device):
        _lib.check(lib.rpgp_step_hyper_backward(dPeff.data_ptr(), W.data_ptr(), d, J, int(n_ls), 1 if prescale else 0, float(zfac),
                                                hyper_dev.data_ptr(), gs.data_ptr(), partials.data_ptr(), int(nparts),
                                                g.data_ptr(), float(gscale), float(dlp_over_n), float(gs_scale), base, base + 4 * n_ls,
                                                base + 4 * (n_ls + 1), base + 4 * (n_ls + 2), _stream()),
                   "rpgp_step_hyper_backward")
    return out[:n_ls], out[n_ls:n_ls + 1], out[n_ls + 1:n_ls + 2], out[n_ls + 2:n_ls + 3]


# ---- float64 parity path of the SKI operator (csrc/rpgp_ski_f64.hip): `--double` for the `ski: true` specifications --------
def _ski64_scatter(Z, gp, V, grid_size):
    lib = _lib.load()
    Z = _require(Z, "Z", 2, allow64=True)
    N, J = Z.shape
    V2 = V.reshape(N, -1).contiguous()
    if V2.dtype != torch.float64:
        raise TypeError("V must be float64 for float64 coordinates")
    T = V2.shape[1]
    hist = torch.empty((J, grid_size, T), dtype=torch.float64, device=Z.device)
    with _on(Z.device):
        _lib.check(lib.rpgp_ski_f64_scatter(Z.data_ptr(), gp.data_ptr(), V2.data_ptr(), hist.data_ptr(), N, J, J, grid_size, T, T,
                                            0, 1, _stream()), "rpgp_ski_f64_scatter")
    return hist


def _ski64_gather(Z, gp, H, V, scale, noise, grid_size):
    lib = _lib.load()
    Z = _require(Z, "Z", 2, allow64=True)
    M, J = Z.shape
    H = H.contiguous()
    if H.dtype != torch.float64 or H.dim() != 3:
        raise TypeError("H must be the float64 J x G x T grid product for float64 coordinates")
    T = H.shape[2]
    V2 = V.reshape(M, -1).contiguous() if noise else None
    out = torch.empty((M, T), dtype=torch.float64, device=Z.device)
    with _on(Z.device):
        _lib.check(lib.rpgp_ski_f64_gather(Z.data_ptr(), gp.data_ptr(), H.data_ptr(), None if V2 is None else V2.data_ptr(),
                                           out.data_ptr(), M, J, J, grid_size, T, float(scale), float(noise), _stream()),
                   "rpgp_ski_f64_gather")
    return out



def _ski64_grid(Z1, Z2, grid_size, weights, rule):
    """The float64 twin of the grid-parameter block, from the same rules (rpgp_ski_grid / rpgp_ski_grid_per_projection), as
    torch operations on the device (a dozen scalars per hyper-parameter step)."""
    G = int(grid_size)
    J = Z1.shape[1]
    z = Z1 if Z2 is None else torch.cat([Z1, Z2], dim=0)
    if rule == "shared":
        mn, mx = z.amin(), z.amax()
        rng = (mx - mn).clamp_min(1e-12)
        h = rng / (G - 5)
        head = torch.stack([mn - 2.0 * h, h, 1.0 / h, torch.zeros_like(h)])
        if weights is None:
            return head.contiguous()
        w = weights.detach().to(device=Z1.device, dtype=torch.float64).reshape(-1)
        if w.numel() != J:
            raise ValueError("weights must have one entry per projection (%d)" % J)
        head[3] = 1.0
        return torch.cat([head, w]).contiguous()
    mn, mx = z.amin(dim=0), z.amax(dim=0)
    rng = (mx - mn).clamp_min(1e-12)
    spacing = rng / (G - 4)
    b0 = mn - 2.01 * spacing
    h = spacing * ((G - 4) + 4.02) / (G - 1)
    h = torch.maximum(h, torch.maximum(mn.abs(), mx.abs()) * 4.5e-16)
    w = torch.ones(J, dtype=torch.float64, device=Z1.device) if weights is None else \
        weights.detach().to(device=Z1.device, dtype=torch.float64).reshape(-1)
    if w.numel() != J:
        raise ValueError("weights must have one entry per projection (%d)" % J)
    head = torch.tensor([0.0, 1.0, 1.0, 2.0 if weights is None else 3.0], dtype=torch.float64, device=Z1.device)
    return torch.cat([head, w, torch.stack([b0, h, 1.0 / h], dim=1).reshape(-1)]).contiguous()


def _ski64_mvm(Z1, Z2, gp, V, scale, noise, grid_size):
    lib = _lib.load()
    Z1 = _require(Z1, "Z1", 2, allow64=True)
    Z2 = _require(Z2, "Z2", 2, allow64=True)
    M, J = Z1.shape
    N = Z2.shape[0]
    V2, squeeze = _as_matrix(V, N, "V", allow64=True)
    T = V2.shape[1]
    out = torch.empty((M, T), dtype=torch.float64, device=Z1.device)
    same = Z1.data_ptr() == Z2.data_ptr() and M == N
    with _on(Z1.device):
        for t0 in range(0, T, 64):                       # (column pieces bound the J x G x T workspace)
            Vc = V2[:, t0:t0 + 64].contiguous()
            oc = torch.empty((M, Vc.shape[1]), dtype=torch.float64, device=Z1.device)
            ws = _workspace(Z1.device, lib.rpgp_ski_f64_workspace_bytes(J, int(grid_size), Vc.shape[1]))
            _lib.check(lib.rpgp_ski_f64_mvm(Z1.data_ptr(), Z2.data_ptr(), gp.data_ptr(), Vc.data_ptr(), oc.data_ptr(), M, N, J, J,
                                            J, int(grid_size), Vc.shape[1], float(scale), float(noise) if same else 0.0,
                                            ws.data_ptr(), ws.numel(), _stream()), "rpgp_ski_f64_mvm")
            out[:, t0:t0 + 64] = oc
    return out.squeeze(1) if squeeze else out


def _ski64_dense(Z1, Z2, gp, scale, grid_size):
    lib = _lib.load()
    Z1 = _require(Z1, "Z1", 2, allow64=True)
    Z2 = _require(Z2, "Z2", 2, allow64=True)
    M, J = Z1.shape
    N = Z2.shape[0]
    out = torch.empty((M, N), dtype=torch.float64, device=Z1.device)
    with _on(Z1.device):
        _lib.check(lib.rpgp_ski_f64_dense(Z1.data_ptr(), Z2.data_ptr(), gp.data_ptr(), out.data_ptr(), M, N, J, J, N, J,
                                          int(grid_size), float(scale), _stream()), "rpgp_ski_f64_dense")
    return out


def _ski64_diag(Z, gp, scale, grid_size):
    lib = _lib.load()
    Z = _require(Z, "Z", 2, allow64=True)
    N, J = Z.shape
    out = torch.empty(N, dtype=torch.float64, device=Z.device)
    with _on(Z.device):
        _lib.check(lib.rpgp_ski_f64_diag(Z.data_ptr(), gp.data_ptr(), out.data_ptr(), N, J, J, int(grid_size), float(scale),
                                         _stream()), "rpgp_ski_f64_diag")
    return out


def _ski64_bilinear(Z, gp, L, R, scale, grid_size):
    """(gZ, gscale, gcomp) in float64 (rpgp_ski_f64_bilinear_grad), 64 columns per piece."""
    lib = _lib.load()
    Z = _require(Z, "Z", 2, allow64=True)
    N, J = Z.shape
    L2, _ = _as_matrix(L, N, "L", allow64=True)
    R2, _ = _as_matrix(R, N, "R", allow64=True)
    T = L2.shape[1]
    gZ = torch.zeros((N, J), dtype=torch.float64, device=Z.device)
    gs = torch.zeros((), dtype=torch.float64, device=Z.device)
    gc = torch.zeros(J, dtype=torch.float64, device=Z.device)
    gZp, gsp, gcp = torch.empty_like(gZ), torch.empty_like(gs), torch.empty_like(gc)
    with _on(Z.device):
        for t0 in range(0, T, 64):
            Lc = L2[:, t0:t0 + 64].contiguous()
            Rc = R2[:, t0:t0 + 64].contiguous()
            ws = _workspace(Z.device, lib.rpgp_ski_f64_workspace_bytes(J, int(grid_size), Lc.shape[1]))
            _lib.check(lib.rpgp_ski_f64_bilinear_grad(Z.data_ptr(), gp.data_ptr(), Lc.data_ptr(), Rc.data_ptr(), gZp.data_ptr(),
                                                      gsp.data_ptr(), gcp.data_ptr(), N, J, J, J, int(grid_size), Lc.shape[1],
                                                      float(scale), ws.data_ptr(), ws.numel(), _stream()),
                       "rpgp_ski_f64_bilinear_grad")
            gZ += gZp
            gs += gsp
            gc += gcp
    return gZ, gs, gc
