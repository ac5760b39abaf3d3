"""ctypes binding of the C-ABI in include/rpgp.h (librpgp.so, built from csrc/rpgp_kernels.hip).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "librpgp.so")

RPGP_EINVAL = 10001
RPGP_EWORKSPACE = 10002
RPGP_ENODEVICE = 10003
RPGP_ENUMERIC = 10004

_c_float_p = ctypes.c_void_p  # device pointers travel as integers
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f32 = ctypes.c_float
_f64 = ctypes.c_double
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/rpgp.h one to one
SIGNATURES = {
    "rpgp_version": (_int, []),
    "rpgp_error_string": (ctypes.c_char_p, [_int]),
    "rpgp_init": (_int, []),
    "rpgp_space_equally": (_int, [_vp, _int, _int, _f32, _int, _vp, _vp]),
    "rpgp_project": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp]),
    "rpgp_project_grad": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp]),
    "rpgp_mvm_sym_workspace_bytes": (_sz, [_i64, _int]),
    "rpgp_mvm_sym": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _f32, _vp, _sz, _vp]),
    "rpgp_prepare_bytes": (_sz, [_i64, _int]),
    "rpgp_prepare": (_int, [_vp, _i64, _int, _int, _vp, _sz, _vp]),
    "rpgp_prepare_status": (_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float), _vp]),
    "rpgp_mvm_sym_prepared": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _f32, _vp, _sz, _vp]),
    "rpgp_mvm_sym_range_workspace_bytes": (_sz, [_i64, _int, _int, _int]),
    "rpgp_mvm_sym_range": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _f32, _f32, _vp, _sz, _vp]),
    "rpgp_mvm_sym_prepared_range": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _f32, _f32, _vp, _sz,
                                           _vp]),
    "rpgp_mvm_rect_workspace_bytes": (_sz, [_i64, _i64, _int]),
    "rpgp_mvm_rect": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _f32, _vp, _sz, _vp]),
    "rpgp_dense": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _int, _i64, _int, _int, _f32, _vp]),
    "rpgp_bilinear_grad_workspace_bytes": (_sz, [_i64, _int]),
    "rpgp_bilinear_grad": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _f32, _vp, _sz, _vp]),
    "rpgp_bilinear_grad_dense": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _i64, _int, _int, _f32, _vp, _sz, _vp]),
    "rpgp_pivoted_cholesky": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _f32, _vp]),
    "rpgp_dense_mvm": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _f32, _vp]),
    "rpgp_symcache_bytes": (_sz, [_i64, _int, _int]),
    "rpgp_symcache_workspace_bytes": (_sz, [_i64, _int, _int, _int]),
    "rpgp_symcache_build": (_int, [_vp, _vp, _sz, _i64, _int, _int, _int, _int, _int, _int, _vp]),
    "rpgp_symcache_mvm": (_int, [_vp, _sz, _int, _vp, _vp, _i64, _int, _f32, _f32, _int, _int, _vp, _sz, _vp]),
    "rpgp_ski_workspace_bytes": (_sz, [_int, _int, _int]),
    "rpgp_ski_grid": (_int, [_vp, _i64, _int, _vp, _i64, _int, _int, _int, _vp, _vp, _sz, _vp]),
    "rpgp_ski_grid_per_projection": (_int, [_vp, _i64, _int, _vp, _i64, _int, _int, _int, _vp, _vp, _sz, _vp]),
    "rpgp_ski_mvm": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _f32, _f32, _vp, _sz, _vp]),
    "rpgp_ski_scatter": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _vp, _sz, _vp]),
    "rpgp_ski_grid_product": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    "rpgp_ski_gather": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _f32, _vp]),
    "rpgp_ski_plan_bytes": (_sz, [_i64, _int, _int]),
    "rpgp_ski_plan_workspace_bytes": (_sz, [_i64, _int, _int]),
    "rpgp_ski_plan": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _sz, _vp, _sz, _vp]),
    "rpgp_ski_mvm_planned": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _f32, _vp, _sz, _vp]),
    "rpgp_ski_scatter_planned": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp, _sz, _vp]),
    "rpgp_ski_gather_fast": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _f32, _vp]),
    "rpgp_ski_pivoted_cholesky_work_floats": (_sz, [_i64, _int]),
    "rpgp_ski_pivoted_cholesky": (_int, [_vp, _vp, _vp, _vp, _sz, _i64, _int, _int, _int, _int, _f32, _vp]),
    "rpgp_ski_dense": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _i64, _int, _int, _f32, _vp]),
    "rpgp_ski_diag": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _f32, _vp]),
    "rpgp_ski_bilinear_grad": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _f32, _vp, _sz,
                                      _vp, _vp]),
    "rpgp_ski_bilinear_scatter": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _vp, _sz, _vp]),
    "rpgp_ski_bilinear_scatter_planned": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp, _sz, _vp]),
    "rpgp_ski_bilinear_finish": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _f32,
                                        _vp, _sz, _vp, _vp]),
    "rpgp_family_mvm_workspace_bytes": (_sz, [_i64, _i64, _int, _int]),
    "rpgp_family_mvm_sym": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _f32, _f32, _vp, _sz, _vp]),
    "rpgp_family_mvm_rect": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _int, _f32, _vp, _sz, _vp]),
    "rpgp_family_dense": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _i64, _f32, _vp]),
    "rpgp_family_bilinear_grad_workspace_bytes": (_sz, [_i64, _int, _int]),
    "rpgp_family_bilinear_grad": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _f32, _vp, _sz, _vp]),
    "rpgp_family_bilinear_grad_dense": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _i64, _f32, _vp, _sz, _vp]),
    "rpgp_family_pivoted_cholesky": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _f32, _f32, _vp]),
    "rpgp_family_generic_mvm": (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _int, _f64, _f64,
                                       _vp]),
    "rpgp_family_generic_dense": (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _i64, _f64, _vp]),
    "rpgp_family_generic_bilinear_workspace_bytes": (_sz, [_int, _i64, _int]),
    "rpgp_family_generic_bilinear": (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int,
                                            _i64, _f64, _vp, _sz, _vp]),
    "rpgp_ski_bilinear_grad_comp": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _f32, _vp,
                                           _sz, _vp, _vp]),
    "rpgp_mbcg_workspace_bytes": (_sz, [_vp, _int, _int]),
    "rpgp_mbcg_solve": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _f32, _int, _vp, _vp, _f32, _vp, _vp, _vp,
                               _vp, _vp, _vp, _sz, _vp]),
    "rpgp_slq_logdet": (_int, [_vp, _vp, _int, _int, _int, _f64, _vp]),
    "rpgp_ski_f64_workspace_bytes": (_sz, [_int, _int, _int]),
    "rpgp_ski_f64_mvm": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _f64, _f64, _vp, _sz, _vp]),
    "rpgp_ski_f64_diag": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _f64, _vp]),
    "rpgp_ski_f64_dense": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _i64, _int, _int, _f64, _vp]),
    "rpgp_ski_f64_bilinear_grad": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _f64, _vp, _sz,
                                          _vp]),
    "rpgp_ski_f64_scatter": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "rpgp_ski_f64_grid_product": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "rpgp_ski_f64_gather": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _f64, _f64, _vp]),
    "rpgp_ski_f64_bilinear_finish": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _f64,
                                            _vp, _sz, _vp]),
    "rpgp_step_hyper": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _int, _int, _int, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rpgp_step_probes": (_int, [_vp, _int, _vp, _vp, _f32, _vp, _vp, _i64, _int, _vp, _vp]),
    "rpgp_step_value_workspace_bytes": (_sz, []),
    "rpgp_step_value": (_int, [_vp, _vp, _i64, _int, _int, _f64, _f64, _f64, _vp, _vp, _sz, _vp, _vp]),
    "rpgp_step_value_wait": (_int, [_int, _vp]),
    "rpgp_step_lr_workspace_bytes": (_sz, []),
    "rpgp_step_lr": (_int, [_vp, _vp, _i64, _vp, _f32, _i64, _int, _vp, _vp, _vp, _vp, _vp]),
    "rpgp_step_hyper_backward": (_int, [_vp, _vp, _int, _int, _int, _int, _f32, _vp, _vp, _vp, _int, _vp, _f32, _f32, _f32, _vp,
                                        _vp, _vp, _vp, _vp]),
    "rpgp_project_f64": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp]),
    "rpgp_project_grad_f64": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp]),
    "rpgp_mvm_f64": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _f64, _f64, _vp]),
    "rpgp_dense_f64": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _int, _i64, _int, _int, _f64, _vp]),
    "rpgp_bilinear_grad_f64": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _f64, _vp, _vp]),
    "rpgp_bilinear_grad_dense_f64": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _i64, _int, _int, _f64, _vp, _vp]),
    "rpgp_comm_create": (_int, [_int, _int, _sz, ctypes.POINTER(_vp), _vp]),
    "rpgp_comm_connect": (_int, [_vp, _vp]),
    "rpgp_comm_capacity": (_sz, [_vp]),
    "rpgp_comm_allreduce": (_int, [_vp, _vp, _sz, _int, _vp]),
    "rpgp_comm_error": (_int, [_vp, ctypes.POINTER(ctypes.c_int)]),
    "rpgp_comm_destroy": (_int, [_vp]),
    "rpgp_gram_f64_workspace_bytes": (_sz, [_int, _int]),
    "rpgp_gram_f64": (_int, [_vp, _i64, _vp, _i64, _i64, _int, _int, _vp, _vp, _sz, _vp]),
    "rpgp_woodbury_apply": (_int, [_vp, _i64, _vp, _i64, _vp, _f64, _vp, _i64, _i64, _int, _int, _vp]),
    "rpgp_woodbury_apply_cinv": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _f64, _vp, _i64, _i64, _int, _int, _vp]),
    "rpgp_woodbury_setup": (_int, [_vp, _f64, _int, _vp, _vp, _vp, _vp]),
    "rpgp_woodbury_setup_pinned": (_int, [_vp, _f64, _int, _vp, _vp, _vp, _vp, _vp]),
    "rpgp_profile_begin": (_int, []),
    "rpgp_profile_end": (_int, [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]),
    "rpgp_prepared_kernel_id": (_int, [ctypes.c_int64, _int, _int]),
    "rpgp_range_push": (_int, [ctypes.c_char_p]),
    "rpgp_range_pop": (_int, []),
    "rpgp_range_available": (_int, []),
}

class RpgpOperator(ctypes.Structure):
    """struct rpgp_operator of include/rpgp.h."""
    _fields_ = [("kind", ctypes.c_int), ("N", ctypes.c_int64), ("J", ctypes.c_int), ("ldz", ctypes.c_int),
                ("j0", ctypes.c_int), ("j1", ctypes.c_int), ("G", ctypes.c_int), ("scale", ctypes.c_float),
                ("noise", ctypes.c_float), ("Z", ctypes.c_void_p), ("prep", ctypes.c_void_p),
                ("grid_params", ctypes.c_void_p), ("Kd", ctypes.c_void_p), ("ldk", ctypes.c_int64),
                ("family", ctypes.c_void_p), ("world", ctypes.c_int), ("rank", ctypes.c_int)]


class RpgpReducer(ctypes.Structure):
    """struct rpgp_reducer of include/rpgp.h."""
    _fields_ = [("mode", ctypes.c_int), ("world", ctypes.c_int), ("rank", ctypes.c_int), ("global_N", ctypes.c_int64),
                ("fn", ctypes.c_void_p), ("ctx", ctypes.c_void_p)]


class RpgpFamily(ctypes.Structure):
    """struct rpgp_family of include/rpgp.h."""
    _fields_ = [("kind", ctypes.c_int), ("group", ctypes.c_int), ("ncomp", ctypes.c_int), ("weights", ctypes.c_void_p)]


RPGP_OP_FUSED, RPGP_OP_FUSED_PREPARED, RPGP_OP_SKI, RPGP_OP_DENSE, RPGP_OP_FAMILY, RPGP_OP_SYMCACHE = 0, 1, 2, 3, 4, 5
RPGP_OP_SUM = 6
RPGP_KIND_RBF, RPGP_KIND_MATERN15, RPGP_KIND_IMQ, RPGP_KIND_COSINE = 0, 1, 2, 3
RPGP_KIND_PRODUCT = 16
RPGP_SYMCACHE_THIN, RPGP_SYMCACHE_WIDE = 0, 1
RPGP_PIVCHOL_SCRATCH = 8192
RPGP_F32, RPGP_F64 = 0, 1
RPGP_SHARD_NONE, RPGP_SHARD_PARTIAL, RPGP_SHARD_ROWS = 0, 1, 2
RPGP_ABI_VERSION = 4
RPGP_COMM_HANDLE_BYTES = 64
# int (*rpgp_allreduce_fn)(void *ctx, void *buf, size_t count, int dtype, void *stream)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p)

_lib = None


def build(verbose=False):
    """Compile librpgp.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-j4", "-C", CSRC_DIR, "all"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building librpgp.so failed (see output above)")
    return LIB_PATH


def load():
    """Load librpgp.so and attach signatures. Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "librpgp.so not found at %s: the HIP extension is required (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C %s`). There is no CPU fallback." % (LIB_PATH, CSRC_DIR))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.rpgp_version() != RPGP_ABI_VERSION:
        raise RuntimeError("librpgp.so ABI version mismatch: %d" % lib.rpgp_version())
    _lib = lib
    return lib


def check(code, what):
    """Turn a C-ABI return code into a Python exception (ValueError for argument errors)."""
    if code == 0:
        return
    msg = load().rpgp_error_string(code).decode()
    if code == RPGP_EINVAL:
        raise ValueError("%s: %s" % (what, msg))
    raise RuntimeError("%s: %s (code %d)" % (what, msg, code))
