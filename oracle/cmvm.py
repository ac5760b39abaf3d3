"""ORACLE — test infrastructure only (never imported by the product package).

ctypes front of oracle/cmvm.c: the float64 additive-kernel matrix and its products, in C with OpenMP, for the parity tests
at the BASELINE sizes (the full N = 50 000 product is 5e10 exponentials).  `build()` compiles the C file with gcc into
oracle/_build/ (git-ignored, travels to the GPU box like the HIP library; rebuilt there on demand if missing or stale).
Same formulas as oracle/dense_gp.py — tests/test_oracle_pinned.py checks them against each other and the reference-generated
golden vectors (memory_efficient_gam_kernel.py:20-30)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "cmvm.c")
_OUT_DIR = os.path.join(_HERE, "_build")
_SO = os.path.join(_OUT_DIR, "liboracle_cmvm.so")
_lib = None


def build(force=False):
    """gcc -O3 -fopenmp oracle/cmvm.c -> oracle/_build/liboracle_cmvm.so (no -ffast-math, no -march: portable across hosts)."""
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(_SRC):
        return _SO
    os.makedirs(_OUT_DIR, exist_ok=True)
    tmp = _SO + ".%d.tmp" % os.getpid()
    subprocess.run(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-std=c99", _SRC, "-o", tmp, "-lm"], check=True,
                   capture_output=True)
    os.replace(tmp, _SO)
    return _SO


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        lib.oracle_kernel_f64.argtypes = [dp, dp, dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_double]
        lib.oracle_kernel_f64.restype = None
        lib.oracle_mvm_f64.argtypes = [dp, dp, dp, dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_double, ctypes.c_int]
        lib.oracle_mvm_f64.restype = None
        lib.oracle_bilinear_gz_f64.argtypes = [dp, dp, dp, ctypes.c_long, ctypes.c_int, ctypes.c_double]
        lib.oracle_bilinear_gz_f64.restype = None
        lib.oracle_cmvm_version.restype = ctypes.c_int
        if lib.oracle_cmvm_version() != 2:          # a stale build of an older source: rebuild once
            del lib
            lib = ctypes.CDLL(build(force=True))
            lib.oracle_kernel_f64.argtypes = [dp, dp, dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_double]
            lib.oracle_kernel_f64.restype = None
            lib.oracle_mvm_f64.argtypes = [dp, dp, dp, dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_int]
            lib.oracle_mvm_f64.restype = None
            lib.oracle_bilinear_gz_f64.argtypes = [dp, dp, dp, ctypes.c_long, ctypes.c_int, ctypes.c_double]
            lib.oracle_bilinear_gz_f64.restype = None
            lib.oracle_cmvm_version.restype = ctypes.c_int
            assert lib.oracle_cmvm_version() == 2
        _lib = lib
    return _lib


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def kernel(Z1, Z2, scale=1.0):
    """scale * K_add(Z1, Z2) as a dense float64 matrix (dense_gp.additive_rbf with weight = scale)."""
    Z1, p1 = _c(Z1)
    Z2, p2 = _c(Z2)
    assert Z1.ndim == 2 and Z2.ndim == 2 and Z1.shape[1] == Z2.shape[1]
    K = np.empty((Z1.shape[0], Z2.shape[0]))
    _load().oracle_kernel_f64(p1, p2, K.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), Z1.shape[0], Z2.shape[0],
                              Z1.shape[1], float(scale))
    return K


def mvm(Z1, Z2, V, scale, noise=0.0):
    """scale * K_add(Z1, Z2) @ V (+ noise * V: then Z1 and Z2 must be the same points) — dense_gp.mvm without the matrix."""
    Z1, p1 = _c(Z1)
    Z2, p2 = _c(Z2)
    V = np.asarray(V, dtype=np.float64)
    vec = V.ndim == 1
    V, pv = _c(V.reshape(V.shape[0], -1))
    assert Z1.shape[1] == Z2.shape[1] and V.shape[0] == Z2.shape[0]
    if noise:
        assert Z1.shape == Z2.shape
    out = np.empty((Z1.shape[0], V.shape[1]))
    _load().oracle_mvm_f64(p1, p2, pv, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), Z1.shape[0], Z2.shape[0],
                           Z1.shape[1], V.shape[1], float(scale), float(noise), 1 if noise else 0)
    return out[:, 0] if vec else out


def bilinear_gz(Z, S, scale):
    """d/dZ of sum(W * scale K_add(Z, Z)) given S = W + W^T (N x N float64): gZ[i][j] = -scale sum_i' S[i,i'] (z_ij - z_i'j)
    exp(-0.5 (z_ij - z_i'j)^2)  (SURVEY.md Appendix A.2)."""
    Z, pz = _c(Z)
    S, ps = _c(S)
    assert S.shape == (Z.shape[0], Z.shape[0])
    out = np.empty_like(Z)
    _load().oracle_bilinear_gz_f64(pz, ps, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), Z.shape[0], Z.shape[1], float(scale))
    return out
