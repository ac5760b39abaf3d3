// rpgp_family_generic.hip — the generalised additive family with RUNTIME (kind, group) in float32 and float64:
//     K[i,i'] = scale * sum_c w[c] * phi_kind( r_c ),   r_c^2 = sum_{m < group} (Z[i][c group + m] - Z'[i'][c group + m])^2
// or, with RPGP_KIND_PRODUCT set in `kind`, the PRODUCT of 1-D sub-kernels over the group (the ProductKernel groups of
// polynomial_projection_kernels.py:70-86; for the RBF the two forms are the same function):
//     K[i,i'] = scale * sum_c w[c] * prod_{m < group} phi_kind( (Z[i][c group + m] - Z'[i'][c group + m])^2 )
// What it serves (everything the templated fast kernels of rpgp_kernels.hip do not instantiate):
//   * `--double` (training_routines.py:481) for every family member — Matern / InverseMQ / Cosine sub-kernels, k > 1 RBF
//     sub-kernels, the per-component weights of the rp_poly / strictly_additive / additive kinds;
//   * k > 1 sub-kernels of the NON-RBF types as the reference builds them for `additive_rp` (training_routines.py:172-174:
//     `kernel(active_dims=range(i, i + k))`, i.e. the RADIAL k-dimensional Matern / InverseMQ (imq_kernel.py:8-9) / Cosine);
//   * any group size up to 32 without padding.
// Parity path, not a performance path: a lane owns a row and walks all columns (64 at a time through LDS), library exp /
// sqrt / cos, no symmetry, no slabs, no atomics (every output has one writer: bitwise reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/rpgp.h"

namespace {

constexpr int kRows = 128;       // rows per workgroup (one per thread)
constexpr int kTile = 64;        // columns staged per step
constexpr int kMaxCols = 64;     // columns of Z (ncomp * group)
constexpr int kMaxGroup = 32;
constexpr int kMaxT = 16;

template <typename F> struct Fn;
template <> struct Fn<float> {
  static __device__ __forceinline__ float exp_(float x) { return expf(x); }
  static __device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
  static __device__ __forceinline__ float cos_(float x) { return cosf(x); }
  static __device__ __forceinline__ float sin_(float x) { return sinf(x); }
  static __device__ __forceinline__ float rsqrt_(float x) { return 1.0f / sqrtf(x); }
};
template <> struct Fn<double> {
  static __device__ __forceinline__ double exp_(double x) { return exp(x); }
  static __device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
  static __device__ __forceinline__ double cos_(double x) { return cos(x); }
  static __device__ __forceinline__ double sin_(double x) { return sin(x); }
  static __device__ __forceinline__ double rsqrt_(double x) { return 1.0 / sqrt(x); }
};

// phi(r^2) and d phi / d(r^2)  (so that d phi / d z_m = 2 dphi2 (z_m - z'_m))
template <typename F>
__device__ __forceinline__ void phi_eval(int kind, F r2, F &phi, F &dphi2) {
  const F s3 = (F)1.7320508075688772935, pi = (F)3.14159265358979323846;
  if (kind == RPGP_KIND_RBF) {
    phi = Fn<F>::exp_((F)-0.5 * r2);
    dphi2 = (F)-0.5 * phi;
  } else if (kind == RPGP_KIND_MATERN15) {
    const F r = Fn<F>::sqrt_(r2);
    const F e = Fn<F>::exp_(-s3 * r);
    phi = ((F)1 + s3 * r) * e;
    dphi2 = (F)-1.5 * e;
  } else if (kind == RPGP_KIND_IMQ) {
    const F q = Fn<F>::rsqrt_((F)1 + r2);
    phi = q;
    dphi2 = (F)-0.5 * q * q * q;
  } else {                                              // cos(pi r);  d/d(r^2) = -pi sin(pi r) / (2 r)  -> -pi^2 / 2 at 0
    const F r = Fn<F>::sqrt_(r2);
    phi = Fn<F>::cos_(pi * r);
    dphi2 = r > (F)1e-6 ? -pi * Fn<F>::sin_(pi * r) / ((F)2 * r) : (F)-0.5 * pi * pi;
  }
}

template <typename F>
__device__ __forceinline__ F pair_value(int kind, int group, int ncomp, const F *__restrict__ w, const F *__restrict__ a,
                                        const F *__restrict__ b) {
  const bool product = (kind & RPGP_KIND_PRODUCT) != 0;
  kind &= 3;
  F acc = 0;
  for (int c = 0; c < ncomp; ++c) {
    F phi, dp;
    if (product) {
      F pr = 1;
      for (int m = 0; m < group; ++m) {
        const F d = a[c * group + m] - b[c * group + m];
        phi_eval<F>(kind, d * d, phi, dp);
        pr *= phi;
      }
      phi = pr;
    } else {
      F r2 = 0;
      for (int m = 0; m < group; ++m) {
        const F d = a[c * group + m] - b[c * group + m];
        r2 += d * d;
      }
      phi_eval<F>(kind, r2, phi, dp);
    }
    acc += w[c] * phi;
  }
  return acc;
}

// LDS: [kRows][ncols] row coordinates | [kTile][ncols] column coordinates | [kTile][T] right-hand sides | [ncomp] weights
template <typename F>
__global__ __launch_bounds__(kRows) void famg_mvm_kernel(int kind, int group, int ncomp, const F *__restrict__ weights,
                                                         const F *__restrict__ Z1, const F *__restrict__ Z2,
                                                         const F *__restrict__ V, F *__restrict__ out, long long M,
                                                         long long N, int ldz1, int ldz2, int T, F scale, F noise, int sym) {
  extern __shared__ unsigned char smem_raw[];
  const int ncols = ncomp * group;
  F *sRow = reinterpret_cast<F *>(smem_raw);
  F *sCol = sRow + kRows * ncols;
  F *sV = sCol + kTile * ncols;
  F *sW = sV + kTile * T;
  const long long row = (long long)blockIdx.x * kRows + threadIdx.x;
  const bool valid = row < M;
  for (int e = threadIdx.x; e < ncomp; e += kRows) sW[e] = weights[e];
  for (int j = 0; j < ncols; ++j) sRow[threadIdx.x * ncols + j] = valid ? Z1[row * ldz1 + j] : (F)0;
  F acc[kMaxT];
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) acc[t] = 0;
  for (long long c0 = 0; c0 < N; c0 += kTile) {
    __syncthreads();
    for (int e = threadIdx.x; e < kTile * ncols; e += kRows) {
      const int c = e / ncols, j = e - c * ncols;
      sCol[e] = (c0 + c < N) ? Z2[(c0 + c) * ldz2 + j] : (F)0;
    }
    for (int e = threadIdx.x; e < kTile * T; e += kRows) {
      const int c = e / T, t = e - c * T;
      sV[e] = (c0 + c < N) ? V[(c0 + c) * T + t] : (F)0;
    }
    __syncthreads();
    const int nc = (N - c0 < kTile) ? (int)(N - c0) : kTile;
    for (int c = 0; c < nc; ++c) {
      const F k = pair_value<F>(kind, group, ncomp, sW, sRow + threadIdx.x * ncols, sCol + c * ncols);
#pragma unroll
      for (int t = 0; t < kMaxT; ++t)
        if (t < T) acc[t] += k * sV[c * T + t];
    }
  }
  if (valid) {
#pragma unroll
    for (int t = 0; t < kMaxT; ++t)
      if (t < T) out[row * T + t] = scale * acc[t] + ((sym && noise != (F)0) ? noise * V[row * T + t] : (F)0);
  }
}

template <typename F>
__global__ __launch_bounds__(kRows) void famg_dense_kernel(int kind, int group, int ncomp, const F *__restrict__ weights,
                                                           const F *__restrict__ Z1, const F *__restrict__ Z2,
                                                           F *__restrict__ out, long long M, long long N, int ldz1,
                                                           int ldz2, long long ldo, F scale) {
  extern __shared__ unsigned char smem_raw[];
  const int ncols = ncomp * group;
  F *sRow = reinterpret_cast<F *>(smem_raw);
  F *sCol = sRow + kRows * ncols;
  F *sW = sCol + kTile * ncols;
  // thread = column of the tile pair (coalesced stores): workgroup (x: 128-column block, y: 64-row block)
  const long long col = (long long)blockIdx.x * kRows + threadIdx.x;
  const long long r0 = (long long)blockIdx.y * kTile;
  for (int e = threadIdx.x; e < ncomp; e += kRows) sW[e] = weights[e];
  for (int j = 0; j < ncols; ++j) sRow[threadIdx.x * ncols + j] = col < N ? Z2[col * ldz2 + j] : (F)0;
  for (int e = threadIdx.x; e < kTile * ncols; e += kRows) {
    const int r = e / ncols, j = e - r * ncols;
    sCol[e] = (r0 + r < M) ? Z1[(r0 + r) * ldz1 + j] : (F)0;
  }
  __syncthreads();
  if (col >= N) return;
  for (int r = 0; r < kTile && r0 + r < M; ++r)
    out[(r0 + r) * ldo + col] = scale * pair_value<F>(kind, group, ncomp, sW, sCol + r * ncols, sRow + threadIdx.x * ncols);
}

// One component per workgroup row (grid.y = component): lane owns row i and accumulates, over all columns c,
//   gZ[i][comp cols] += scale w S(i,c) 2 dphi2 (z_i - z_c),    rowC[comp][i] = 0.5 sum_c S(i,c) phi
// with S(i,c) = sum_t L[i,t] R[c,t] + R[i,t] L[c,t]  or  S[c][i] from memory (symmetric).
template <typename F, bool DENSE>
__global__ __launch_bounds__(kRows) void famg_bilinear_kernel(int kind, int group, int ncomp, const F *__restrict__ weights,
                                                              const F *__restrict__ Z, const F *__restrict__ L,
                                                              const F *__restrict__ Rm, const F *__restrict__ S,
                                                              F *__restrict__ gZ, F *__restrict__ rowC, long long N, int ldz,
                                                              int ldg, int T, long long lds_, F scale) {
  extern __shared__ unsigned char smem_raw[];
  F *sCol = reinterpret_cast<F *>(smem_raw);            // [kTile][group]
  F *sLR = sCol + kTile * group;                        // [kTile][2 T]
  const int comp = blockIdx.y;
  const long long row = (long long)blockIdx.x * kRows + threadIdx.x;
  const bool valid = row < N;
  F a[kMaxGroup], g[kMaxGroup], li[kMaxT], ri[kMaxT];
#pragma unroll
  for (int m = 0; m < kMaxGroup; ++m) {
    a[m] = (valid && m < group) ? Z[row * ldz + comp * group + m] : (F)0;
    g[m] = 0;
  }
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) {
    li[t] = (!DENSE && valid && t < T) ? L[row * T + t] : (F)0;
    ri[t] = (!DENSE && valid && t < T) ? Rm[row * T + t] : (F)0;
  }
  const F w = weights[comp];
  const bool product = (kind & RPGP_KIND_PRODUCT) != 0 && group > 1;
  kind &= 3;
  F accC = 0;
  for (long long c0 = 0; c0 < N; c0 += kTile) {
    __syncthreads();
    for (int e = threadIdx.x; e < kTile * group; e += kRows) {
      const int c = e / group, m = e - c * group;
      sCol[e] = (c0 + c < N) ? Z[(c0 + c) * ldz + comp * group + m] : (F)0;
    }
    if (!DENSE) {
      for (int e = threadIdx.x; e < kTile * 2 * T; e += kRows) {
        const int c = e / (2 * T), q = e - c * 2 * T;
        F v = 0;
        if (c0 + c < N) v = q < T ? L[(c0 + c) * T + q] : Rm[(c0 + c) * T + q - T];
        sLR[e] = v;
      }
    }
    __syncthreads();
    const int nc = (N - c0 < kTile) ? (int)(N - c0) : kTile;
    for (int c = 0; c < nc; ++c) {
      F Sv = 0;
      if (DENSE) {
        Sv = valid ? S[(c0 + c) * lds_ + row] : (F)0;
      } else {
#pragma unroll
        for (int t = 0; t < kMaxT; ++t)
          if (t < T) Sv += li[t] * sLR[c * 2 * T + T + t] + ri[t] * sLR[c * 2 * T + t];
      }
      F r2 = 0, dd[kMaxGroup];
#pragma unroll
      for (int m = 0; m < kMaxGroup; ++m) {
        dd[m] = m < group ? a[m] - sCol[c * group + m] : (F)0;
        r2 += dd[m] * dd[m];
      }
      F phi, dp;
      if (product) {
        // d/dz_m of prod_m phi(d_m^2) = 2 dphi2(d_m^2) d_m * (the product of the OTHER factors): prefix times suffix
        // products (no division: a factor may be zero — the cosine has roots)
        F ph[kMaxGroup], dq[kMaxGroup];
        F pre = 1;
#pragma unroll
        for (int m = 0; m < kMaxGroup; ++m) {
          ph[m] = 1;
          dq[m] = 0;
          if (m < group) {
            phi_eval<F>(kind, dd[m] * dd[m], ph[m], dp);
            dq[m] = pre * (F)2 * dp * dd[m];        // prefix product * own derivative
            pre *= ph[m];
          }
        }
        accC += Sv * pre;
        F suf = 1;
#pragma unroll
        for (int m = kMaxGroup - 1; m >= 0; --m) {
          if (m < group) {
            g[m] += Sv * dq[m] * suf;
            suf *= ph[m];
          }
        }
      } else {
        phi_eval<F>(kind, r2, phi, dp);
        accC += Sv * phi;
        const F f = Sv * (F)2 * dp;
#pragma unroll
        for (int m = 0; m < kMaxGroup; ++m) g[m] += f * dd[m];
      }
    }
  }
  if (valid) {
#pragma unroll
    for (int m = 0; m < kMaxGroup; ++m)
      if (m < group) gZ[row * ldg + comp * group + m] = scale * w * g[m];
    rowC[(long long)comp * N + row] = (F)0.5 * accC;
  }
}

// gcomp[c] = sum_i rowC[c][i]  (fixed-order tree per component)
template <typename F>
__global__ __launch_bounds__(256) void famg_sum_rows_kernel(const F *__restrict__ rowC, F *__restrict__ gcomp, long long N) {
  __shared__ F sh[256];
  const int comp = blockIdx.x;
  F acc = 0;
  for (long long i = threadIdx.x; i < N; i += 256) acc += rowC[(long long)comp * N + i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) gcomp[comp] = sh[0];
}

template <typename K>
int ensure_lds(K kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return 0;
  if (bytes > 150 * 1024) return RPGP_EINVAL;
  return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

bool bad_family(int kind, int group, int ncomp) {
  return kind < 0 || (kind & ~(3 | RPGP_KIND_PRODUCT)) != 0 || group < 1 || group > kMaxGroup || ncomp < 1 ||
         (long long)group * ncomp > kMaxCols;
}

template <typename F>
int mvm_impl(int kind, int group, int ncomp, const void *w, const void *Z1, const void *Z2, const void *V, void *out, int64_t M,
             int64_t N, int ldz1, int ldz2, int T, double scale, double noise, int sym, hipStream_t st) {
  const int ncols = group * ncomp;
  const size_t lds = ((size_t)(kRows + kTile) * ncols + (size_t)kTile * T + ncomp) * sizeof(F);
  int rc = ensure_lds(famg_mvm_kernel<F>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL((famg_mvm_kernel<F>), dim3((unsigned)((M + kRows - 1) / kRows)), dim3(kRows), lds, st, kind, group, ncomp,
                     reinterpret_cast<const F *>(w), reinterpret_cast<const F *>(Z1), reinterpret_cast<const F *>(Z2),
                     reinterpret_cast<const F *>(V), reinterpret_cast<F *>(out), (long long)M, (long long)N, ldz1, ldz2, T,
                     (F)scale, (F)noise, sym);
  return (int)hipGetLastError();
}

template <typename F>
int dense_impl(int kind, int group, int ncomp, const void *w, const void *Z1, const void *Z2, void *out, int64_t M, int64_t N,
               int ldz1, int ldz2, int64_t ldo, double scale, hipStream_t st) {
  const int ncols = group * ncomp;
  const size_t lds = ((size_t)(kRows + kTile) * ncols + ncomp) * sizeof(F);
  int rc = ensure_lds(famg_dense_kernel<F>, lds);
  if (rc) return rc;
  dim3 grid((unsigned)((N + kRows - 1) / kRows), (unsigned)((M + kTile - 1) / kTile));
  hipLaunchKernelGGL((famg_dense_kernel<F>), grid, dim3(kRows), lds, st, kind, group, ncomp, reinterpret_cast<const F *>(w),
                     reinterpret_cast<const F *>(Z1), reinterpret_cast<const F *>(Z2), reinterpret_cast<F *>(out), (long long)M,
                     (long long)N, ldz1, ldz2, (long long)ldo, (F)scale);
  return (int)hipGetLastError();
}

template <typename F>
int bilinear_impl(int kind, int group, int ncomp, const void *w, const void *Z, const void *L, const void *R, const void *S,
                  void *gZ, void *gcomp, int64_t N, int ldz, int ldg, int T, int64_t lds_, double scale, void *ws, hipStream_t st) {
  const size_t lds = ((size_t)kTile * group + (size_t)kTile * 2 * (S ? 0 : T)) * sizeof(F);
  dim3 grid((unsigned)((N + kRows - 1) / kRows), (unsigned)ncomp);
  F *rowC = reinterpret_cast<F *>(ws);
  if (S)
    hipLaunchKernelGGL((famg_bilinear_kernel<F, true>), grid, dim3(kRows), lds, st, kind, group, ncomp,
                       reinterpret_cast<const F *>(w), reinterpret_cast<const F *>(Z), (const F *)nullptr, (const F *)nullptr,
                       reinterpret_cast<const F *>(S), reinterpret_cast<F *>(gZ), rowC, (long long)N, ldz, ldg, 0,
                       (long long)lds_, (F)scale);
  else
    hipLaunchKernelGGL((famg_bilinear_kernel<F, false>), grid, dim3(kRows), lds, st, kind, group, ncomp,
                       reinterpret_cast<const F *>(w), reinterpret_cast<const F *>(Z), reinterpret_cast<const F *>(L),
                       reinterpret_cast<const F *>(R), (const F *)nullptr, reinterpret_cast<F *>(gZ), rowC, (long long)N, ldz,
                       ldg, T, 0LL, (F)scale);
  hipLaunchKernelGGL((famg_sum_rows_kernel<F>), dim3((unsigned)ncomp), dim3(256), 0, st, rowC, reinterpret_cast<F *>(gcomp),
                     (long long)N);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int rpgp_family_generic_mvm(int dtype, int kind, int group, int ncomp, const void *weights, const void *Z1, const void *Z2,
                            const void *V, void *out, int64_t M, int64_t N, int ldz1, int ldz2, int T, double scale,
                            double noise, void *stream) {
  if (bad_family(kind, group, ncomp) || !weights || !Z1 || !V || !out || M <= 0 || N <= 0 || T <= 0 || T > kMaxT ||
      ldz1 < group * ncomp || (Z2 && ldz2 < group * ncomp) || (dtype != RPGP_F32 && dtype != RPGP_F64))
    return RPGP_EINVAL;
  const int sym = Z2 == nullptr;
  if (sym && M != N) return RPGP_EINVAL;
  if (!sym && noise != 0.0) return RPGP_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == RPGP_F64)
    return mvm_impl<double>(kind, group, ncomp, weights, Z1, sym ? Z1 : Z2, V, out, M, N, ldz1, sym ? ldz1 : ldz2, T, scale, noise,
                            sym, st);
  return mvm_impl<float>(kind, group, ncomp, weights, Z1, sym ? Z1 : Z2, V, out, M, N, ldz1, sym ? ldz1 : ldz2, T, scale, noise, sym,
                         st);
}

int rpgp_family_generic_dense(int dtype, int kind, int group, int ncomp, const void *weights, const void *Z1, const void *Z2,
                              void *out, int64_t M, int64_t N, int ldz1, int ldz2, int64_t ldo, double scale, void *stream) {
  if (bad_family(kind, group, ncomp) || !weights || !Z1 || !Z2 || !out || M <= 0 || N <= 0 || ldz1 < group * ncomp ||
      ldz2 < group * ncomp || ldo < N || (dtype != RPGP_F32 && dtype != RPGP_F64))
    return RPGP_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == RPGP_F64) return dense_impl<double>(kind, group, ncomp, weights, Z1, Z2, out, M, N, ldz1, ldz2, ldo, scale, st);
  return dense_impl<float>(kind, group, ncomp, weights, Z1, Z2, out, M, N, ldz1, ldz2, ldo, scale, st);
}

size_t rpgp_family_generic_bilinear_workspace_bytes(int dtype, int64_t N, int ncomp) {
  if (N <= 0 || ncomp <= 0) return 0;
  return (size_t)N * ncomp * (dtype == RPGP_F64 ? 8 : 4);
}

int rpgp_family_generic_bilinear(int dtype, int kind, int group, int ncomp, const void *weights, const void *Z, const void *L,
                                 const void *R, const void *S, void *gZ, void *gcomp, int64_t N, int ldz, int ldg, int T,
                                 int64_t lds, double scale, void *workspace, size_t workspace_bytes, void *stream) {
  if (bad_family(kind, group, ncomp) || !weights || !Z || !gZ || !gcomp || N <= 0 || ldz < group * ncomp ||
      ldg < group * ncomp || (dtype != RPGP_F32 && dtype != RPGP_F64))
    return RPGP_EINVAL;
  if (!S && (!L || !R || T <= 0 || T > kMaxT)) return RPGP_EINVAL;
  if (S && lds < N) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_family_generic_bilinear_workspace_bytes(dtype, N, ncomp)) return RPGP_EWORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == RPGP_F64)
    return bilinear_impl<double>(kind, group, ncomp, weights, Z, L, R, S, gZ, gcomp, N, ldz, ldg, T, lds, scale, workspace, st);
  return bilinear_impl<float>(kind, group, ncomp, weights, Z, L, R, S, gZ, gcomp, N, ldz, ldg, T, lds, scale, workspace, st);
}

}  // extern "C"
