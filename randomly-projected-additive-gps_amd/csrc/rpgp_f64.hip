// rpgp_f64.hip — float64 variants of the hot-path kernels for `--double` (training_routines.py:481:
// `type_ = torch.double if double else torch.float`).  Parity path, not a performance path: software exp, plain
// lane-owns-row tiling (no symmetry / rotation / slabs), every j-piece accumulates into the output in a fixed order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/rpgp.h"

namespace {

#define F64_CHECK_LAUNCH() ((int)hipGetLastError())

inline hipStream_t as_stream64(void *s) { return reinterpret_cast<hipStream_t>(s); }

const int kPieces64[] = {20, 8, 4, 2, 1};
inline int next_piece64(int remaining) {
  for (int p : kPieces64)
    if (p <= remaining) return p;
  return 1;
}

__global__ __launch_bounds__(256) void project_f64_kernel(const double *__restrict__ X, const double *__restrict__ P,
                                                          double *__restrict__ Z, long long N, int d, int J) {
  const long long total = N * J;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
    const long long n = g / J;
    const int j = (int)(g % J);
    double acc = 0.0;
    for (int k = 0; k < d; ++k) acc = fma(X[n * d + k], P[k * J + j], acc);
    Z[g] = acc;
  }
}

// dP[k][j] = sum_n X[n][k] G[n][j]: one workgroup per output element, fixed-order tree reduction
__global__ __launch_bounds__(256) void project_grad_f64_kernel(const double *__restrict__ X, const double *__restrict__ G,
                                                               double *__restrict__ dP, long long N, int d, int J) {
  __shared__ double sh[256];
  const int k = blockIdx.x / J, j = blockIdx.x % J;
  double acc = 0.0;
  for (long long n = threadIdx.x; n < N; n += 256) acc = fma(X[n * d + k], G[n * J + j], acc);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) dP[k * J + j] = sh[0];
}

// out[row][t] (+)= scale * sum_c sum_j exp(-0.5 (z1 - z2)^2) V[c][t]  (+ noise V[row][t] on the first piece)
template <int JT, int TT>
__global__ __launch_bounds__(256) void mvm_f64_kernel(const double *__restrict__ Z1, const double *__restrict__ Z2,
                                                      const double *__restrict__ V, double *__restrict__ out, int M,
                                                      int N, int ldz1, int ldz2, int T, int j0, int t0, int tcnt,
                                                      double scale, int cols_per_split) {
  // out must be initialised (noise * V or the running sum): every (row block, column split) workgroup ADDS its part
  // atomically.  The column split is what fills the chip: a row-only grid is N / 256 workgroups (29 at N = 7k).
  __shared__ double sB[64 * JT];
  __shared__ double sV[64 * TT];
  const int row = blockIdx.x * 256 + threadIdx.x;
  const bool valid = row < M;
  double a[JT], acc[TT];
#pragma unroll
  for (int j = 0; j < JT; ++j) a[j] = valid ? Z1[(size_t)row * ldz1 + j0 + j] : 0.0;
#pragma unroll
  for (int t = 0; t < TT; ++t) acc[t] = 0.0;
  const int cbeg = blockIdx.y * cols_per_split;
  const int cend = (cbeg + cols_per_split < N) ? cbeg + cols_per_split : N;
  for (int c0 = cbeg; c0 < cend; c0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * JT; e += 256) {
      const int c = e / JT, j = e % JT;
      sB[e] = (c0 + c < cend) ? Z2[(size_t)(c0 + c) * ldz2 + j0 + j] : 0.0;
    }
    for (int e = threadIdx.x; e < 64 * TT; e += 256) {
      const int c = e / TT, t = e % TT;
      sV[e] = (c0 + c < cend && t < tcnt) ? V[(size_t)(c0 + c) * T + t0 + t] : 0.0;
    }
    __syncthreads();
    const int nc = (cend - c0 < 64) ? cend - c0 : 64;
    for (int c = 0; c < nc; ++c) {
      double ks = 0.0;
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const double dd = a[j] - sB[c * JT + j];
        ks += exp(-0.5 * dd * dd);
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) acc[t] = fma(ks, sV[c * TT + t], acc[t]);
    }
  }
  if (valid) {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (t < tcnt) {
        atomicAdd(out + (size_t)row * T + t0 + t, scale * acc[t]);
      }
    }
  }
}

// out = noise * V  (noise may be 0): the base value the split MVM workgroups add to
__global__ __launch_bounds__(256) void mvm_init_f64_kernel(const double *__restrict__ V, double *__restrict__ out,
                                                           long long total, double noise) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < total) out[i] = noise != 0.0 ? noise * V[i] : 0.0;
}

template <int JT>
__global__ __launch_bounds__(256) void dense_f64_kernel(const double *__restrict__ Z1, const double *__restrict__ Z2,
                                                        double *__restrict__ out, int M, int N, int ldz1, int ldz2,
                                                        long long ldo, int j0, double scale, int accumulate) {
  constexpr int RT = 16;
  __shared__ double sA[RT][JT];
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int m0 = blockIdx.y * RT;
  for (int e = threadIdx.x; e < RT * JT; e += 256) {
    const int r = e / JT, j = e % JT;
    sA[r][j] = (m0 + r < M) ? Z1[(size_t)(m0 + r) * ldz1 + j0 + j] : 0.0;
  }
  __syncthreads();
  if (col >= N) return;
  double b[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) b[j] = Z2[(size_t)col * ldz2 + j0 + j];
  for (int r = 0; r < RT && m0 + r < M; ++r) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const double dd = sA[r][j] - b[j];
      acc += exp(-0.5 * dd * dd);
    }
    double *dst = out + (size_t)(m0 + r) * ldo + col;
    *dst = accumulate ? fma(scale, acc, *dst) : scale * acc;
  }
}

// gZ[i][j0+j] = -scale * sum_c S(i,c) e_j (z_i - z_c);  rowS[i] (+)= sum_c S(i,c) sum_j e_j
// S(i,c) = sum_t L[i,t] R[c,t] + R[i,t] L[c,t]  (DENSE = false)   or   S[c][i] read from memory (DENSE = true)
template <int JT, int TT, bool DENSE>
__global__ __launch_bounds__(256) void bilinear_f64_kernel(const double *__restrict__ Z, const double *__restrict__ L,
                                                           const double *__restrict__ Rm, const double *__restrict__ S,
                                                           double *__restrict__ gZ, double *__restrict__ rowS, int N,
                                                           int ldz, int ldg, long long lds_, int T, int j0, int t0,
                                                           int tcnt, double scale, int cols_per_split) {
  // gZ[:, j0 : j0 + JT) and rowS are zero-initialised by the host; (row block, column split) workgroups add atomically
  __shared__ double sC[64 * (JT + 2 * TT)];
  constexpr int STR = JT + 2 * TT;
  const int row = blockIdx.x * 256 + threadIdx.x;
  const bool valid = row < N;
  double a[JT], li[TT], ri[TT], accG[JT];
  double accS = 0.0;
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    a[j] = valid ? Z[(size_t)row * ldz + j0 + j] : 0.0;
    accG[j] = 0.0;
  }
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    li[t] = (!DENSE && valid && t < tcnt) ? L[(size_t)row * T + t0 + t] : 0.0;
    ri[t] = (!DENSE && valid && t < tcnt) ? Rm[(size_t)row * T + t0 + t] : 0.0;
  }
  const int cbeg = blockIdx.y * cols_per_split;
  const int cend = (cbeg + cols_per_split < N) ? cbeg + cols_per_split : N;
  for (int c0 = cbeg; c0 < cend; c0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * STR; e += 256) {
      const int c = e / STR, q = e % STR;
      const int col = c0 + c;
      double val = 0.0;
      if (col < cend) {
        if (q < JT) val = Z[(size_t)col * ldz + j0 + q];
        else if (!DENSE && q < JT + TT) { const int t = q - JT; val = t < tcnt ? L[(size_t)col * T + t0 + t] : 0.0; }
        else if (!DENSE) { const int t = q - JT - TT; val = t < tcnt ? Rm[(size_t)col * T + t0 + t] : 0.0; }
      }
      sC[e] = val;
    }
    __syncthreads();
    const int nc = (cend - c0 < 64) ? cend - c0 : 64;
    for (int c = 0; c < nc; ++c) {
      const double *p = sC + c * STR;
      double Sv = 0.0;
      if (DENSE) {
        Sv = valid ? S[(size_t)(c0 + c) * lds_ + row] : 0.0;
      } else {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          Sv = fma(li[t], p[JT + TT + t], Sv);
          Sv = fma(ri[t], p[JT + t], Sv);
        }
      }
      double ks = 0.0;
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const double dd = a[j] - p[j];
        const double e = exp(-0.5 * dd * dd);
        ks += e;
        accG[j] = fma(Sv * e, dd, accG[j]);
      }
      accS = fma(Sv, ks, accS);
    }
  }
  if (valid) {
#pragma unroll
    for (int j = 0; j < JT; ++j) atomicAdd(gZ + (size_t)row * ldg + j0 + j, -scale * accG[j]);
    atomicAdd(rowS + row, accS);
  }
}

__global__ __launch_bounds__(1024) void sum_f64_kernel(const double *__restrict__ x, double *__restrict__ out, int n,
                                                       double mul) {
  __shared__ double sh[1024];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) acc += x[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0] * mul;
}

inline int mvm64_splits(int M, int N) {
  const int nrb = (M + 255) / 256;
  int ns = (1024 + nrb - 1) / nrb;
  const int maxs = (N + 63) / 64;
  if (ns > maxs) ns = maxs;
  return ns < 1 ? 1 : ns;
}

template <int JT>
int launch_mvm64(int tt, const double *Z1, const double *Z2, const double *V, double *out, int M, int N, int ldz1,
                 int ldz2, int T, int j0, int t0, int tcnt, double scale, hipStream_t st) {
  const int ns = mvm64_splits(M, N);
  int cps = (N + ns - 1) / ns;
  cps = (cps + 63) / 64 * 64;
  dim3 grid((M + 255) / 256, (N + cps - 1) / cps), block(256);
  if (tt == 1)
    hipLaunchKernelGGL((mvm_f64_kernel<JT, 1>), grid, block, 0, st, Z1, Z2, V, out, M, N, ldz1, ldz2, T, j0, t0, tcnt, scale, cps);
  else if (tt == 4)
    hipLaunchKernelGGL((mvm_f64_kernel<JT, 4>), grid, block, 0, st, Z1, Z2, V, out, M, N, ldz1, ldz2, T, j0, t0, tcnt, scale, cps);
  else
    hipLaunchKernelGGL((mvm_f64_kernel<JT, 12>), grid, block, 0, st, Z1, Z2, V, out, M, N, ldz1, ldz2, T, j0, t0, tcnt, scale, cps);
  return F64_CHECK_LAUNCH();
}

template <int JT, bool DENSE>
int launch_bil64(int tt, const double *Z, const double *L, const double *R, const double *S, double *gZ, double *rowS,
                 int N, int ldz, int ldg, long long lds_, int T, int j0, int t0, int tcnt, double scale, hipStream_t st) {
  const int ns = mvm64_splits(N, N);
  int cps = (N + ns - 1) / ns;
  cps = (cps + 63) / 64 * 64;
  dim3 grid((N + 255) / 256, (N + cps - 1) / cps), block(256);
  if (DENSE || tt <= 4)
    hipLaunchKernelGGL((bilinear_f64_kernel<JT, 4, DENSE>), grid, block, 0, st, Z, L, R, S, gZ, rowS, N, ldz, ldg, lds_, T,
                       j0, t0, tcnt, scale, cps);
  else
    hipLaunchKernelGGL((bilinear_f64_kernel<JT, 12, DENSE>), grid, block, 0, st, Z, L, R, S, gZ, rowS, N, ldz, ldg, lds_,
                       T, j0, t0, tcnt, scale, cps);
  return F64_CHECK_LAUNCH();
}

template <int JT>
int launch_dense64(const double *Z1, const double *Z2, double *out, int M, int N, int ldz1, int ldz2, long long ldo,
                   int j0, double scale, int accumulate, hipStream_t st) {
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)((M + 15) / 16));
  hipLaunchKernelGGL((dense_f64_kernel<JT>), grid, dim3(256), 0, st, Z1, Z2, out, M, N, ldz1, ldz2, ldo, j0, scale,
                     accumulate);
  return F64_CHECK_LAUNCH();
}

#define DISPATCH_JT(jt, CALL)               \
  switch (jt) {                             \
    case 20: rc = CALL(20); break;          \
    case 8: rc = CALL(8); break;            \
    case 4: rc = CALL(4); break;            \
    case 2: rc = CALL(2); break;            \
    default: rc = CALL(1); break;           \
  }

template <bool DENSE>
int bilinear64_common(const double *Z, const double *L, const double *R, const double *S, double *gZ, double *gscale,
                      int64_t N, int ldz, int ldg, int64_t lds_, int T, int j0, int j1, double scale, double *rowS,
                      double gmul, void *stream) {
  hipStream_t st = as_stream64(stream);
  if (hipMemsetAsync(rowS, 0, (size_t)N * sizeof(double), st) != hipSuccess) return (int)hipGetLastError();
  if (hipMemset2DAsync(gZ + j0, (size_t)ldg * sizeof(double), 0, (size_t)(j1 - j0) * sizeof(double), (size_t)N, st) !=
      hipSuccess)
    return (int)hipGetLastError();
  for (int j = j0; j < j1;) {
    const int jt = next_piece64(j1 - j);
    const int nT = DENSE ? 1 : T;
    for (int t0 = 0; t0 < nT;) {
      const int tt = DENSE ? 4 : ((T - t0 > 4) ? 12 : 4);
      const int tcnt = DENSE ? 0 : ((T - t0 < tt) ? T - t0 : tt);
      int rc;
#define CALL_BIL(JTV) launch_bil64<JTV, DENSE>(tt, Z, L, R, S, gZ, rowS, (int)N, ldz, ldg, lds_, T, j, t0, tcnt, scale, st)
      DISPATCH_JT(jt, CALL_BIL)
#undef CALL_BIL
      if (rc) return rc;
      t0 += DENSE ? 1 : tcnt;
    }
    j += jt;
  }
  hipLaunchKernelGGL(sum_f64_kernel, dim3(1), dim3(1024), 0, st, rowS, gscale, (int)N, gmul);
  return F64_CHECK_LAUNCH();
}

}  // namespace

extern "C" {

int rpgp_project_f64(const double *X, const double *Peff, double *Z, int64_t N, int d, int J, void *stream) {
  if (!X || !Peff || !Z || N <= 0 || d <= 0 || J <= 0) return RPGP_EINVAL;
  const long long total = N * J;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(project_f64_kernel, dim3(blocks), dim3(256), 0, as_stream64(stream), X, Peff, Z, (long long)N, d, J);
  return F64_CHECK_LAUNCH();
}

int rpgp_project_grad_f64(const double *X, const double *G, double *dPeff, int64_t N, int d, int J, void *stream) {
  if (!X || !G || !dPeff || N <= 0 || d <= 0 || J <= 0) return RPGP_EINVAL;
  hipLaunchKernelGGL(project_grad_f64_kernel, dim3(d * J), dim3(256), 0, as_stream64(stream), X, G, dPeff, (long long)N,
                     d, J);
  return F64_CHECK_LAUNCH();
}

int rpgp_mvm_f64(const double *Z1, const double *Z2, const double *V, double *out, int64_t M, int64_t N, int ldz1,
                 int ldz2, int T, int j0, int j1, double scale, double noise, void *stream) {
  if (!Z1 || !Z2 || !V || !out || M <= 0 || N <= 0 || T <= 0 || j0 < 0 || j1 <= j0 || ldz1 < j1 || ldz2 < j1 ||
      M > 0x7fffffffLL || N > 0x7fffffffLL || (noise != 0.0 && M != N))
    return RPGP_EINVAL;
  hipStream_t st = as_stream64(stream);
  const long long total = (long long)M * T;
  hipLaunchKernelGGL(mvm_init_f64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, V, out, total, noise);
  for (int j = j0; j < j1;) {
    const int jt = next_piece64(j1 - j);
    for (int t0 = 0; t0 < T;) {
      const int tt = (T - t0 > 4) ? 12 : ((T - t0 > 1) ? 4 : 1);
      const int tcnt = (T - t0 < tt) ? T - t0 : tt;
      int rc;
#define CALL_MVM(JTV) launch_mvm64<JTV>(tt, Z1, Z2, V, out, (int)M, (int)N, ldz1, ldz2, T, j, t0, tcnt, scale, st)
      DISPATCH_JT(jt, CALL_MVM)
#undef CALL_MVM
      if (rc) return rc;
      t0 += tcnt;
    }
    j += jt;
  }
  return 0;
}

int rpgp_dense_f64(const double *Z1, const double *Z2, double *out, int64_t M, int64_t N, int ldz1, int ldz2,
                   int64_t ldo, int j0, int j1, double scale, void *stream) {
  if (!Z1 || !Z2 || !out || M <= 0 || N <= 0 || j0 < 0 || j1 <= j0 || ldz1 < j1 || ldz2 < j1 || ldo < N ||
      M > 0x7fffffffLL || N > 0x7fffffffLL)
    return RPGP_EINVAL;
  hipStream_t st = as_stream64(stream);
  int first = 1;
  for (int j = j0; j < j1;) {
    const int jt = next_piece64(j1 - j);
    int rc;
#define CALL_DENSE(JTV) launch_dense64<JTV>(Z1, Z2, out, (int)M, (int)N, ldz1, ldz2, (long long)ldo, j, scale, first ? 0 : 1, st)
    DISPATCH_JT(jt, CALL_DENSE)
#undef CALL_DENSE
    if (rc) return rc;
    first = 0;
    j += jt;
  }
  return 0;
}

int rpgp_bilinear_grad_f64(const double *Z, const double *L, const double *R, double *gZ, double *gscale, int64_t N,
                           int ldz, int ldg, int T, int j0, int j1, double scale, double *row_scratch, void *stream) {
  if (!Z || !L || !R || !gZ || !gscale || !row_scratch || N <= 0 || T <= 0 || j0 < 0 || j1 <= j0 || ldz < j1 ||
      ldg < j1 || N > 0x7fffffffLL)
    return RPGP_EINVAL;
  return bilinear64_common<false>(Z, L, R, nullptr, gZ, gscale, N, ldz, ldg, 0, T, j0, j1, scale, row_scratch, 0.5,
                                  stream);
}

int rpgp_bilinear_grad_dense_f64(const double *Z, const double *S, double *gZ, double *gscale, int64_t N, int ldz,
                                 int ldg, int64_t lds, int j0, int j1, double scale, double *row_scratch,
                                 void *stream) {
  if (!Z || !S || !gZ || !gscale || !row_scratch || N <= 0 || j0 < 0 || j1 <= j0 || ldz < j1 || ldg < j1 || lds < N ||
      N > 0x7fffffffLL)
    return RPGP_EINVAL;
  return bilinear64_common<true>(Z, nullptr, nullptr, S, gZ, gscale, N, ldz, ldg, lds, 1, j0, j1, scale, row_scratch,
                                 0.5, stream);
}

}  // extern "C"
