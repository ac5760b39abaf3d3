// rpgp_ski_f64.hip — float64 parity kernels of the grid-interpolation (SKI) operator: `--double` (torch.set_default_dtype /
// `.double()` at /root/reference/training_routines.py:481) for the `ski: true` specifications
// (/root/reference/training_routines.py:157-158, model_specs/*_ski.json).
//
// Same operator as rpgp_ski.hip / the SKI kernels of rpgp_kernels.hip —
//     K ~= scale * sum_j w_j W_j Tm_j W_j^T,   W_j: Keys cubic-convolution weights (4 taps) of projection j on its grid of G
//     points,   Tm_j[m][m'] = exp(-0.5 ((m - m') h_j)^2)
// — in float64 throughout, written for CLARITY, not speed (the float32 path is the product; this one is what its results are
// compared with, and what a `--double` run of the reference's own parity checks needs): a thread per output element, the
// scatter with hardware float64 atomics (sums are order-dependent in the last bits — 1e-16 relative, irrelevant at this
// precision), the Toeplitz stage as a plain O(G^2) sum.  The grid block is the float64 twin of the float32 one
// (rpgp_ski_common.h): [g0, h, 1/h, flags, w_0 .. w_{J-1} (flags & 1), (g0_j, h_j, 1/h_j) x J (flags & 2)].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rpgp.h"

namespace {

inline int launch_status() { return (int)hipGetLastError(); }
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__device__ __forceinline__ double cubic_w(double U) {
  return (U < 1.0) ? ((1.5 * U - 2.5) * U) * U + 1.0 : ((-0.5 * U + 2.5) * U - 4.0) * U + 2.0;
}
__device__ __forceinline__ double cubic_dw(double U) { return (U < 1.0) ? (4.5 * U - 5.0) * U : (-1.5 * U + 5.0) * U - 4.0; }
__device__ __forceinline__ int gp_flags(const double *__restrict__ gp) { return (int)gp[3]; }
// the 1-D sub-kernel of the grid's Toeplitz matrix, (flags >> 2) & 3 = RPGP_KIND_* (rpgp_ski_common.h: ski_radial_f64)
__device__ __forceinline__ double gp_radial(const double *__restrict__ gp, double d) {
  switch ((gp_flags(gp) >> 2) & 3) {
    case 1: {
      const double s = 1.7320508075688772 * fabs(d);
      return (1.0 + s) * exp(-s);
    }
    case 2: return 1.0 / sqrt(1.0 + d * d);
    case 3: return cos(3.14159265358979323846 * d);
    default: return exp(-0.5 * d * d);
  }
}
__device__ __forceinline__ double gp_wj(const double *__restrict__ gp, int j) { return (gp_flags(gp) & 1) ? gp[4 + j] : 1.0; }
__device__ __forceinline__ const double *gp_grid(const double *__restrict__ gp, int J, int j) {
  return (gp_flags(gp) & 2) ? gp + 4 + J + 3 * j : gp;
}
// first tap index, the 4 weights and (DERIV) their derivatives w.r.t. z — the rule of ski_taps (rpgp_ski_common.h)
template <bool DERIV>
__device__ __forceinline__ int taps(double z, double g0, double inv_h, int G, double (&w)[4], double (&dw)[4]) {
  double u = (z - g0) * inv_h;
  u = u < 1.0 ? 1.0 : (u > (double)(G - 2) ? (double)(G - 2) : u);
  const double fl = floor(u), fr = u - fl;
  int idx0 = (int)fl - 1;
  idx0 = idx0 < 0 ? 0 : (idx0 > G - 4 ? G - 4 : idx0);
  const double s[4] = {fr + 1.0, fr, 1.0 - fr, 2.0 - fr};
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = cubic_w(s[k]);
  if constexpr (DERIV) {
    dw[0] = cubic_dw(s[0]) * inv_h;
    dw[1] = cubic_dw(s[1]) * inv_h;
    dw[2] = -cubic_dw(s[2]) * inv_h;
    dw[3] = -cubic_dw(s[3]) * inv_h;
  }
  return idx0;
}

// hist[j][g][t] += w_q(z_ij) V[i][t]
// (hist rows are HT wide, the T columns land at hoff: the staged derivative fills [W^T L | W^T R] of a J x G x 2T block)
__global__ __launch_bounds__(256) void k_scatter(const double *__restrict__ Z, const double *__restrict__ gp,
                                                 const double *__restrict__ V, double *__restrict__ hist, long long N, int ldz,
                                                 int J, int G, int T, int HT, int hoff) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= N * T) return;
  const long long i = e / T;
  const int t = (int)(e - i * T);
  const double v = V[e];
  for (int j = 0; j < J; ++j) {
    const double *gj = gp_grid(gp, J, j);
    double w[4], dw[4];
    const int idx = taps<false>(Z[i * ldz + j], gj[0], gj[2], G, w, dw);
#pragma unroll
    for (int q = 0; q < 4; ++q) unsafeAtomicAdd(&hist[((size_t)j * G + idx + q) * HT + hoff + t], w[q] * v);
  }
}

// H[j][g][t] = (weighted ? w_j : 1) sum_g' exp(-0.5 ((g - g') h_j)^2) hist[j][g'][t]
__global__ __launch_bounds__(256) void k_grid_product(const double *__restrict__ hist, const double *__restrict__ gp,
                                                      double *__restrict__ H, int J, int G, int T, int weighted) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)J * G * T) return;
  const int t = (int)(e % T);
  const long long jg = e / T;
  const int g = (int)(jg % G), j = (int)(jg / G);
  const double h = gp_grid(gp, J, j)[1];
  double acc = 0.0;
  for (int gg = 0; gg < G; ++gg) {
    const double dd = (double)(g - gg) * h;
    acc = fma(gp_radial(gp, dd), hist[((size_t)j * G + gg) * T + t], acc);
  }
  H[e] = (weighted ? gp_wj(gp, j) : 1.0) * acc;
}

// out[i][t] = scale sum_j sum_q w_q(z_ij) H[j][idx + q][t] + noise V[i][t]
__global__ __launch_bounds__(256) void k_gather(const double *__restrict__ Z, const double *__restrict__ gp,
                                                const double *__restrict__ H, const double *__restrict__ V,
                                                double *__restrict__ out, long long M, int ldz, int J, int G, int T, double scale,
                                                double noise) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= M * T) return;
  const long long i = e / T;
  const int t = (int)(e - i * T);
  double acc = 0.0;
  for (int j = 0; j < J; ++j) {
    const double *gj = gp_grid(gp, J, j);
    double w[4], dw[4];
    const int idx = taps<false>(Z[i * ldz + j], gj[0], gj[2], G, w, dw);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = fma(w[q], H[((size_t)j * G + idx + q) * T + t], acc);
  }
  double r = scale * acc;
  if (noise != 0.0 && V) r = fma(noise, V[e], r);
  out[e] = r;
}

// K[i][i'] = scale sum_j w_j sum_{q,q'} w_q(z_ij) w_q'(z_i'j) exp(-0.5 ((idx_i + q - idx_i' - q') h_j)^2)
__global__ __launch_bounds__(256) void k_dense(const double *__restrict__ Z1, const double *__restrict__ Z2,
                                               const double *__restrict__ gp, double *__restrict__ out, long long M, long long N,
                                               int ldz1, int ldz2, long long ldo, int J, int G, double scale) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= M * N) return;
  const long long i = e / N, ip = e - i * N;
  double acc = 0.0;
  for (int j = 0; j < J; ++j) {
    const double *gj = gp_grid(gp, J, j);
    double w1[4], w2[4], dw[4];
    const int i1 = taps<false>(Z1[i * ldz1 + j], gj[0], gj[2], G, w1, dw);
    const int i2 = taps<false>(Z2[ip * ldz2 + j], gj[0], gj[2], G, w2, dw);
    double tl[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const double dd = (double)(i1 - i2 + u - 3) * gj[1];
      tl[u] = gp_radial(gp, dd);
    }
    double aj = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) aj = fma(w1[q] * w2[qq], tl[q - qq + 3], aj);
    acc = fma(gp_wj(gp, j), aj, acc);
  }
  out[i * ldo + ip] = scale * acc;
}

__global__ __launch_bounds__(256) void k_diag(const double *__restrict__ Z, const double *__restrict__ gp,
                                              double *__restrict__ out, long long N, int ldz, int J, int G, double scale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  double acc = 0.0;
  for (int j = 0; j < J; ++j) {
    const double *gj = gp_grid(gp, J, j);
    double w[4], dw[4];
    taps<false>(Z[i * ldz + j], gj[0], gj[2], G, w, dw);
    double aj = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const double dd = (double)(q - qq) * gj[1];
        aj = fma(w[q] * w[qq], gp_radial(gp, dd), aj);
      }
    acc = fma(gp_wj(gp, j), aj, acc);
  }
  out[i] = scale * acc;
}

// derivative of sum((L R^T) * K): thread per (i, j).  HL / HR: UNWEIGHTED Toeplitz products of W^T L / W^T R (J x G x T).
//   gZ[i][j] = scale w_j sum_t sum_q dw_q ( L_it HR[j][idx + q][t] + R_it HL[j][idx + q][t] )
//   gcomp[j] += w_j sum_t sum_q w_q L_it HR[j][idx + q][t]          (gscale = sum_j gcomp[j])
__global__ __launch_bounds__(256) void k_bilinear_finish(const double *__restrict__ Z, const double *__restrict__ gp,
                                                         const double *__restrict__ HL, const double *__restrict__ HR,
                                                         const double *__restrict__ L, const double *__restrict__ R,
                                                         double *__restrict__ gZ, double *__restrict__ gcomp, long long N,
                                                         int ldz, int ldg, int J, int G, int T, double scale, int ldh) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= N * J) return;
  const long long i = e / J;
  const int j = (int)(e - i * J);
  const double *gj = gp_grid(gp, J, j);
  double w[4], dw[4];
  const int idx = taps<true>(Z[i * ldz + j], gj[0], gj[2], G, w, dw);
  double gz = 0.0, gc = 0.0;
  for (int t = 0; t < T; ++t) {
    const double l = L[i * T + t], r = R[i * T + t];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double hr = HR[((size_t)j * G + idx + q) * ldh + t], hl = HL[((size_t)j * G + idx + q) * ldh + t];
      gz = fma(dw[q], l * hr + r * hl, gz);
      gc = fma(w[q], l * hr, gc);
    }
  }
  const double wj = gp_wj(gp, j);
  gZ[i * ldg + j] = scale * wj * gz;
  unsafeAtomicAdd(&gcomp[j], wj * gc);
}

__global__ void k_sum_small(const double *__restrict__ src, int n, double *__restrict__ dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int q = 0; q < n; ++q) s += src[q];
    *dst = s;
  }
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

size_t rpgp_ski_f64_workspace_bytes(int J, int G, int T) {
  if (J <= 0 || G < 8 || T <= 0) return 0;
  return 4 * align256((size_t)J * G * T * sizeof(double)) + align256((size_t)J * sizeof(double));
}

int rpgp_ski_f64_mvm(const double *Z1, const double *Z2, const double *grid_params, const double *V, double *out, int64_t M,
                     int64_t N, int ldz1, int ldz2, int J, int G, int T, double scale, double noise, void *workspace,
                     size_t workspace_bytes, void *stream) {
  if (!Z1 || !Z2 || !grid_params || !V || !out || M <= 0 || N <= 0 || J <= 0 || G < 8 || T <= 0 || ldz1 < J || ldz2 < J)
    return RPGP_EINVAL;
  if (noise != 0.0 && (Z1 != Z2 || M != N)) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_f64_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t nh = (size_t)J * G * T;
  double *hist = reinterpret_cast<double *>(workspace);
  double *H = reinterpret_cast<double *>(reinterpret_cast<char *>(workspace) + align256(nh * sizeof(double)));
  hipError_t e = hipMemsetAsync(hist, 0, nh * sizeof(double), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_scatter, dim3(blocks_for((long long)N * T)), dim3(256), 0, st, Z2, grid_params, V, hist, (long long)N, ldz2,
                     J, G, T, T, 0);
  hipLaunchKernelGGL(k_grid_product, dim3(blocks_for((long long)nh)), dim3(256), 0, st, hist, grid_params, H, J, G, T, 1);
  hipLaunchKernelGGL(k_gather, dim3(blocks_for((long long)M * T)), dim3(256), 0, st, Z1, grid_params, H, noise != 0.0 ? V : nullptr,
                     out, (long long)M, ldz1, J, G, T, scale, noise);
  return launch_status();
}

int rpgp_ski_f64_diag(const double *Z, const double *grid_params, double *diag, int64_t N, int ldz, int J, int G, double scale,
                      void *stream) {
  if (!Z || !grid_params || !diag || N <= 0 || J <= 0 || G < 8 || ldz < J) return RPGP_EINVAL;
  hipLaunchKernelGGL(k_diag, dim3(blocks_for(N)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), Z, grid_params, diag,
                     (long long)N, ldz, J, G, scale);
  return launch_status();
}

int rpgp_ski_f64_dense(const double *Z1, const double *Z2, const double *grid_params, double *out, int64_t M, int64_t N, int ldz1,
                       int ldz2, int64_t ldo, int J, int G, double scale, void *stream) {
  if (!Z1 || !Z2 || !grid_params || !out || M <= 0 || N <= 0 || J <= 0 || G < 8 || ldz1 < J || ldz2 < J || ldo < N)
    return RPGP_EINVAL;
  hipLaunchKernelGGL(k_dense, dim3(blocks_for((long long)M * N)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), Z1, Z2,
                     grid_params, out, (long long)M, (long long)N, ldz1, ldz2, (long long)ldo, J, G, scale);
  return launch_status();
}

int rpgp_ski_f64_bilinear_grad(const double *Z, const double *grid_params, const double *L, const double *R, double *gZ,
                               double *gscale, double *gcomp, int64_t N, int ldz, int ldg, int J, int G, int T, double scale,
                               void *workspace, size_t workspace_bytes, void *stream) {
  if (!Z || !grid_params || !L || !R || !gZ || !gscale || N <= 0 || J <= 0 || G < 8 || T <= 0 || ldz < J || ldg < J)
    return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_f64_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t nh = (size_t)J * G * T, nhb = align256(nh * sizeof(double));
  char *w = reinterpret_cast<char *>(workspace);
  double *histL = reinterpret_cast<double *>(w), *histR = reinterpret_cast<double *>(w + nhb);
  double *HL = reinterpret_cast<double *>(w + 2 * nhb), *HR = reinterpret_cast<double *>(w + 3 * nhb);
  double *gc = reinterpret_cast<double *>(w + 4 * nhb);
  hipError_t e = hipMemsetAsync(w, 0, 2 * nhb, st);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(gc, 0, (size_t)J * sizeof(double), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_scatter, dim3(blocks_for((long long)N * T)), dim3(256), 0, st, Z, grid_params, L, histL, (long long)N, ldz, J,
                     G, T, T, 0);
  hipLaunchKernelGGL(k_scatter, dim3(blocks_for((long long)N * T)), dim3(256), 0, st, Z, grid_params, R, histR, (long long)N, ldz, J,
                     G, T, T, 0);
  hipLaunchKernelGGL(k_grid_product, dim3(blocks_for((long long)nh)), dim3(256), 0, st, histL, grid_params, HL, J, G, T, 0);
  hipLaunchKernelGGL(k_grid_product, dim3(blocks_for((long long)nh)), dim3(256), 0, st, histR, grid_params, HR, J, G, T, 0);
  hipLaunchKernelGGL(k_bilinear_finish, dim3(blocks_for((long long)N * J)), dim3(256), 0, st, Z, grid_params, HL, HR, L, R, gZ, gc,
                     (long long)N, ldz, ldg, J, G, T, scale, T);
  hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(64), 0, st, gc, J, gscale);
  if (gcomp) {
    e = hipMemcpyAsync(gcomp, gc, (size_t)J * sizeof(double), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
  }
  return launch_status();
}

// ---- the stages as separate calls: `--double` (training_routines.py:481) for the ROW-SHARDED operator (one process per GPU, the
// ---- counterpart of MultiDeviceKernel around the grid-interpolation kernel, training_routines.py:407-408 with :157-158): every
// ---- rank scatters ITS rows, the float64 histogram is all-reduced by the caller, the grid product is replicated, every rank
// ---- gathers ITS rows.  Same contracts as rpgp_ski_scatter / rpgp_ski_grid_product / rpgp_ski_gather /
// ---- rpgp_ski_bilinear_scatter / rpgp_ski_bilinear_finish, float64 throughout.
int rpgp_ski_f64_scatter(const double *Z, const double *grid_params, const double *V, double *hist, int64_t N, int ldz, int J,
                         int G, int T, int HT, int hoff, int zero_first, void *stream) {
  if (!Z || !grid_params || !V || !hist || N <= 0 || J <= 0 || G < 8 || T <= 0 || ldz < J || HT < T || hoff < 0 || hoff + T > HT)
    return RPGP_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (zero_first) {
    const hipError_t e = hipMemsetAsync(hist, 0, (size_t)J * G * HT * sizeof(double), st);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k_scatter, dim3(blocks_for((long long)N * T)), dim3(256), 0, st, Z, grid_params, V, hist, (long long)N, ldz, J, G,
                     T, HT, hoff);
  return launch_status();
}

int rpgp_ski_f64_grid_product(const double *hist, const double *grid_params, double *H, int J, int G, int T, int weighted,
                              void *stream) {
  if (!hist || !grid_params || !H || J <= 0 || G < 8 || T <= 0) return RPGP_EINVAL;
  hipLaunchKernelGGL(k_grid_product, dim3(blocks_for((long long)J * G * T)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), hist,
                     grid_params, H, J, G, T, weighted ? 1 : 0);
  return launch_status();
}

int rpgp_ski_f64_gather(const double *Z, const double *grid_params, const double *H, const double *V, double *out, int64_t M,
                        int ldz, int J, int G, int T, double scale, double noise, void *stream) {
  if (!Z || !grid_params || !H || !out || M <= 0 || J <= 0 || G < 8 || T <= 0 || ldz < J) return RPGP_EINVAL;
  if (noise != 0.0 && !V) return RPGP_EINVAL;
  hipLaunchKernelGGL(k_gather, dim3(blocks_for((long long)M * T)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), Z, grid_params,
                     H, noise != 0.0 ? V : nullptr, out, (long long)M, ldz, J, G, T, scale, noise);
  return launch_status();
}

// hist2: J x G x 2T float64 = [W^T L | W^T R] summed over ALL rows (all-reduced by the caller); workspace: J * G * 2T + J doubles
int rpgp_ski_f64_bilinear_finish(const double *Z, const double *grid_params, const double *hist2, const double *L, const double *R,
                                 double *gZ, double *gscale, double *gcomp, int64_t N, int ldz, int ldg, int J, int G, int T,
                                 double scale, void *workspace, size_t workspace_bytes, void *stream) {
  if (!Z || !grid_params || !hist2 || !L || !R || !gZ || !gscale || N <= 0 || J <= 0 || G < 8 || T <= 0 || ldz < J || ldg < J)
    return RPGP_EINVAL;
  const size_t nh2 = (size_t)J * G * 2 * T;
  if (!workspace || workspace_bytes < align256(nh2 * sizeof(double)) + align256((size_t)J * sizeof(double))) return RPGP_EWORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double *H2 = reinterpret_cast<double *>(workspace);
  double *gc = reinterpret_cast<double *>(reinterpret_cast<char *>(workspace) + align256(nh2 * sizeof(double)));
  const hipError_t e = hipMemsetAsync(gc, 0, (size_t)J * sizeof(double), st);
  if (e != hipSuccess) return (int)e;
  // UNWEIGHTED Toeplitz products of both halves at once (2T columns), then the per-row finish reading HL / HR out of H2
  hipLaunchKernelGGL(k_grid_product, dim3(blocks_for((long long)nh2)), dim3(256), 0, st, hist2, grid_params, H2, J, G, 2 * T, 0);
  hipLaunchKernelGGL(k_bilinear_finish, dim3(blocks_for((long long)N * J)), dim3(256), 0, st, Z, grid_params, H2, H2 + T, L, R, gZ, gc,
                     (long long)N, ldz, ldg, J, G, T, scale, 2 * T);
  hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(64), 0, st, gc, J, gscale);
  if (gcomp) {
    const hipError_t e2 = hipMemcpyAsync(gcomp, gc, (size_t)J * sizeof(double), hipMemcpyDeviceToDevice, st);
    if (e2 != hipSuccess) return (int)e2;
  }
  return launch_status();
}

}  // extern "C"
