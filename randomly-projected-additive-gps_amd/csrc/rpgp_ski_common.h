// rpgp_ski_common.h — device helpers of the SKI path shared by rpgp_kernels.hip and rpgp_ski.hip (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

namespace {

__device__ __forceinline__ float cubic_w(float U) {      // Keys cubic convolution kernel, U = |distance| / h in [0, 2]
  return (U < 1.0f) ? ((1.5f * U - 2.5f) * U) * U + 1.0f : ((-0.5f * U + 2.5f) * U - 4.0f) * U + 2.0f;
}
__device__ __forceinline__ float cubic_dw(float U) {     // d/dU of the above
  return (U < 1.0f) ? (4.5f * U - 5.0f) * U : (-1.5f * U + 5.0f) * U - 4.0f;
}

// Grid parameter block (device floats):
//   shared grid (flags & 2 == 0):  [g0, h, 1/h, flags, (w_0 .. w_{J-1} if flags & 1)]
//   per-projection grids (flags & 2):  [., ., ., flags, w_0 .. w_{J-1} (ones when flags & 1 == 0), (g0_j, h_j, 1/h_j) x J]
// flags & 1: per-projection output scales (the `weighted` components of polynomial_projection_kernels.py:88-98 under SKI),
// K = scale * sum_j w_j W_j Tm_j W_j^T — the weight rides on the Toeplitz stage.  flags & 2: the reference's grid rule
// (polynomial_projection_kernels.py:54-63: every projection has its own bounds, spacing (max - min) / (G - 4), +- 2.01
// spacings of margin), so interpolation cells, and the Toeplitz first column, differ per projection.
// flags & 12: the 1-D sub-kernel the grid's Toeplitz matrix is built from, (flags >> 2) = RPGP_KIND_* of include/rpgp.h
// (training_routines.py:157-158 wraps WHATEVER `_map_to_kernel` returned, :47-88, in GridInterpolationKernel): 0 RBF
// exp(-d^2 / 2), 1 Matern-1.5 (1 + sqrt3 |d|) exp(-sqrt3 |d|), 2 InverseMQ (1 + d^2)^(-1/2) (imq_kernel.py:8-9), 3 Cosine
// cos(pi d).  Scatter, gather and the derivative's staging never look at it: only the grid-to-grid entries do.
__device__ __forceinline__ int ski_flags(const float *__restrict__ gp) { return (int)gp[3]; }
__device__ __forceinline__ int ski_kind(const float *__restrict__ gp) { return (ski_flags(gp) >> 2) & 3; }
// the sub-kernel between two grid points `lag` spacings h apart — float64 (Toeplitz first columns) ...
__device__ __forceinline__ double ski_radial_f64(int kind, double d) {
  switch (kind) {
    case 1: {
      const double s = 1.7320508075688772 * fabs(d);
      return (1.0 + s) * exp(-s);
    }
    case 2: return 1.0 / sqrt(1.0 + d * d);
    case 3: return cos(3.14159265358979323846 * d);
    default: return exp(-0.5 * d * d);
  }
}
// ... and float32 (dense blocks, diagonal, wide Toeplitz): the RBF branch is the arithmetic these kernels always used
// (exp2 of the pre-scaled square: same bits as before the other kinds existed)
__device__ __forceinline__ float ski_radial_f32(int kind, int lag, float h) {
  if (kind == 0) {
    const float d = (float)lag * (h * 0.8493218002880191f);
    return __builtin_amdgcn_exp2f(-(d * d));
  }
  const float d = (float)lag * h;
  if (kind == 1) {
    const float s = 1.7320508075688772f * d;
    return (1.0f + s) * __builtin_amdgcn_exp2f(-1.4426950408889634f * s);
  }
  if (kind == 2) return __builtin_amdgcn_rsqf(__builtin_fmaf(d, d, 1.0f));
  return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(0.5f * d));
}
__device__ __forceinline__ float ski_wj(const float *__restrict__ gp, int j) { return (ski_flags(gp) & 1) ? gp[4 + j] : 1.0f; }
// (g0, h, 1/h) of projection j
__device__ __forceinline__ const float *ski_grid_of(const float *__restrict__ gp, int J, int j) {
  return (ski_flags(gp) & 2) ? gp + 4 + J + 3 * j : gp;
}

// the grid coordinate of z, clamped into the interior (extrapolation guard): what the chunked product keeps in its plan
__device__ __forceinline__ float ski_grid_coord(float z, float g0, float inv_h, int G) {
  float u = (z - g0) * inv_h;
  return u < 1.0f ? 1.0f : (u > (float)(G - 2) ? (float)(G - 2) : u);
}

// taps idx0..idx0+3 and their weights for the clamped grid coordinate u; DERIV also returns d w_k / d z
template <bool DERIV>
__device__ __forceinline__ int ski_taps_u(float u, float inv_h, int G, float (&w)[4], float (&dw)[4]) {
  const float fl = __builtin_floorf(u);
  const float fr = u - fl;
  int idx0 = (int)fl - 1;
  idx0 = idx0 < 0 ? 0 : (idx0 > G - 4 ? G - 4 : idx0);
  const float s[4] = {fr + 1.0f, fr, 1.0f - fr, 2.0f - fr};          // |signed distance| of the 4 taps
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = cubic_w(s[k]);
  if constexpr (DERIV) {
    // signed distance s_k = fr + 1 - k: positive for k = 0,1; negative for k = 2,3;  dU/dz = sign / h
    dw[0] = cubic_dw(s[0]) * inv_h;
    dw[1] = cubic_dw(s[1]) * inv_h;
    dw[2] = -cubic_dw(s[2]) * inv_h;
    dw[3] = -cubic_dw(s[3]) * inv_h;
  }
  return idx0;
}

// the four tap weights from the tap fraction fr = u - floor(u): the arithmetic of ski_taps_u, so the same bits
__device__ __forceinline__ float4 ski_weights_of_frac(float fr) {
  return make_float4(cubic_w(fr + 1.0f), cubic_w(fr), cubic_w(1.0f - fr), cubic_w(2.0f - fr));
}

// taps idx0..idx0+3 and their weights for coordinate z; DERIV also returns d w_k / d z
template <bool DERIV>
__device__ __forceinline__ int ski_taps(float z, float g0, float inv_h, int G, float (&w)[4], float (&dw)[4]) {
  return ski_taps_u<DERIV>(ski_grid_coord(z, g0, inv_h, G), inv_h, G, w, dw);
}


}  // namespace
