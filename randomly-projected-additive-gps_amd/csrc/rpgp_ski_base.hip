// rpgp_ski_base.hip — the grid-interpolation (SKI) operator without a plan: grid construction, scatter / Toeplitz / gather
// stages for rectangular and square products, wide blocks, dense blocks, diagonal, and the staged derivative — the
// counterpart of `GridInterpolationKernel` around the 1-D sub-kernels (training_routines.py:157-158; SURVEY.md §8(f) rank 1,
// Appendix E).  The planned (cell-sorted) product of a whole CG solve lives in rpgp_ski.hip and reuses the Toeplitz and gather
// stages of this file through rpgp_internal.h.  Split out of rpgp_kernels.hip in round 6 (one object per concern).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stddef.h>

#include "../../include/rpgp.h"
#include "rpgp_internal.h"

namespace {

constexpr float kExp2Scale = 0.8493218002880191f;  // sqrt(0.5 * log2(e)):  exp(-d^2/2) = exp2(-(c d)^2)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// NaN-propagating min / max so that non-finite inputs are caught by the range guard
__device__ __forceinline__ float min_nan(float a, float b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
__device__ __forceinline__ float max_nan(float a, float b) { return (a != a) ? a : ((b != b) ? b : (b > a ? b : a)); }

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
#define RPGP_CHECK(expr)                          \
  do {                                            \
    hipError_t _e = (expr);                       \
    if (_e != hipSuccess) return (int)_e;         \
  } while (0)
inline int launch_status() { return (int)hipGetLastError(); }

// ---------------------------------------------------------------------------------------------
// SKI path (spec additive_spread_prescale_Jd_ski.json):
//   K_j ~= W_j Tm W_j^T,  W_j: cubic-convolution interpolation (Keys, 4 taps) of projection j onto ONE shared regular
//   1-D grid of G points, Tm: symmetric Toeplitz with first column exp(-0.5 (k h)^2).
// MVM = scatter (W^T v: per-(chunk, projection) LDS histograms in fixed point, slabs) -> Toeplitz product (MFMA) -> gather.
// HBM traffic ~ N (J + 2T) floats: this path is bandwidth/latency bound, not exp bound.
// grid params (device): gp[0] = g0 (first grid point), gp[1] = h (spacing), gp[2] = 1/h
// ---------------------------------------------------------------------------------------------
#include "rpgp_ski_common.h"   // cubic_w, cubic_dw, ski_wj, ski_taps (shared with rpgp_ski.hip)

// global min / max of all N x J projected coordinates of up to two arrays -> grid parameters
__global__ __launch_bounds__(256) void ski_minmax_kernel(const float *__restrict__ Z1, long long n1, int ld1,
                                                         const float *__restrict__ Z2, long long n2, int ld2, int J,
                                                         float *__restrict__ part) {
  __shared__ float smin[256], smax[256];
  float mn = 3.4e38f, mx = -3.4e38f;
  const long long t1 = n1 * J, t2 = n2 * J;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < t1 + t2; g += (long long)gridDim.x * 256) {
    float z;
    if (g < t1) z = Z1[(g / J) * ld1 + (g % J)];
    else { const long long q = g - t1; z = Z2[(q / J) * ld2 + (q % J)]; }
    mn = min_nan(mn, z);
    mx = max_nan(mx, z);
  }
  smin[threadIdx.x] = mn;
  smax[threadIdx.x] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      smin[threadIdx.x] = min_nan(smin[threadIdx.x], smin[threadIdx.x + w]);
      smax[threadIdx.x] = max_nan(smax[threadIdx.x], smax[threadIdx.x + w]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = smin[0];
    part[2 * blockIdx.x + 1] = smax[0];
  }
}

__global__ __launch_bounds__(64) void ski_grid_finish_kernel(const float *__restrict__ part, int nparts, int G,
                                                             float *__restrict__ gp) {
  // one wave: strided partial extrema, then a shuffle reduction (a single thread walking ~1000 partials took 47 us)
  float mn = 3.4e38f, mx = -3.4e38f;
  for (int p = threadIdx.x; p < nparts; p += 64) {
    mn = min_nan(mn, part[2 * p]);
    mx = max_nan(mx, part[2 * p + 1]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    mn = min_nan(mn, __shfl_xor(mn, off));
    mx = max_nan(mx, __shfl_xor(mx, off));
  }
  if (threadIdx.x != 0) return;
  float range = mx - mn;
  if (!(range > 1e-12f)) range = 1e-12f;          // all points identical (or NaN -> propagates through h)
  const float h = (mx - mn == mx - mn) ? range / (float)(G - 5) : (mx - mn);
  gp[0] = mn - 2.0f * h;                          // data lie in [g_2, g_{G-3}]: every 4-tap stencil is interior
  gp[1] = h;
  gp[2] = 1.0f / h;
  gp[3] = 0.f;
}

// Per-projection extrema (grid (chunks, J)) and the reference's grid rule (polynomial_projection_kernels.py:54-63):
//   spacing_j = (max_j - min_j) / (G - 4);  bounds_j = [min_j - 2.01 spacing_j, max_j + 2.01 spacing_j];
// the G grid points span the bounds uniformly, so the data keep a margin of 2.01 (G - 1) / (G + 0.02) ~ 2 cells (> 1.9 for
// G >= 16) on either side and every 4-tap stencil is interior.  (What GPyTorch's GridInterpolationKernel does with explicit bounds
// beyond this — it pads them by one more cell of its own — is not reproduced: GPyTorch is not available to pin it.)
__global__ __launch_bounds__(256) void ski_minmax_proj_kernel(const float *__restrict__ Z1, long long n1, int ld1,
                                                              const float *__restrict__ Z2, long long n2, int ld2, int J,
                                                              float *__restrict__ part) {
  __shared__ float smin[256], smax[256];
  const int j = blockIdx.y;
  float mn = 3.4e38f, mx = -3.4e38f;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n1 + n2; g += (long long)gridDim.x * 256) {
    const float z = g < n1 ? Z1[g * ld1 + j] : Z2[(g - n1) * ld2 + j];
    mn = min_nan(mn, z);
    mx = max_nan(mx, z);
  }
  smin[threadIdx.x] = mn;
  smax[threadIdx.x] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      smin[threadIdx.x] = min_nan(smin[threadIdx.x], smin[threadIdx.x + w]);
      smax[threadIdx.x] = max_nan(smax[threadIdx.x], smax[threadIdx.x + w]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[2 * ((size_t)j * gridDim.x + blockIdx.x)] = smin[0];
    part[2 * ((size_t)j * gridDim.x + blockIdx.x) + 1] = smax[0];
  }
}

__global__ __launch_bounds__(64) void ski_grid_finish_proj_kernel(const float *__restrict__ part, int nparts, int J, int G,
                                                                  float *__restrict__ gp) {
  // one wave per projection: strided partial extrema + shuffle reduction
  const int j = blockIdx.x;
  if (j == 0 && threadIdx.x == 0) {
    gp[0] = 0.f; gp[1] = 1.f; gp[2] = 1.f;      // (unused with per-projection grids)
    gp[3] = 2.f;                                // flags: per-projection grids, no weights
  }
  float mn = 3.4e38f, mx = -3.4e38f;
  for (int p = threadIdx.x; p < nparts; p += 64) {
    mn = min_nan(mn, part[2 * ((size_t)j * nparts + p)]);
    mx = max_nan(mx, part[2 * ((size_t)j * nparts + p) + 1]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    mn = min_nan(mn, __shfl_xor(mn, off));
    mx = max_nan(mx, __shfl_xor(mx, off));
  }
  if (threadIdx.x != 0) return;
  float range = mx - mn;
  if (!(range > 1e-12f)) range = 1e-12f;
  const bool finite = (mx - mn == mx - mn);
  const float spacing = range / (float)(G - 4);
  const float b0 = mn - 2.01f * spacing;
  // h from the clamped spacing, not from b1 - b0: for a constant column (mn == mx) the 2.01 spacing margins round away next
  // to |mn| in fp32, b1 - b0 = 0 and 1/h = inf turned every product of the operator into NaN; the floor keeps b0 + k h
  // distinct grid points at the magnitude of the coordinates.  (b1 - b0) / (G - 1) = spacing (G - 4 + 4.02) / (G - 1).
  float h = spacing * ((float)(G - 4) + 4.02f) / (float)(G - 1);
  const float hmin = fmaxf(fabsf(mn), fabsf(mx)) * 2.4e-7f;
  if (h < hmin) h = hmin;
  if (!finite) h = mx - mn;
  gp[4 + j] = 1.0f;                             // weight slot (ones until the host sets flags |= 1 and fills them)
  float *gj = gp + 4 + J + 3 * j;
  gj[0] = b0;
  gj[1] = h;
  gj[2] = 1.0f / h;
}

// Scatter for T <= 12, parallel over projections AND point chunks: workgroup (chunk, j) accumulates its chunk's
// contributions to projection j's histogram in LDS and stores it as a slab; ski_slab_sum_kernel adds the slabs.
// Lanes are laid out (point, t): LPP = 1 / 4 / 16 lanes per point, so one LDS atomic instruction updates the TT
// consecutive words of a grid cell for 256 / LPP points (conflict-free across t, V read coalesced).
// The LDS accumulation is INTEGER: gfx950 executes ds_add_f32 at ~80 ns per wave-instruction per CU against 2.6 ns for
// ds_add_u32 (tools/lds_atomic_bench.hip, profiles/r1_lds_atomic_bench.txt).  Each column gets a power-of-two
// fixed-point scale from the chunk's own max|v| and its densest 4-cell neighbourhood (so no cell sum can overflow 2^30);
// the rounding error per update is <= 2^-31 of that bound — at the fp32 rounding level of the float sum it replaces —
// and integer addition commutes, so the SKI product is bitwise reproducible.
template <int TT, bool CNT16>
__global__ __launch_bounds__(256) void ski_scatter3_kernel(const float *__restrict__ Z, const float *__restrict__ gp,
                                                           const float *__restrict__ V, float *__restrict__ slab,
                                                           long long N, int ldz, int J, int G, int T, int tcnt,
                                                           long long pts_per_chunk) {
  // G * TT fixed-point accumulators | G/2 ints: points per first-tap cell, two 16-bit counters per word (scale bound).
  // 16-bit counters keep the T = 12 workgroup at 53.4 KB of LDS: three per CU (with 32-bit counters it was 55.4 KB ->
  // two per CU and the (chunk, projection) grid ran in two rounds); the host keeps chunks below 65 536 points.
  extern __shared__ int shi[];
  __shared__ float smax[256];
  __shared__ int scmax[256];
  __shared__ float sscale[16], sinv[16];
  constexpr int LPP = TT == 1 ? 1 : (TT == 4 ? 4 : 16);
  constexpr int PPI = 256 / LPP;
  const int j = blockIdx.y;
  const float *gj = ski_grid_of(gp, J, j);
  const float g0 = gj[0], inv_h = gj[2];
  const long long n0 = (long long)blockIdx.x * pts_per_chunk;
  const long long n1 = (n0 + pts_per_chunk < N) ? n0 + pts_per_chunk : N;
  int *scnt = shi + G * TT;
  for (int e = threadIdx.x; e < G * TT + (CNT16 ? (G + 1) / 2 : G); e += 256) shi[e] = 0;
  __syncthreads();
  const int t = threadIdx.x % LPP, pl = threadIdx.x / LPP;
  // pass 1: max |v| of the chunk per column, and how many points start their 4-tap stencil at each grid cell: a cell
  // receives at most one tap (|w| <= 1) from every point whose stencil starts in [cell - 3, cell], so
  //   |cell sum| <= (max over cells of that 4-cell count) * max|v|
  // — a bound ~100x tighter than points * max|v|, i.e. a fixed-point quantum at the fp32 rounding level
  float vm = 0.f;
  if (t < tcnt) {
    // 8 independent loads in flight per thread (the loop is a chain of dependent-latency loads otherwise)
    long long i = n0 + pl;
    for (; i + 15LL * PPI < n1; i += 16LL * PPI) {
      float x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) x[u] = V[(i + (long long)u * PPI) * T + t];
#pragma unroll
      for (int u = 0; u < 16; ++u) vm = max_nan(vm, __builtin_fabsf(x[u]));
    }
    for (; i < n1; i += PPI) vm = max_nan(vm, __builtin_fabsf(V[i * T + t]));
  }
  smax[threadIdx.x] = vm;
  {
    long long i = n0 + threadIdx.x;
    for (; i + 3 * 256 < n1; i += 4 * 256) {
      float z[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) z[u] = Z[(i + u * 256) * ldz + j];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float w[4], dw[4];
        const int c = ski_taps<false>(z[u], g0, inv_h, G, w, dw);
        if constexpr (CNT16) atomicAdd(&scnt[c >> 1], 1 << (16 * (c & 1)));
        else atomicAdd(&scnt[c], 1);
      }
    }
    for (; i < n1; i += 256) {
      float w[4], dw[4];
      const int c = ski_taps<false>(Z[i * ldz + j], g0, inv_h, G, w, dw);
      if constexpr (CNT16) atomicAdd(&scnt[c >> 1], 1 << (16 * (c & 1)));
      else atomicAdd(&scnt[c], 1);
    }
  }
  __syncthreads();
  auto cell_count = [&](int g) {
    if constexpr (CNT16) return (int)(((unsigned)scnt[g >> 1] >> (16 * (g & 1))) & 0xffffu);
    else return scnt[g];
  };
  int cm = 0;
  for (int g = threadIdx.x; g < G; g += 256) {
    int c4 = cell_count(g);
    if (g >= 1) c4 += cell_count(g - 1);
    if (g >= 2) c4 += cell_count(g - 2);
    if (g >= 3) c4 += cell_count(g - 3);
    cm = c4 > cm ? c4 : cm;
  }
  scmax[threadIdx.x] = cm;
  __syncthreads();
  if ((int)threadIdx.x < LPP) {
    float m = 0.f;
    for (int q = 0; q < PPI; ++q) m = max_nan(m, smax[q * LPP + threadIdx.x]);
    int cmax = 1;
    for (int q = 0; q < 256; ++q) cmax = scmax[q] > cmax ? scmax[q] : cmax;
    const float bound = 1.05f * (float)cmax * m;
    float sc = 1.0f, inv = 1.0f;
    if (bound > 0.f && bound < 3.0e38f) {
      int ex;
      (void)frexpf(bound, &ex);                       // bound < 2^ex
      sc = ldexpf(1.0f, 30 - ex);
      inv = ldexpf(1.0f, ex - 30);
    } else if (!(bound == bound) || bound >= 3.0e38f) {
      // a NaN / Inf entry in this chunk's column: integer accumulation would silently drop it (NaN converts to 0) and
      // the product would stay finite — poison the chunk's histogram column instead so that it propagates like in the
      // exact operator (and the CG NaN guard sees it)
      inv = __builtin_nanf("");
    }
    sscale[threadIdx.x] = sc;
    sinv[threadIdx.x] = inv;
  }
  __syncthreads();
  if (t < tcnt) {
    const float sc = sscale[t];
    constexpr int U = TT > 4 ? 8 : 4;      // points in flight per thread (hides the Z / V load latency; the wide form
                                           // runs only 3 workgroups per CU, so the depth has to come from each thread)
    for (long long i0 = n0 + pl; i0 < n1; i0 += (long long)PPI * U) {
      float zv[U], vv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = i0 + (long long)u * PPI;
        const bool ok = i < n1;
        zv[u] = ok ? Z[i * ldz + j] : 0.f;
        vv[u] = ok ? V[i * T + t] * sc : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i0 + (long long)u * PPI < n1) {
          float w[4], dw[4];
          const int idx0 = ski_taps<false>(zv[u], g0, inv_h, G, w, dw);
#pragma unroll
          for (int k = 0; k < 4; ++k) atomicAdd(&shi[(idx0 + k) * TT + t], __float2int_rn(w[k] * vv[u]));
        }
      }
    }
  }
  __syncthreads();
  float *dst = slab + ((size_t)blockIdx.x * J + j) * G * TT;
  for (int e = threadIdx.x; e < G * TT; e += 256) {
    const int tt = e % TT;
    dst[e] = tt < tcnt ? (float)shi[e] * sinv[tt] : 0.f;
  }
}

// hist[j][g][hoff + t] = sum_c slab[c][j][g][t]   (t < tcnt; hist row stride HT).  The grid histogram and the Toeplitz
// product below are carried in FLOAT64: they cost nothing (J G T values), and with |K| ~ s N the fp32 rounding of these
// two small stages alone (~1e-6 |K v|) exceeds sigma^2 |v| once N s / sigma^2 reaches a few million — at N = 391k the
// operator then stops being numerically positive definite and CG diverges.
__global__ __launch_bounds__(256) void ski_slab_sum_kernel(const float *__restrict__ slab, double *__restrict__ hist,
                                                           int nchunks, int J, int G, int TT, int tcnt, int HT,
                                                           int hoff) {
  // A workgroup owns 32 consecutive histogram entries; its 8 groups of 32 threads split the chunk slabs (slab c goes to
  // group c % 8: each load is one 128-byte segment, 4 in flight per thread) and the 8 partial sums are added in a fixed
  // order.  (One thread per entry looping over ~340 slabs was a serial chain of dependent-latency loads: 80 us at the
  // C5 shape, J = 3, against 13 us with J = 20 where there are 20x more entries and 7x fewer slabs.)
  __shared__ double part[8][32];
  const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
  const size_t per = (size_t)J * G * TT;
  const size_t e = (size_t)blockIdx.x * 32 + o;
  double acc = 0.0;
  if (e < per) {
    int c = g;
    for (; c + 24 < nchunks; c += 32) {
      const float x0 = slab[(size_t)c * per + e], x1 = slab[(size_t)(c + 8) * per + e];
      const float x2 = slab[(size_t)(c + 16) * per + e], x3 = slab[(size_t)(c + 24) * per + e];
      acc += (double)x0;
      acc += (double)x1;
      acc += (double)x2;
      acc += (double)x3;
    }
    for (; c < nchunks; c += 8) acc += (double)slab[(size_t)c * per + e];
  }
  part[g][o] = acc;
  __syncthreads();
  if (g != 0 || e >= per) return;
  const int t = (int)(e % TT);
  if (t >= tcnt) return;
  double tot = part[0][o];
#pragma unroll
  for (int q = 1; q < 8; ++q) tot += part[q][o];
  hist[(e / TT) * HT + hoff + t] = tot;
}

// H[j][m][t] = sum_m' exp(-0.5 ((m - m') h)^2) hist[j][m'][t]
template <class HT_>
__global__ __launch_bounds__(256) void ski_toeplitz_kernel(const HT_ *__restrict__ hist, const float *__restrict__ gp,
                                                           float *__restrict__ H, int G, int T) {
  extern __shared__ double scd[];   // G toeplitz coefficients (float64)
  double *sc = scd;
  const double hd = (double)ski_grid_of(gp, gridDim.y, blockIdx.y)[1];
  const int kind = ski_kind(gp);
  for (int k = threadIdx.x; k < G; k += 256) sc[k] = ski_radial_f64(kind, (double)k * hd);
  __syncthreads();
  const int j = blockIdx.y;
  // thread -> (m, t): 256 threads cover (256 / Tp) rows x Tp columns, Tp = T rounded up to a power of two <= 16
  int Tp = 1;
  while (Tp < T && Tp < 16) Tp <<= 1;
  const int rows_per_block = 256 / Tp;
  const int m = blockIdx.x * rows_per_block + threadIdx.x / Tp;
  if (m >= G) return;
  for (int t = threadIdx.x % Tp; t < T; t += Tp) {
    const HT_ *hj = hist + (size_t)j * G * T + t;
    double acc = 0.0;
    for (int mp = 0; mp < G; ++mp) {
      const int k = m > mp ? m - mp : mp - m;
      acc = fma(sc[k], (double)hj[(size_t)mp * T], acc);
    }
    H[((size_t)j * G + m) * T + t] = (float)(acc * (double)ski_wj(gp, j));
  }
}

// Toeplitz product on the matrix cores (16 right-hand sides per workgroup, T <= 64), in float64: H_j (G x T) = Toep(G x G) @ hist_j (G x T) as 16 x 16
// output tiles, v_mfma_f64_16x16x4_f64 over the G grid points (K = 4 per issue).  One wave per 16-row tile, four tiles
// per workgroup; hist_j is staged through LDS in panels of 512 grid rows; the Toeplitz entry sc[|m - k|] is read from LDS.
//   A (16x4): lane l holds Toep[m0 + l%16][k0 + l/16]     B (4x16): lane l holds hist_j[k0 + l/16][l%16]
//   D (16x16): lane l holds H_j[m0 + l/16 + 4 r][l%16], r = 0..3   (NOT the fp32 instruction's 4*(l/16) + r)
typedef double doublex4m __attribute__((ext_vector_type(4)));
// One workgroup owns ONE 16-row output tile; its 4 waves split the grid points (the K loop) four ways and add their
// partial tiles through LDS in a fixed order.  (The first version gave each wave its own tile and the whole K loop:
// 64 dependent K-steps x 4 MFMAs per wave and only 16 J workgroups — 31 us of the 66 us SKI MVM at the C5 shape.)
template <int NW>
__global__ __launch_bounds__(64 * NW) void ski_toeplitz_mfma_kernel(const double *__restrict__ hist,
                                                                const float *__restrict__ gp, float *__restrict__ H,
                                                                int G, int T, const double *__restrict__ tcol) {
  extern __shared__ double dmem[];          // sc[G16] | red[NW - 1][256]
  const int G16 = (G + 15) & ~15;
  double *sc = dmem;
  double *red = dmem + G16;
  const int j = blockIdx.y;
  const double hd = (double)ski_grid_of(gp, gridDim.y, j)[1];
  if (tcol) {                               // first column of the Toeplitz matrix from the per-step plan (no exp here)
    const double *tc = tcol + (size_t)((ski_flags(gp) & 2) ? j : 0) * G16;
    for (int k = threadIdx.x; k < G16; k += 64 * NW) sc[k] = k < G ? tc[k] : 0.0;
  } else {
    const int kind = ski_kind(gp);
    for (int k = threadIdx.x; k < G16; k += 64 * NW) sc[k] = k < G ? ski_radial_f64(kind, (double)k * hd) : 0.0;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 16;
  const int mrow = m0 + (lane & 15), q = lane >> 4;
  const int nb = 16 * blockIdx.z + (lane & 15);      // blockIdx.z: 16-column block of the right-hand sides (T > 16)
  const int nbc = nb < T ? nb : T - 1;               // clamped column: every load below is unconditional
  const double bmask = nb < T ? 1.0 : 0.0;
  const double amask = (mrow < G) ? 1.0 : 0.0;
  const double *hj = hist + (size_t)j * G * T;
  // this wave's share of the grid points, in steps of 16 (4 MFMAs on 4 independent accumulators per step)
  const int ksteps = G16 / 16;
  const int s_begin = (ksteps * wave) / NW, s_end = (ksteps * (wave + 1)) / NW;
  doublex4m acc4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) acc4[u] = doublex4m{0.0, 0.0, 0.0, 0.0};
  // The B operands (hist_j, L2-resident) of EIGHT steps are requested together: with one step's four loads per
  // iteration the loop was a chain of 16 dependent L2 round trips (14 us for a 3 x 1024 x 1024 x 11 product).
  constexpr int SB = NW > 4 ? 4 : 8;
  for (int st0 = s_begin; st0 < s_end; st0 += SB) {
    double b[SB][4];
#pragma unroll
    for (int ss = 0; ss < SB; ++ss) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = (st0 + ss) * 16 + 4 * u + q;
        const bool ok = k < G && st0 + ss < s_end;
        const int kc = ok ? k : 0;
        const double x = hj[(size_t)kc * T + nbc];
        b[ss][u] = ok ? x * bmask : 0.0;
      }
    }
    if (st0 == s_begin) __syncthreads();              // sc[] is complete (the loads above are already in flight)
#pragma unroll
    for (int ss = 0; ss < SB; ++ss) {
      if (st0 + ss < s_end) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = (st0 + ss) * 16 + 4 * u + q;
          int dist = mrow > k ? mrow - k : k - mrow;
          dist = dist < G16 ? dist : G16 - 1;           // only rows >= G can exceed it; they carry amask = 0
          acc4[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(sc[dist] * amask, b[ss][u], acc4[u], 0, 0, 0);
        }
      }
    }
  }
  if (s_begin >= s_end) __syncthreads();
  doublex4m acc = acc4[0] + acc4[1] + acc4[2] + acc4[3];
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave - 1) * 4 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0 || m0 >= G) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int ww = 0; ww < NW - 1; ++ww) acc[r] += red[(ww * 4 + r) * 64 + lane];       // fixed order
  }
  const double wj = (double)ski_wj(gp, j);
  if (nb < T) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + q + 4 * r;          // float64 MFMA result layout (tools/mfma_f64_probe.hip): row = l/16 + 4 r
      if (m < G) H[((size_t)j * G + m) * T + nb] = (float)(acc[r] * wj);
    }
  }
}

// out[i][t] = scale * sum_j sum_k w_k(z_ij) H[j][idx0+k][t] + noise * V[i][t]
// Lanes are laid out (point, t) with LPP = 1 / 4 / 16 lanes per point: a tap's T values are one contiguous segment for
// the point's lanes (a lane-per-point layout touched 64 cache lines per load instruction and was bound by the L1
// transaction rate: 85 us at N = 50k, J = 20, T = 11).
template <int TT>
__global__ __launch_bounds__(256) void ski_gather_kernel(const float *__restrict__ Z, const float *__restrict__ gp,
                                                         const float *__restrict__ H, const float *__restrict__ V,
                                                         float *__restrict__ out, long long M, int ldz, int J, int G,
                                                         int T, int t0, int tcnt, float scale, float noise) {
  constexpr int LPP = TT == 1 ? 1 : (TT == 4 ? 4 : 16);
  constexpr int PPB = 256 / LPP;
  // The LPP lanes of a point share its stencils: lane t computes the taps of projection j0 + t ONCE and parks them in
  // LDS; every lane then reads (idx0, 4 weights) per projection as an LDS broadcast.  (Each lane recomputing all J
  // stencils and converting every term to float64 made this kernel VALU-bound: 40 us for 39 MB at the C5 shape.)
  __shared__ float sTap[LPP > 1 ? PPB * LPP * 5 : 1];
  const int t = threadIdx.x % LPP, pl = threadIdx.x / LPP;
  const long long i = (long long)blockIdx.x * PPB + pl;
  const bool live = i < M;
  const bool writer = live && t < tcnt;
  const float *zrow = Z + (live ? i : 0) * ldz;
  double acc = 0.0;                          // J terms per output in float64; the 4 taps of a projection in fp32 FMAs
  if constexpr (LPP == 1) {
    if (!writer) return;
#pragma unroll 4
    for (int j = 0; j < J; ++j) {
      float w[4], dw[4];
      const float *gj = ski_grid_of(gp, J, j);
      const int idx0 = ski_taps<false>(zrow[j], gj[0], gj[2], G, w, dw);
      const float *hp = H + ((size_t)j * G + idx0) * T + t0;
      float p = w[0] * hp[0];
      p = __builtin_fmaf(w[1], hp[(size_t)T], p);
      p = __builtin_fmaf(w[2], hp[2 * (size_t)T], p);
      p = __builtin_fmaf(w[3], hp[3 * (size_t)T], p);
      acc += (double)p;
    }
  } else {
    for (int j0 = 0; j0 < J; j0 += LPP) {
      const int jj = j0 + t;
      if (jj < J) {
        float w[4], dw[4];
        const float *gj = ski_grid_of(gp, J, jj);
        const int idx0 = live ? ski_taps<false>(zrow[jj], gj[0], gj[2], G, w, dw) : 0;
        float *dst = sTap + (pl * LPP + t) * 5;
        dst[0] = __builtin_bit_cast(float, idx0);
        dst[1] = w[0]; dst[2] = w[1]; dst[3] = w[2]; dst[4] = w[3];
      }
      __syncthreads();
      const int jn = (J - j0 < LPP) ? J - j0 : LPP;
      if (writer) {
        for (int q = 0; q < jn; ++q) {
          const float *tp = sTap + (pl * LPP + q) * 5;
          const int idx0 = __builtin_bit_cast(int, tp[0]);
          const float *hp = H + ((size_t)(j0 + q) * G + idx0) * T + t0 + t;
          float p = tp[1] * hp[0];
          p = __builtin_fmaf(tp[2], hp[(size_t)T], p);
          p = __builtin_fmaf(tp[3], hp[2 * (size_t)T], p);
          p = __builtin_fmaf(tp[4], hp[3 * (size_t)T], p);
          acc += (double)p;
        }
      }
      __syncthreads();
    }
    if (!writer) return;
  }
  float r = scale * (float)acc;
  if (noise != 0.f) r = __builtin_fmaf(noise, V[i * T + t0 + t], r);
  out[i * T + t0 + t] = r;
}

// ---- wide right-hand sides (T > 12: predictive covariance blocks, dense evaluation): lane = column t, so every
// histogram update / read is a 256-byte contiguous wave access (the efficient shape for float atomics) ------------
__global__ __launch_bounds__(256) void ski_scatter_wide_kernel(const float *__restrict__ Z, const float *__restrict__ gp,
                                                               const float *__restrict__ V, float *__restrict__ hist,
                                                               long long N, int ldz, int J, int G, int T, int HT,
                                                               int hoff, long long pts_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.y * 64 + lane;
  const long long n0 = (long long)blockIdx.x * pts_per_block;
  const long long n1 = (n0 + pts_per_block < N) ? n0 + pts_per_block : N;
  if (t >= T) return;
  for (long long i = n0 + wave; i < n1; i += 4) {
    const float v = V[i * T + t];
    for (int j = 0; j < J; ++j) {
      float w[4], dw[4];
      const float *gj = ski_grid_of(gp, J, j);
      const int idx0 = ski_taps<false>(Z[i * ldz + j], gj[0], gj[2], G, w, dw);
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(&hist[((size_t)j * G + idx0 + k) * HT + hoff + t], w[k] * v);
    }
  }
}

__global__ __launch_bounds__(256) void ski_toeplitz_wide_kernel(const float *__restrict__ hist,
                                                                const float *__restrict__ gp, float *__restrict__ H,
                                                                int G, int T) {
  extern __shared__ float sc[];   // G toeplitz coefficients
  const float hgrid = ski_grid_of(gp, gridDim.z, blockIdx.z)[1];
  const int kind = ski_kind(gp);
  for (int k = threadIdx.x; k < G; k += 256) sc[k] = ski_radial_f32(kind, k, hgrid);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  const int j = blockIdx.z;
  const int m0 = (blockIdx.y * 4 + wave) * 4;     // 4 rows per wave, 16 per block
  if (t >= T || m0 >= G) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float *hj = hist + (size_t)j * G * T + t;
  for (int mp = 0; mp < G; ++mp) {
    const float hv = hj[(size_t)mp * T];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + r;
      const int k = m > mp ? m - mp : mp - m;
      acc[r] = __builtin_fmaf(sc[k < G ? k : G - 1], hv, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (m0 + r < G) H[((size_t)j * G + m0 + r) * T + t] = acc[r] * ski_wj(gp, j);
}

__global__ __launch_bounds__(256) void ski_gather_wide_kernel(const float *__restrict__ Z, const float *__restrict__ gp,
                                                              const float *__restrict__ H, const float *__restrict__ V,
                                                              float *__restrict__ out, long long M, int ldz, int J,
                                                              int G, int T, float scale, float noise,
                                                              long long pts_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.y * 64 + lane;
  const long long n0 = (long long)blockIdx.x * pts_per_block;
  const long long n1 = (n0 + pts_per_block < M) ? n0 + pts_per_block : M;
  if (t >= T) return;
  for (long long i = n0 + wave; i < n1; i += 4) {
    float acc = 0.f;
    for (int j = 0; j < J; ++j) {
      float w[4], dw[4];
      const float *gj = ski_grid_of(gp, J, j);
      const int idx0 = ski_taps<false>(Z[i * ldz + j], gj[0], gj[2], G, w, dw);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = __builtin_fmaf(w[k], H[((size_t)j * G + idx0 + k) * T + t], acc);
    }
    float r = scale * acc;
    if (noise != 0.f) r = __builtin_fmaf(noise, V[i * T + t], r);
    out[i * T + t] = r;
  }
}

// Derivative gather.  H holds Toeplitz-smoothed histograms of the 2T columns [L | R]:
//   gZ[i][j] = scale * sum_k dw_k(z_ij) * sum_t ( L[i,t] H_R[j][idx+k][t] + R[i,t] H_L[j][idx+k][t] )
//   rowS[b]  = sum over workgroup b's rows i of  sum_t L[i,t] * sum_j sum_k w_k H_R[j][idx+k][t]   (= L[i,:] . (K R)[i,:] / scale)
// (round 6: one partial per WORKGROUP, fixed-order tree — the N per-row values were only ever added up, by one workgroup
//  walking 391k floats: 34 us at C5)
template <int TT>
__global__ __launch_bounds__(256) void ski_grad_gather_kernel(const float *__restrict__ Z, const float *__restrict__ gp,
                                                              const float *__restrict__ H, const float *__restrict__ L,
                                                              const float *__restrict__ Rm, float *__restrict__ gZ,
                                                              float *__restrict__ rowS, long long N, int ldz, int ldg,
                                                              int J, int G, int T, float scale, float *__restrict__ rowC) {
  __shared__ float ssum[256];
  const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool valid = i0 < N;
  const long long i = valid ? i0 : N - 1;
  // (loads from clamped indices, slots t >= T masked through li / ri = 0: a load under `t < T` compiles to a branch with its
  //  own wait — 88 dependent round trips per projection in the loop below, which made this kernel latency-bound)
  float li[TT], ri[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int tc = t < T ? t : T - 1;
    const float m = t < T ? 1.f : 0.f;
    li[t] = L[i * T + tc] * m;
    ri[t] = Rm[i * T + tc] * m;
  }
  float accS = 0.f;
  const int T2 = 2 * T;
  // The training block (10 probes + the residual: T = 11): a grid row is 22 floats = 88 bytes, 8-byte aligned — eleven 8-byte
  // loads instead of twenty-two 4-byte ones.  The lanes of a wave sit in 64 different grid rows, so every load instruction is
  // 64 separate requests to the texture path whatever its width, and that path's request rate is what this kernel runs at
  // (C5: 391k points x 3 projections x 4 taps x 22 four-byte requests = 140 us; the L2 holds all of H).
  const bool pairs = TT == 12 && T == 11 && ((reinterpret_cast<uintptr_t>(H) & 7) == 0);
  for (int j = 0; j < J; ++j) {
    float w[4], dw[4];
    const float *gj = ski_grid_of(gp, J, j);
    const int idx0 = ski_taps<true>(Z[i * ldz + j], gj[0], gj[2], G, w, dw);
    float gz = 0.f, accj = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float *hp = H + ((size_t)j * G + idx0 + k) * T2;   // [H_L (T) | H_R (T)]
      float hl[TT], hr[TT];
      if (TT == 12 && pairs) {
        float flat[24];
        const float2 *hp2 = reinterpret_cast<const float2 *>(hp);
#pragma unroll
        for (int q = 0; q < 11; ++q) {
          const float2 v = hp2[q];
          flat[2 * q] = v.x;
          flat[2 * q + 1] = v.y;
        }
        flat[22] = flat[23] = 0.f;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          hl[t] = t < 11 ? flat[t] : 0.f;                 // (slot 11 meets li / ri = 0 either way)
          hr[t] = t < 11 ? flat[11 + t] : 0.f;
        }
      } else {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const int tc = t < T ? t : T - 1;
          hr[t] = hp[T + tc];
          hl[t] = hp[tc];
        }
      }
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        a = __builtin_fmaf(li[t], hr[t], a);            // L . H_R
        b = __builtin_fmaf(ri[t], hl[t], b);            // R . H_L
      }
      gz = __builtin_fmaf(dw[k], a + b, gz);
      accj = __builtin_fmaf(w[k], a, accj);
    }
    if (valid) {
      gZ[i * ldg + j] = scale * gz;
      if (rowC) rowC[i * J + j] = accj;      // per-projection part (H already carries the projection's weight)
    }
    accS += accj;
  }
  ssum[threadIdx.x] = valid ? accS : 0.f;
  __syncthreads();
  for (int wd = 128; wd > 0; wd >>= 1) {
    if ((int)threadIdx.x < wd) ssum[threadIdx.x] += ssum[threadIdx.x + wd];
    __syncthreads();
  }
  if (threadIdx.x == 0) rowS[blockIdx.x] = ssum[0];
}

// Dense block of the SKI operator: out[m][n] = scale * sum_j w_j sum_{q,q'} w_q(z1_mj) w_q'(z2_nj) Toep[(idx_m + q) - (idx_n + q')]
// (7 distinct lags per column, looked up in an LDS copy of the Toeplitz column).  One thread per output column, 16
// rows per workgroup with their taps in LDS.  Used for `to_dense`, row gathers and — below N ~ 32k, where a library
// GEMM on the dense matrix beats the wide scatter/gather — the wide solves of the predictive covariance.
__global__ __launch_bounds__(256) void ski_dense_kernel(const float *__restrict__ Z1, const float *__restrict__ Z2,
                                                        const float *__restrict__ gp, float *__restrict__ out, int M,
                                                        int N, int ldz1, int ldz2, long long ldo, int J, int G,
                                                        float scale) {
  constexpr int RT = 16;
  extern __shared__ float smem[];           // sc[G] | sW[RT][J][4] | sI[RT][J] (ints)
  float *sc = smem;
  float *sW = smem + G;
  int *sI = reinterpret_cast<int *>(sW + RT * J * 4);
  const bool per_proj = (ski_flags(gp) & 2) != 0;     // per-projection grids: the 7 lags are evaluated on the fly
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * RT;
  const int kind = ski_kind(gp);
  if (!per_proj) {
    for (int q = tid; q < G; q += 256) sc[q] = ski_radial_f32(kind, q, gp[1]);
  }
  for (int e = tid; e < RT * J; e += 256) {
    const int r = e / J, j = e % J;
    float w[4] = {0.f, 0.f, 0.f, 0.f}, dw[4];
    int idx = 0;
    const float *gj = ski_grid_of(gp, J, j);
    if (m0 + r < M) idx = ski_taps<false>(Z1[(size_t)(m0 + r) * ldz1 + j], gj[0], gj[2], G, w, dw);
    sI[e] = idx;
#pragma unroll
    for (int q = 0; q < 4; ++q) sW[e * 4 + q] = w[q];
  }
  __syncthreads();
  const int col = blockIdx.x * 256 + tid;
  if (col >= N) return;
  float acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = 0.f;
  for (int j = 0; j < J; ++j) {
    float wc[4], dw[4];
    const float *gj = ski_grid_of(gp, J, j);
    const int idc = ski_taps<false>(Z2[(size_t)col * ldz2 + j], gj[0], gj[2], G, wc, dw);
    const float wj = ski_wj(gp, j);
    const float hj = gj[1];
#pragma unroll 4
    for (int r = 0; r < RT; ++r) {
      const int delta = sI[r * J + j] - idc;
      const float *wr = sW + (r * J + j) * 4;
      float tl[7];
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        int lag = delta + u - 3;
        lag = lag < 0 ? -lag : lag;
        if (per_proj) {
          tl[u] = ski_radial_f32(kind, lag, hj);
        } else {
          tl[u] = lag < G ? sc[lag] : 0.f;
        }
      }
      float aj = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) aj = __builtin_fmaf(wr[q] * wc[qq], tl[q - qq + 3], aj);
      acc[r] = __builtin_fmaf(wj, aj, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < RT; ++r)
    if (m0 + r < M) out[(size_t)(m0 + r) * ldo + col] = scale * acc[r];
}

// diag[i] = scale * sum_j sum_{k,k'} w_k w_k' exp(-0.5 ((k-k') h)^2)
__global__ __launch_bounds__(256) void ski_diag_kernel(const float *__restrict__ Z, const float *__restrict__ gp,
                                                       float *__restrict__ diag, long long N, int ldz, int J, int G,
                                                       float scale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float acc = 0.f;
  for (int j = 0; j < J; ++j) {
    float w[4], dw[4];
    const float *gj = ski_grid_of(gp, J, j);
    float c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = ski_radial_f32(ski_kind(gp), k, gj[1]);
    ski_taps<false>(Z[i * ldz + j], gj[0], gj[2], G, w, dw);
    float aj = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) aj = __builtin_fmaf(w[k] * w[kk], c[k > kk ? k - kk : kk - k], aj);
    acc = __builtin_fmaf(ski_wj(gp, j), aj, acc);
  }
  diag[i] = scale * acc;
}


}  // namespace

// ------------------------------------ SKI entry points ----------------------------------------
namespace {
inline int ski_tpiece(int remaining) { return remaining > 4 ? 12 : (remaining > 1 ? 4 : 1); }
constexpr int kSkiMaxParts = 512;

// chunks of points per projection for ski_scatter3_kernel: at least 256 points each and at most ~1024 workgroups in
// total (the slab workspace is sized for that); with wide right-hand sides the (chunk, projection) workgroups are
// limited to what is resident at once — the LDS histogram of a T = 12 workgroup is 53 KB, three per CU — so that the
// launch is ONE round of equally long workgroups instead of 1.3 (measured at the C5 shape: 130 -> 103 us)
inline int ski_max_chunks(int J) { return (1024 + J - 1) / J; }
inline int ski_chunks(long long N, int J, int tt = 1) {
  long long c = ski_max_chunks(J);
  if (tt > 4) {
    const long long resident = (768 + J - 1) / J;
    if (c > resident) c = resident;
  }
  const long long by_pts = (N + 255) / 256;
  if (c > by_pts) c = by_pts;
  if (c < 1) c = 1;
  return (int)c;
}
inline size_t ski_slab_floats(int J, int G) { return (size_t)ski_max_chunks(J) * J * G * 12; }

// hist[j][g][hoff + t] (row stride HT) = sum_i w(z_ij)[g] V[i][t], T <= 12 columns of V (row stride T); no atomics on hist
int ski_scatter_narrow(const float *Z, const float *gp, const float *V, double *hist, float *slab, long long N, int ldz,
                       int J, int G, int T, int HT, int hoff, hipStream_t st) {
  const int tt = ski_tpiece(T);
  const int nch = ski_chunks(N, J, tt);
  const long long ppc = (N + nch - 1) / nch;
  dim3 grid((unsigned)nch, (unsigned)J);
  // 16-bit per-cell point counters (two per LDS word) for the wide form when a chunk has < 65 536 points: 53.4 KB of
  // LDS per workgroup, three per CU
  const bool cnt16 = tt > 4 && ppc < 65536;
  const size_t lds = ((size_t)G * tt + (cnt16 ? (G + 1) / 2 : G)) * sizeof(float);
  if (tt == 1)
    hipLaunchKernelGGL((ski_scatter3_kernel<1, false>), grid, dim3(256), lds, st, Z, gp, V, slab, N, ldz, J, G, T, T, ppc);
  else if (tt == 4)
    hipLaunchKernelGGL((ski_scatter3_kernel<4, false>), grid, dim3(256), lds, st, Z, gp, V, slab, N, ldz, J, G, T, T, ppc);
  else if (cnt16)
    hipLaunchKernelGGL((ski_scatter3_kernel<12, true>), grid, dim3(256), lds, st, Z, gp, V, slab, N, ldz, J, G, T, T, ppc);
  else
    hipLaunchKernelGGL((ski_scatter3_kernel<12, false>), grid, dim3(256), lds, st, Z, gp, V, slab, N, ldz, J, G, T, T, ppc);
  int rc = launch_status();
  if (rc) return rc;
  const size_t per = (size_t)J * G * tt;
  hipLaunchKernelGGL(ski_slab_sum_kernel, dim3((unsigned)((per + 31) / 32)), dim3(256), 0, st, slab, hist, nch, J, G,
                     tt, T, HT, hoff);
  return launch_status();
}

// T <= 12: float64 histogram (returns 1 in *hist_is_double); wider blocks: float histogram with global float atomics
int ski_scatter_all(const float *Z, const float *gp, const float *V, float *hist, float *slab, long long N, int ldz,
                    int J, int G, int T, hipStream_t st, int *hist_is_double) {
  *hist_is_double = T <= 12;
  if (T > 12) {
    RPGP_CHECK(hipMemsetAsync(hist, 0, (size_t)J * G * T * sizeof(float), st));
    long long nblk = (N + 255) / 256;
    if (nblk > 2048) nblk = 2048;
    const long long ppb = (N + nblk - 1) / nblk;
    hipLaunchKernelGGL(ski_scatter_wide_kernel, dim3((unsigned)nblk, (unsigned)((T + 63) / 64)), dim3(256), 0, st, Z,
                       gp, V, hist, N, ldz, J, G, T, T, 0, ppb);
    return launch_status();
  }
  return ski_scatter_narrow(Z, gp, V, reinterpret_cast<double *>(hist), slab, N, ldz, J, G, T, T, 0, st);
}

int ski_toeplitz(const void *hist, int hist_is_double, const float *gp, float *H, int J, int G, int T, hipStream_t st,
                 const double *tcol = nullptr) {
  if (!hist_is_double && T > 24) {
    dim3 grid((T + 63) / 64, (G + 15) / 16, J);
    hipLaunchKernelGGL(ski_toeplitz_wide_kernel, grid, dim3(256), (size_t)G * sizeof(float), st,
                       reinterpret_cast<const float *>(hist), gp, H, G, T);
    return launch_status();
  }
  const int G16 = (G + 15) & ~15;
  // few projections (C5: J = 3 -> 192 tiles): 16 waves per tile split the grid points, so that the launch is not 192
  // workgroups each walking a 16-step dependent loop; many projections: 4 waves per tile (the matrix pipe is the limit)
  const bool wide_wg = (size_t)J * ((G + 15) / 16) <= 640 && G16 >= 256;
  const int nw = wide_wg ? 16 : 4;
  const size_t lds = ((size_t)G16 + (size_t)(nw - 1) * 256) * sizeof(double);
  if (hist_is_double && T <= 64 && lds <= 64 * 1024) {        // matrix-core path, float64 (16 columns per workgroup)
    dim3 grid((G + 15) / 16, J, (T + 15) / 16);
    if (wide_wg)
      hipLaunchKernelGGL(ski_toeplitz_mfma_kernel<16>, grid, dim3(1024), lds, st, reinterpret_cast<const double *>(hist), gp,
                         H, G, T, tcol);
    else
      hipLaunchKernelGGL(ski_toeplitz_mfma_kernel<4>, grid, dim3(256), lds, st, reinterpret_cast<const double *>(hist), gp, H,
                         G, T, tcol);
    return launch_status();
  }
  int Tp = 1;
  while (Tp < T && Tp < 16) Tp <<= 1;
  const int rows_per_block = 256 / Tp;
  dim3 grid((G + rows_per_block - 1) / rows_per_block, J);
  if (hist_is_double)
    hipLaunchKernelGGL((ski_toeplitz_kernel<double>), grid, dim3(256), (size_t)G * sizeof(double), st,
                       reinterpret_cast<const double *>(hist), gp, H, G, T);
  else
    hipLaunchKernelGGL((ski_toeplitz_kernel<float>), grid, dim3(256), (size_t)G * sizeof(double), st,
                       reinterpret_cast<const float *>(hist), gp, H, G, T);
  return launch_status();
}

int ski_gather_all(const float *Z, const float *gp, const float *H, const float *V, float *out, long long M, int ldz,
                   int J, int G, int T, float scale, float noise, hipStream_t st) {
  if (T > 12) {
    long long nblk = (M + 255) / 256;
    if (nblk > 2048) nblk = 2048;
    const long long ppb = (M + nblk - 1) / nblk;
    hipLaunchKernelGGL(ski_gather_wide_kernel, dim3((unsigned)nblk, (unsigned)((T + 63) / 64)), dim3(256), 0, st, Z, gp,
                       H, V, out, M, ldz, J, G, T, scale, noise, ppb);
    return launch_status();
  }
  for (int t0 = 0; t0 < T;) {
    const int tt = ski_tpiece(T - t0);
    const int tcnt = (T - t0 < tt) ? T - t0 : tt;
    const int lpp = tt == 1 ? 1 : (tt == 4 ? 4 : 16);
    const unsigned nb = (unsigned)((M * lpp + 255) / 256);
    if (tt == 1)
      hipLaunchKernelGGL((ski_gather_kernel<1>), dim3(nb), dim3(256), 0, st, Z, gp, H, V, out, M, ldz, J, G, T, t0, tcnt, scale, noise);
    else if (tt == 4)
      hipLaunchKernelGGL((ski_gather_kernel<4>), dim3(nb), dim3(256), 0, st, Z, gp, H, V, out, M, ldz, J, G, T, t0, tcnt, scale, noise);
    else
      hipLaunchKernelGGL((ski_gather_kernel<12>), dim3(nb), dim3(256), 0, st, Z, gp, H, V, out, M, ldz, J, G, T, t0, tcnt, scale, noise);
    int rc = launch_status();
    if (rc) return rc;
    t0 += tcnt;
  }
  return 0;
}
}  // namespace

namespace rpgp_internal {
int ski_toeplitz_launch(const void *hist, int hist_is_double, const float *gp, float *H, int J, int G, int T,
                        hipStream_t st, const double *tcol) {
  return ski_toeplitz(hist, hist_is_double, gp, H, J, G, T, st, tcol);
}
int ski_gather_launch(const float *Z, const float *gp, const float *H, const float *V, float *out, long long M, int ldz,
                      int J, int G, int T, float scale, float noise, hipStream_t st) {
  return ski_gather_all(Z, gp, H, V, out, M, ldz, J, G, T, scale, noise, st);
}
size_t ski_scratch_floats(int J, int G) { return ski_slab_floats(J, G); }
size_t ski_scratch_offset_floats(int J, int G, int T) { return 3 * (size_t)J * G * (2 * T) + 2 * kSkiMaxParts; }
}  // namespace rpgp_internal

extern "C" {

size_t rpgp_ski_workspace_bytes(int J, int G, int T) {
  if (J <= 0 || G <= 0 || T <= 0) return 0;
  // hist + H for up to 2T columns (the derivative uses [L | R]) + min/max partials + per-chunk scatter slabs
  // (the histogram region is sized for float64 entries)
  return (3 * (size_t)J * G * (2 * T) + 2 * kSkiMaxParts + ski_slab_floats(J, G)) * sizeof(float);
}

int rpgp_ski_grid(const float *Z1, int64_t N1, int ld1, const float *Z2, int64_t N2, int ld2, int J, int G,
                  float *grid_params, void *workspace, size_t workspace_bytes, void *stream) {
  if (!Z1 || N1 <= 0 || J <= 0 || G < 8 || !grid_params || ld1 < J || (Z2 && (N2 <= 0 || ld2 < J))) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < 2 * kSkiMaxParts * sizeof(float)) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  float *part = reinterpret_cast<float *>(workspace);
  const long long n2 = Z2 ? N2 : 0;
  const long long total = (N1 + n2) * J;
  int nblk = (int)((total + 4095) / 4096);
  if (nblk > kSkiMaxParts) nblk = kSkiMaxParts;
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(ski_minmax_kernel, dim3(nblk), dim3(256), 0, st, Z1, (long long)N1, ld1, Z2 ? Z2 : Z1, n2,
                     Z2 ? ld2 : ld1, J, part);
  hipLaunchKernelGGL(ski_grid_finish_kernel, dim3(1), dim3(64), 0, st, part, nblk, G, grid_params);
  return launch_status();
}

int rpgp_ski_grid_per_projection(const float *Z1, int64_t N1, int ld1, const float *Z2, int64_t N2, int ld2, int J, int G,
                                 float *grid_params, void *workspace, size_t workspace_bytes, void *stream) {
  if (!Z1 || N1 <= 0 || J <= 0 || G < 8 || !grid_params || ld1 < J || (Z2 && (N2 <= 0 || ld2 < J))) return RPGP_EINVAL;
  const long long n2 = Z2 ? N2 : 0;
  int nblk = (int)((N1 + n2 + 4095) / 4096);
  const int cap = kSkiMaxParts / (J > 0 ? J : 1) > 0 ? kSkiMaxParts / J : 1;
  if (nblk > cap) nblk = cap;
  if (nblk < 1) nblk = 1;
  if (!workspace || workspace_bytes < 2 * (size_t)kSkiMaxParts * sizeof(float) || (size_t)nblk * J > (size_t)kSkiMaxParts)
    return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  float *part = reinterpret_cast<float *>(workspace);
  hipLaunchKernelGGL(ski_minmax_proj_kernel, dim3((unsigned)nblk, (unsigned)J), dim3(256), 0, st, Z1, (long long)N1, ld1,
                     Z2 ? Z2 : Z1, n2, Z2 ? ld2 : ld1, J, part);
  hipLaunchKernelGGL(ski_grid_finish_proj_kernel, dim3((unsigned)J), dim3(64), 0, st, part, nblk, J, G,
                     grid_params);
  return launch_status();
}

int rpgp_ski_mvm(const float *Z1, const float *Z2, const float *grid_params, const float *V, float *out, int64_t M,
                 int64_t N, int ldz1, int ldz2, int J, int G, int T, float scale, float noise, void *workspace,
                 size_t workspace_bytes, void *stream) {
  if (!Z1 || !Z2 || !grid_params || !V || !out || M <= 0 || N <= 0 || J <= 0 || G < 8 || T <= 0 || ldz1 < J ||
      ldz2 < J)
    return RPGP_EINVAL;
  if ((size_t)G * 13 * sizeof(float) > 64 * 1024) return RPGP_EINVAL;   // LDS histogram + cell counts: G <= 1260
  if (noise != 0.f && (M != N)) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  float *hist = reinterpret_cast<float *>(workspace);
  float *H = hist + 2 * (size_t)J * G * (2 * T);
  float *slab = H + (size_t)J * G * (2 * T) + 2 * kSkiMaxParts;
  int hist_is_double = 0;
  int rc = ski_scatter_all(Z2, grid_params, V, hist, slab, N, ldz2, J, G, T, st, &hist_is_double);
  if (rc) return rc;
  rc = ski_toeplitz(hist, hist_is_double, grid_params, H, J, G, T, st);
  if (rc) return rc;
  return ski_gather_all(Z1, grid_params, H, V, out, M, ldz1, J, G, T, scale, noise, st);
}

// ---- the three stages of rpgp_ski_mvm as separate calls (row-sharded multi-GPU SKI: the J x G x T float64 grid
// histogram is all-reduced across ranks between the scatter and the Toeplitz product) --------------------------------
int rpgp_ski_scatter(const float *Z, const float *grid_params, const float *V, double *hist, int64_t N, int ldz, int J,
                     int G, int T, void *workspace, size_t workspace_bytes, void *stream) {
  if (!Z || !grid_params || !V || !hist || N <= 0 || J <= 0 || G < 8 || T <= 0 || T > 12 || ldz < J) return RPGP_EINVAL;
  if ((size_t)G * 13 * sizeof(float) > 64 * 1024) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  float *slab = reinterpret_cast<float *>(workspace) + 3 * (size_t)J * G * (2 * T) + 2 * kSkiMaxParts;
  return ski_scatter_narrow(Z, grid_params, V, hist, slab, (long long)N, ldz, J, G, T, T, 0, as_stream(stream));
}

int rpgp_ski_grid_product(const double *hist, const float *grid_params, float *H, int J, int G, int T, void *stream) {
  if (!hist || !grid_params || !H || J <= 0 || G < 8 || T <= 0 || T > 12) return RPGP_EINVAL;
  return ski_toeplitz(hist, 1, grid_params, H, J, G, T, as_stream(stream));
}

int rpgp_ski_gather(const float *Z, const float *grid_params, const float *H, const float *V, float *out, int64_t M,
                    int ldz, int J, int G, int T, float scale, float noise, void *stream) {
  if (!Z || !grid_params || !H || !out || M <= 0 || J <= 0 || G < 8 || T <= 0 || T > 12 || ldz < J) return RPGP_EINVAL;
  if (noise != 0.f && !V) return RPGP_EINVAL;
  return ski_gather_all(Z, grid_params, H, V, out, (long long)M, ldz, J, G, T, scale, noise, as_stream(stream));
}

int rpgp_ski_dense(const float *Z1, const float *Z2, const float *grid_params, float *out, int64_t M, int64_t N,
                   int ldz1, int ldz2, int64_t ldo, int J, int G, float scale, void *stream) {
  if (!Z1 || !Z2 || !grid_params || !out || M <= 0 || N <= 0 || J <= 0 || J > 64 || G < 8 || ldz1 < J || ldz2 < J ||
      ldo < N || M > 0x7fffffffLL || N > 0x7fffffffLL)
    return RPGP_EINVAL;
  const size_t lds = ((size_t)G + 16 * J * 5) * sizeof(float);
  if (lds > 64 * 1024) return RPGP_EINVAL;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)((M + 15) / 16));
  hipLaunchKernelGGL(ski_dense_kernel, grid, dim3(256), lds, as_stream(stream), Z1, Z2, grid_params, out, (int)M, (int)N,
                     ldz1, ldz2, (long long)ldo, J, G, scale);
  return launch_status();
}

int rpgp_ski_diag(const float *Z, const float *grid_params, float *diag, int64_t N, int ldz, int J, int G, float scale,
                  void *stream) {
  if (!Z || !grid_params || !diag || N <= 0 || J <= 0 || G < 8 || ldz < J) return RPGP_EINVAL;
  hipLaunchKernelGGL(ski_diag_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), Z,
                     grid_params, diag, (long long)N, ldz, J, G, scale);
  return launch_status();
}

// The derivative in two stages (the row-sharded operator all-reduces the histogram in between):
//   scatter: hist2[j][g][0..T) = W_j^T L, hist2[j][g][T..2T) = W_j^T R  over the given rows (float64)
//   finish : H = Tm hist2 (all 2T columns), per-row gather of the stencil derivatives -> gZ, row sums -> gscale (, gcomp)
static int ski_bilinear_scatter_stage(const float *Z, const float *grid_params, const float *L, const float *R,
                                      double *hist, int64_t N, int ldz, int J, int G, int T, void *workspace,
                                      size_t workspace_bytes, void *stream) {
  if (!Z || !grid_params || !L || !R || !hist || N <= 0 || J <= 0 || G < 8 || T <= 0 || T > 12 || ldz < J)
    return RPGP_EINVAL;
  if ((size_t)G * 13 * sizeof(float) > 64 * 1024) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const int T2 = 2 * T;
  float *slab = reinterpret_cast<float *>(workspace) + 3 * (size_t)J * G * T2 + 2 * kSkiMaxParts;
  // two passes writing into column offsets 0 and T of a [J][G][2T] float64 histogram
  for (int half = 0; half < 2; ++half) {
    const int rcs = ski_scatter_narrow(Z, grid_params, half == 0 ? L : R, hist, slab, (long long)N, ldz, J, G, T, T2,
                                       half * T, st);
    if (rcs) return rcs;
  }
  return 0;
}

static int ski_bilinear_finish_stage(const float *Z, const float *grid_params, const double *hist, const float *L,
                                     const float *R, float *gZ, float *gscale, float *gcomp, int64_t N, int ldz, int ldg,
                                     int J, int G, int T, float scale, void *workspace, size_t workspace_bytes,
                                     float *row_scratch, void *stream) {
  if (!Z || !grid_params || !hist || !L || !R || !gZ || !gscale || !row_scratch || N <= 0 || J <= 0 || G < 8 || T <= 0 ||
      T > 12 || ldz < J || ldg < J)
    return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const int T2 = 2 * T;
  float *H = reinterpret_cast<float *>(workspace) + 2 * (size_t)J * G * T2;
  int rc = ski_toeplitz(const_cast<double *>(hist), 1, grid_params, H, J, G, T2, st);
  if (rc) return rc;
  const unsigned nb = (unsigned)((N + 255) / 256);
  float *rowC = gcomp ? row_scratch + N : nullptr;
  if (T <= 1)
    hipLaunchKernelGGL((ski_grad_gather_kernel<1>), dim3(nb), dim3(256), 0, st, Z, grid_params, H, L, R, gZ, row_scratch, (long long)N, ldz, ldg, J, G, T, scale, rowC);
  else if (T <= 4)
    hipLaunchKernelGGL((ski_grad_gather_kernel<4>), dim3(nb), dim3(256), 0, st, Z, grid_params, H, L, R, gZ, row_scratch, (long long)N, ldz, ldg, J, G, T, scale, rowC);
  else
    hipLaunchKernelGGL((ski_grad_gather_kernel<12>), dim3(nb), dim3(256), 0, st, Z, grid_params, H, L, R, gZ, row_scratch, (long long)N, ldz, ldg, J, G, T, scale, rowC);
  rc = launch_status();
  if (rc) return rc;
  int src = rpgp_internal::sum_vector_launch(row_scratch, gscale, (int)nb, 1.0f, st);      // (one partial per workgroup)
  if (src) return src;
  if (gcomp) {
    src = rpgp_internal::sum_columns_launch(rowC, gcomp, (int)N, J, 1.0f, st);
    if (src) return src;
  }
  return launch_status();
}

// gcomp == nullptr: plain form.  Otherwise gcomp[j] (J floats) = sum_i of the per-projection parts of gscale (they
// carry the projection's weight: divide by w_j for the unweighted component sums) and row_scratch holds N * (J + 1) floats.
static int ski_bilinear_common(const float *Z, const float *grid_params, const float *L, const float *R, float *gZ,
                               float *gscale, float *gcomp, int64_t N, int ldz, int ldg, int J, int G, int T,
                               float scale, void *workspace, size_t workspace_bytes, float *row_scratch, void *stream) {
  if (!gZ || !gscale || !row_scratch || ldg < J) return RPGP_EINVAL;
  double *hist = reinterpret_cast<double *>(workspace);
  int rc = ski_bilinear_scatter_stage(Z, grid_params, L, R, hist, N, ldz, J, G, T, workspace, workspace_bytes, stream);
  if (rc) return rc;
  return ski_bilinear_finish_stage(Z, grid_params, hist, L, R, gZ, gscale, gcomp, N, ldz, ldg, J, G, T, scale, workspace,
                                   workspace_bytes, row_scratch, stream);
}

int rpgp_ski_bilinear_scatter(const float *Z, const float *grid_params, const float *L, const float *R, double *hist2,
                              int64_t N, int ldz, int J, int G, int T, void *workspace, size_t workspace_bytes,
                              void *stream) {
  return ski_bilinear_scatter_stage(Z, grid_params, L, R, hist2, N, ldz, J, G, T, workspace, workspace_bytes, stream);
}

int rpgp_ski_bilinear_finish(const float *Z, const float *grid_params, const double *hist2, const float *L, const float *R,
                             float *gZ, float *gscale, float *gcomp, int64_t N, int ldz, int ldg, int J, int G, int T,
                             float scale, void *workspace, size_t workspace_bytes, float *row_scratch, void *stream) {
  return ski_bilinear_finish_stage(Z, grid_params, hist2, L, R, gZ, gscale, gcomp, N, ldz, ldg, J, G, T, scale, workspace,
                                   workspace_bytes, row_scratch, stream);
}

int rpgp_ski_bilinear_grad(const float *Z, const float *grid_params, const float *L, const float *R, float *gZ,
                           float *gscale, int64_t N, int ldz, int ldg, int J, int G, int T, float scale,
                           void *workspace, size_t workspace_bytes, float *row_scratch, void *stream) {
  return ski_bilinear_common(Z, grid_params, L, R, gZ, gscale, nullptr, N, ldz, ldg, J, G, T, scale, workspace,
                             workspace_bytes, row_scratch, stream);
}

int rpgp_ski_bilinear_grad_comp(const float *Z, const float *grid_params, const float *L, const float *R, float *gZ,
                                float *gscale, float *gcomp, int64_t N, int ldz, int ldg, int J, int G, int T,
                                float scale, void *workspace, size_t workspace_bytes, float *row_scratch,
                                void *stream) {
  if (!gcomp) return RPGP_EINVAL;
  return ski_bilinear_common(Z, grid_params, L, R, gZ, gscale, gcomp, N, ldz, ldg, J, G, T, scale, workspace,
                             workspace_bytes, row_scratch, stream);
}

}  // extern "C"
