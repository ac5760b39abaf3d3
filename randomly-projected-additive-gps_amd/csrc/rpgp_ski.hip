// rpgp_ski.hip — the planned (cell-sorted) SKI product.
//
// Z is fixed for a whole CG solve (one hyper-parameter step = 20 .. 100 products), so the interpolation structure is
// prepared ONCE per step (`rpgp_ski_plan`): every projection's points are sorted by the grid cell of their first tap
// (stable radix sort: the order inside a cell is the point order, so everything downstream is deterministic), with the
// tap fraction kept in sorted order.  The scatter  hist_j = W_j^T V  is then a SEGMENTED REDUCTION over the sorted
// points — no atomics at all (round 2: fixed-point LDS atomics, 58 - 65 us for 22 MB at the C5 shape = 0.36 TB/s):
//   ski_scatter_cell   : one workgroup per grid cell (dispatched centre-out): the cell's point indices and fractions arrive
//                        in coalesced loads, the V rows are gathered with all loads of a round in flight, every lane
//                        accumulates its 4 tap sums in registers, lane groups are combined by shuffles;
//                        cellpart[cell][tap][t]  (fp32)
//   ski_cellsum4       : hist[j][g][t] = sum over the taps k of cell g - k, fixed order, FLOAT64
//   Toeplitz product   : ski_toeplitz_mfma_kernel (rpgp_kernels.hip), first column from the plan (no exp per product)
//   gather             : with the whole H (J x G x T floats) resident in LDS when it fits (C5: 135 KB of the 160 KB):
//                        the 12 tap rows per point are LDS reads instead of L2 requests that miss L1 four lines at a time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#include <cstring>
#include <stdlib.h>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/rpgp.h"
#include "rpgp_internal.h"
#include "rpgp_ski_common.h"

namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int launch_status() { return (int)hipGetLastError(); }

// ---- plan layout -------------------------------------------------------------------------------------------------------
struct PlanView {
  int2 *rec;           // [N * J]  (point index, tap fraction as bits) of the s-th entry in (projection, cell, point) order:
                       //          ONE 8-byte record per entry (rounds 3 - 4: a 4-byte index and a 16-byte weight record,
                       //          20 of the 64 bytes a scattered row cost); the four weights are re-derived from the fraction
                       //          by the arithmetic that produced them (ski_weights_of_frac: same bits)
  float *fnat;         // [N][J]   tap fraction in point order (plan construction only)
  int *cell_start;     // [J * G + 1]  first entry of cell (j, c);  cell = index of the first tap, 0 .. G - 4
  double *tcol;        // [J][G16]  first column of the Toeplitz matrix of every projection, exp(-0.5 (k h_j)^2)
  size_t bytes;
};

inline PlanView plan_view(void *base, long long N, int J, int G) {
  PlanView v;
  char *p = reinterpret_cast<char *>(base);
  const size_t nj = (size_t)N * J, cells = (size_t)J * G + 1, G16 = (size_t)((G + 15) & ~15);
  v.rec = reinterpret_cast<int2 *>(p); p += align256(nj * sizeof(int2));
  v.fnat = reinterpret_cast<float *>(p); p += align256(nj * sizeof(float));
  v.cell_start = reinterpret_cast<int *>(p); p += align256(cells * sizeof(int));
  v.tcol = reinterpret_cast<double *>(p); p += align256((size_t)J * G16 * sizeof(double));
  v.bytes = (size_t)(p - reinterpret_cast<char *>(base));
  return v;
}

// ---- plan construction -------------------------------------------------------------------------------------------------
// entry e = j * N + i:  key = j * G + (first-tap cell of z_ij), value = e; the 4 tap weights and the first tap index of
// every (point, projection) in point order
__global__ __launch_bounds__(256) void plan_keys_kernel(const float *__restrict__ Z, const float *__restrict__ gp,
                                                        long long N, int ldz, int J, int G, unsigned *__restrict__ keys,
                                                        unsigned *__restrict__ vals, float *__restrict__ fnat) {
  const long long total = N * J;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e / N);
    const long long i = e - (long long)j * N;
    const float *gj = ski_grid_of(gp, J, j);
    const float u = ski_grid_coord(Z[i * ldz + j], gj[0], gj[2], G);
    float w[4], dw[4];
    const int idx0 = ski_taps_u<false>(u, gj[2], G, w, dw);
    keys[e] = (unsigned)(j * G + idx0);
    vals[e] = (unsigned)e;
    fnat[i * J + j] = u - __builtin_floorf(u);               // (the fraction ski_taps_u derives the weights from)
  }
}

// cell_start[c] = first sorted entry with key >= c (c = 0 .. J * G); Toeplitz first column
__global__ __launch_bounds__(256) void plan_starts_kernel(const unsigned *__restrict__ keys_sorted, long long total, int cells,
                                                          int *__restrict__ cell_start, const float *__restrict__ gp, int G,
                                                          double *__restrict__ tcol) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c <= cells) {
    long long lo = 0, hi = total;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (keys_sorted[mid] < (unsigned)c) lo = mid + 1;
      else hi = mid;
    }
    cell_start[c] = (int)lo;
  }
  const int G16 = (G + 15) & ~15;
  const int J = cells / G;
  if (c < J * G16) {                         // (with a shared grid only row 0 is read)
    const int j = c / G16, k = c % G16;
    const double d = (double)k * (double)ski_grid_of(gp, J, j)[1];
    tcol[c] = k < G ? ski_radial_f64(ski_kind(gp), d) : 0.0;
  }
}

// perm / tap weights in sorted order
__global__ __launch_bounds__(256) void plan_finish_kernel(const unsigned *__restrict__ keys_sorted,
                                                          const unsigned *__restrict__ vals_sorted,
                                                          const float *__restrict__ fnat, long long N, int J, int G,
                                                          long long total, int2 *__restrict__ rec) {
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < total; s += (long long)gridDim.x * 256) {
    const unsigned e = vals_sorted[s];
    const int j = (int)(keys_sorted[s] / (unsigned)G);
    const long long i = (long long)e - (long long)j * N;
    rec[s] = make_int2((int)i, __builtin_bit_cast(int, fnat[i * J + j]));
  }
}

template <int LPP, int CPL>
__global__ __launch_bounds__(256) void ski_scatter_cell_kernel(const int2 *__restrict__ rec,
                                                               const int *__restrict__ cell_start, const float *__restrict__ V,
                                                               float *__restrict__ cellpart, int T, int t0, int tcnt,
                                                               int Gorder) {
  constexpr int TT = LPP * CPL;
  constexpr int PPW = 64 / LPP;
  constexpr int STEPS = 64 / PPW;
  __shared__ float sP[4][4][12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Dispatch order (round 5): a cell's workgroup walks ALL its points, so its lifetime follows the cell's count, and with
  // ~1.5 rounds of resident workgroups the cells that start LAST decide when the kernel ends.  Projected coordinates pile up
  // around the middle of the grid: the workgroups take the cells centre-out (middle cell first, alternating sides, the
  // projections interleaved), so the long cells start first and the short ones fill the tail.  Which workgroup computes a cell
  // does not touch what it computes: same bits.  Gorder = 0: storage order.
  int cell = blockIdx.x;
  if (Gorder > 0) {
    const int J = (int)gridDim.x / Gorder, j = (int)blockIdx.x % J, r = (int)blockIdx.x / J;
    const int g = Gorder / 2 + ((r & 1) ? -((r + 1) >> 1) : (r >> 1));
    cell = j * Gorder + g;
  }
  const int sb = cell_start[cell], se = cell_start[cell + 1];
  const int c = lane % LPP, pg = lane / LPP;
  float acc[4][CPL];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < CPL; ++r) acc[k][r] = 0.f;
  int base = sb + wave * 64;
  int2 pin[STEPS];
  if (base < se) {
    const int cnt = se - base < 64 ? se - base : 64;
#pragma unroll
    for (int m = 0; m < STEPS; ++m) {
      const int src = m * PPW + pg;
      pin[m] = rec[base + (src < cnt ? src : 0)];
    }
  }
  for (; base < se; base += 256) {
    const int cnt = se - base < 64 ? se - base : 64;
    int2 pi[STEPS];
#pragma unroll
    for (int m = 0; m < STEPS; ++m) pi[m] = pin[m];
    const int nbase = base + 256;
    if (nbase < se) {                          // next round's records: in flight while this round's rows arrive
      const int ncnt = se - nbase < 64 ? se - nbase : 64;
#pragma unroll
      for (int m = 0; m < STEPS; ++m) {
        const int src = m * PPW + pg;
        pin[m] = rec[nbase + (src < ncnt ? src : 0)];
      }
    }
    float v[STEPS][CPL];
    float4 w[STEPS];
#pragma unroll
    for (int m = 0; m < STEPS; ++m) {
      const int src = m * PPW + pg;
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        const int col = c + LPP * r;
        const bool ok = src < cnt && col < tcnt;
        const float x = V[ok ? (size_t)pi[m].x * T + t0 + col : 0];
        v[m][r] = ok ? x : 0.f;
      }
      w[m] = ski_weights_of_frac(__builtin_bit_cast(float, pi[m].y));      // (on the vector units, under the rows' latency)
    }
#pragma unroll
    for (int m = 0; m < STEPS; ++m) {
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        acc[0][r] = __builtin_fmaf(w[m].x, v[m][r], acc[0][r]);
        acc[1][r] = __builtin_fmaf(w[m].y, v[m][r], acc[1][r]);
        acc[2][r] = __builtin_fmaf(w[m].z, v[m][r], acc[2][r]);
        acc[3][r] = __builtin_fmaf(w[m].w, v[m][r], acc[3][r]);
      }
    }
  }
#pragma unroll
  for (int off = LPP; off < 64; off <<= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int r = 0; r < CPL; ++r) acc[k][r] += __shfl_xor(acc[k][r], off, 64);
  }
  if (lane < LPP) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int r = 0; r < CPL; ++r) sP[wave][k][c + LPP * r] = acc[k][r];
  }
  __syncthreads();
  if (threadIdx.x < 4 * TT) {
    const int k = threadIdx.x / TT, tt = threadIdx.x % TT;
    if (tt < tcnt) cellpart[((size_t)cell * 4 + k) * TT + tt] = ((sP[0][k][tt] + sP[1][k][tt]) + sP[2][k][tt]) + sP[3][k][tt];
  }
}

// hist[j][g][hoff + t] (row stride HT, float64) = sum_k cellpart[cell (j, g - k)][k][t]   (cells 0 .. G - 4)
__global__ __launch_bounds__(256) void ski_cellsum4_kernel(const float *__restrict__ cellpart, double *__restrict__ hist, int J,
                                                           int G, int TT, int tcnt, int HT, int hoff) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)J * G * TT) return;
  const int t = (int)(e % TT);
  const long long jg = e / TT;
  const int g = (int)(jg % G), j = (int)(jg / G);
  if (t >= tcnt) return;
  float x[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cc = g - k;
    const bool ok = !(cc < 0 || cc > G - 4);
    const float y = cellpart[(((size_t)j * G + (ok ? cc : 0)) * 4 + k) * TT + t];
    x[k] = ok ? y : 0.f;
  }
  hist[jg * HT + hoff + t] = (((double)x[0] + (double)x[1]) + (double)x[2]) + (double)x[3];
}

// ---- gather with H resident in LDS -----------------------------------------------------------------------------------------
// out[i][t] = scale * sum_j sum_k w_k(z_ij) H[j][idx0 + k][t] + noise V[i][t];  one persistent 1024-thread workgroup per CU
// holds all J x G x T floats of H in LDS (dynamic, up to 160 KB).  After the load the 16 waves are independent: a wave owns
// 4 points per step (16 lanes per point: lane t < J computes the stencil of projection t, the group shares it by
// shuffles — no barrier), U steps in flight.
// 4 lanes per point, CPL columns per lane (T <= 4 CPL): 16 points per wave step.  Lane c of a point derives the stencil of
// projection j0 + c; the quad shares it with DPP quad_perm broadcasts (VALU, not the LDS pipe, which the tap reads need).
template <int Q>
__device__ __forceinline__ float quad_bcast(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), Q * 0x55, 0xf, 0xf, true));
}
template <int Q>
__device__ __forceinline__ int quad_bcast_i(int x) {
  return __builtin_amdgcn_update_dpp(0, x, Q * 0x55, 0xf, 0xf, true);
}

template <int CPL, int U>
__global__ __launch_bounds__(1024) void ski_gather_lds_kernel(const float *__restrict__ Z, const float *__restrict__ gp,
                                                              const float *__restrict__ H, const float *__restrict__ V,
                                                              float *__restrict__ out, long long M, int ldz, int J, int G, int T,
                                                              float scale, float noise) {
  extern __shared__ float sH[];              // [J][G][T]
  const int nH = J * G * T;
  // float4 granules, 16 independent 16-byte loads per thread in flight (the launcher takes this kernel only when
  // nH = J G T is a multiple of 4: odd G or T with an odd J would read past H and write past the LDS allocation)
  for (int e0 = threadIdx.x * 4; e0 < nH; e0 += 16 * 4096) {
    float4 q[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = e0 + u * 4096;
      q[u] = e < nH ? *reinterpret_cast<const float4 *>(H + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = e0 + u * 4096;
      if (e < nH) *reinterpret_cast<float4 *>(sH + e) = q[u];
    }
  }
  const int lane = threadIdx.x & 63;
  const int c = lane & 3, pg = lane >> 2;                     // 16 points per wave step, 4 lanes per point
  const long long gw = (long long)blockIdx.x * 16 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 16;
  const long long groups = (M + 15) / 16;                     // point groups of 16
  int colc[CPL];
#pragma unroll
  for (int r = 0; r < CPL; ++r) colc[r] = (c + 4 * r < T) ? c + 4 * r : 0;
  __syncthreads();
  for (long long q0 = gw * U; q0 < groups; q0 += nw * U) {
    float zc[U], vin[U][CPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {                             // all coordinate / right-hand-side loads of U groups first
      const long long p = (q0 + u) * 16 + pg;
      const bool live = q0 + u < groups && p < M;
      zc[u] = (live && c < J) ? Z[p * ldz + c] : 0.f;
#pragma unroll
      for (int r = 0; r < CPL; ++r) vin[u][r] = (live && c + 4 * r < T && noise != 0.f) ? V[p * T + c + 4 * r] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long p = (q0 + u) * 16 + pg;
      const bool live = q0 + u < groups && p < M;
      double acc[CPL];
#pragma unroll
      for (int r = 0; r < CPL; ++r) acc[r] = 0.0;
      for (int j0 = 0; j0 < J; j0 += 4) {
        float zz = zc[u];
        if (j0 > 0) zz = (live && j0 + c < J) ? Z[p * ldz + j0 + c] : 0.f;
        float w[4], dw[4];
        const float *gj = ski_grid_of(gp, J, j0 + c < J ? j0 + c : 0);
        const int idx_mine = ski_taps<false>(zz, gj[0], gj[2], G, w, dw);
        auto one = [&](int jq, int idx0, float w0, float w1, float w2, float w3) {
          if (jq < J) {
            const float *hp = sH + ((size_t)jq * G + idx0) * T;
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
              const float *hq = hp + colc[r];
              float pr = w0 * hq[0];
              pr = __builtin_fmaf(w1, hq[T], pr);
              pr = __builtin_fmaf(w2, hq[2 * T], pr);
              pr = __builtin_fmaf(w3, hq[3 * T], pr);
              acc[r] += (double)pr;
            }
          }
        };
        one(j0 + 0, quad_bcast_i<0>(idx_mine), quad_bcast<0>(w[0]), quad_bcast<0>(w[1]), quad_bcast<0>(w[2]), quad_bcast<0>(w[3]));
        one(j0 + 1, quad_bcast_i<1>(idx_mine), quad_bcast<1>(w[0]), quad_bcast<1>(w[1]), quad_bcast<1>(w[2]), quad_bcast<1>(w[3]));
        one(j0 + 2, quad_bcast_i<2>(idx_mine), quad_bcast<2>(w[0]), quad_bcast<2>(w[1]), quad_bcast<2>(w[2]), quad_bcast<2>(w[3]));
        one(j0 + 3, quad_bcast_i<3>(idx_mine), quad_bcast<3>(w[0]), quad_bcast<3>(w[1]), quad_bcast<3>(w[2]), quad_bcast<3>(w[3]));
      }
      if (live) {
#pragma unroll
        for (int r = 0; r < CPL; ++r)
          if (c + 4 * r < T) out[p * T + c + 4 * r] = __builtin_fmaf(noise, vin[u][r], scale * (float)acc[r]);
      }
    }
  }
}

constexpr size_t kGatherLdsMax = 150 * 1024;

// out = scale * W H + noise V: the gather with H resident in LDS when the shape allows it (the SKI product of a large point
// set: J G T floats within the LDS of a CU, a multiple of 4, >= 32 768 rows), else the general gather of rpgp_ski_base.hip
int gather_planned(const float *Z, const float *gp, const float *H, const float *V, float *out, long long M, int ldz, int J,
                   int G, int T, float scale, float noise, hipStream_t st) {
  const size_t lds = (size_t)J * G * T * sizeof(float);
  if (T > 1 && T <= 12 && lds <= kGatherLdsMax && M >= 32768 && ((size_t)J * G * T) % 4 == 0) {
    // (the attribute is per device: remembered per device ordinal — ADVICE r5)
    static bool attr_set_dev[64] = {};
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    bool &attr_set = attr_set_dev[dev];
    if (!attr_set) {
      bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(ski_gather_lds_kernel<1, 4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGatherLdsMax) == hipSuccess;
      ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(ski_gather_lds_kernel<2, 4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGatherLdsMax) == hipSuccess;
      ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(ski_gather_lds_kernel<3, 3>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGatherLdsMax) == hipSuccess;
      if (!ok) return rpgp_internal::ski_gather_launch(Z, gp, H, V, out, M, ldz, J, G, T, scale, noise, st);
      attr_set = true;
    }
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (T <= 4)
      hipLaunchKernelGGL((ski_gather_lds_kernel<1, 4>), dim3((unsigned)cus), dim3(1024), lds, st, Z, gp, H, V, out, M, ldz, J, G, T, scale, noise);
    else if (T <= 8)
      hipLaunchKernelGGL((ski_gather_lds_kernel<2, 4>), dim3((unsigned)cus), dim3(1024), lds, st, Z, gp, H, V, out, M, ldz, J, G, T, scale, noise);
    else
      hipLaunchKernelGGL((ski_gather_lds_kernel<3, 3>), dim3((unsigned)cus), dim3(1024), lds, st, Z, gp, H, V, out, M, ldz, J, G, T, scale, noise);
    return launch_status();
  }
  return rpgp_internal::ski_gather_launch(Z, gp, H, V, out, M, ldz, J, G, T, scale, noise, st);
}

inline int ski_tpiece(int remaining) { return remaining > 4 ? 12 : (remaining > 1 ? 4 : 1); }

// hist[j][g][hoff .. hoff + T) (row stride HT, float64) from the plan; `partial` holds J * G * 4 * 12 floats.
// One workgroup per cell, dispatched centre-out (the busiest cells of bell-shaped coordinates first: round 5), then the four
// shifted reads of ski_cellsum4_kernel.
int scatter_planned(const PlanView &pv, const float *V, double *hist, float *partial, long long N, int J, int G, int T, int HT,
                    int hoff, hipStream_t st) {
  const int cells = J * G;
  for (int t0 = 0; t0 < T;) {
    const int tt = ski_tpiece(T - t0);
    const int tcnt = (T - t0 < tt) ? T - t0 : tt;
    const long long n = (long long)J * G * tt;
    if (tt == 1)
      hipLaunchKernelGGL((ski_scatter_cell_kernel<1, 1>), dim3((unsigned)cells), dim3(256), 0, st, pv.rec, pv.cell_start, V, partial, T, t0, tcnt, G);
    else if (tt == 4)
      hipLaunchKernelGGL((ski_scatter_cell_kernel<4, 1>), dim3((unsigned)cells), dim3(256), 0, st, pv.rec, pv.cell_start, V, partial, T, t0, tcnt, G);
    else
      hipLaunchKernelGGL((ski_scatter_cell_kernel<4, 3>), dim3((unsigned)cells), dim3(256), 0, st, pv.rec, pv.cell_start, V, partial, T, t0, tcnt, G);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(ski_cellsum4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, hist, J, G, tt, tcnt,
                       HT, hoff + t0);
    rc = launch_status();
    if (rc) return rc;
    t0 += tcnt;
  }
  return 0;
}

// floats of per-cell tap records one scatter pass writes (<= 12 columns per pass) against the scratch of the SKI workspace
inline bool scratch_fits(int J, int G) { return (size_t)J * G * 48 <= rpgp_internal::ski_scratch_floats(J, G); }

inline bool plan_args_ok(long long N, int J, int G) {
  return N > 0 && J > 0 && G >= 8 && (long long)N * J < 0x7fffffffLL && (long long)J * G < (1 << 24);
}

}  // namespace

extern "C" {

size_t rpgp_ski_plan_bytes(int64_t N, int J, int G) {
  if (!plan_args_ok(N, J, G)) return 0;
  return plan_view(nullptr, N, J, G).bytes;
}

size_t rpgp_ski_plan_workspace_bytes(int64_t N, int J, int G) {
  if (!plan_args_ok(N, J, G)) return 0;
  const size_t nj = (size_t)N * J;
  size_t temp = 0;
  unsigned *k = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, temp, k, k, k, k, nj, 0, 32, hipStreamDefault);
  return 4 * align256(nj * sizeof(unsigned)) + align256(temp) + 256;
}

int rpgp_ski_plan(const float *Z, const float *grid_params, int64_t N, int ldz, int J, int G, void *plan, size_t plan_bytes,
                  void *workspace, size_t workspace_bytes, void *stream) {
  if (!Z || !grid_params || !plan || !plan_args_ok(N, J, G) || ldz < J) return RPGP_EINVAL;
  if (plan_bytes < rpgp_ski_plan_bytes(N, J, G)) return RPGP_EWORKSPACE;
  if (!workspace || workspace_bytes < rpgp_ski_plan_workspace_bytes(N, J, G)) return RPGP_EWORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const PlanView pv = plan_view(plan, N, J, G);
  const size_t nj = (size_t)N * J;
  char *w = reinterpret_cast<char *>(workspace);
  unsigned *keys_in = reinterpret_cast<unsigned *>(w); w += align256(nj * sizeof(unsigned));
  unsigned *keys_out = reinterpret_cast<unsigned *>(w); w += align256(nj * sizeof(unsigned));
  unsigned *vals_in = reinterpret_cast<unsigned *>(w); w += align256(nj * sizeof(unsigned));
  unsigned *vals_out = reinterpret_cast<unsigned *>(w); w += align256(nj * sizeof(unsigned));
  void *temp = w;
  size_t temp_bytes = workspace_bytes - (size_t)(w - reinterpret_cast<char *>(workspace));
  long long nblk = ((long long)nj + 255) / 256;
  if (nblk > 4096) nblk = 4096;
  hipLaunchKernelGGL(plan_keys_kernel, dim3((unsigned)nblk), dim3(256), 0, st, Z, grid_params, (long long)N, ldz, J, G, keys_in,
                     vals_in, pv.fnat);
  int rc = launch_status();
  if (rc) return rc;
  int bits = 1;
  while ((1LL << bits) < (long long)J * G) ++bits;
  // stable LSD radix sort on the (projection, cell) key: entries of a cell keep their point order
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, nj, 0, (unsigned)bits, st);
  if (e != hipSuccess) return (int)e;
  const int cells = J * G;
  const int G16 = (G + 15) & ~15;
  const int nthreads = (cells + 1 > J * G16 ? cells + 1 : J * G16);
  hipLaunchKernelGGL(plan_starts_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, keys_out, (long long)nj,
                     cells, pv.cell_start, grid_params, G, pv.tcol);
  hipLaunchKernelGGL(plan_finish_kernel, dim3((unsigned)nblk), dim3(256), 0, st, keys_out, vals_out, pv.fnat, (long long)N, J, G,
                     (long long)nj, pv.rec);
  return launch_status();
}

int rpgp_ski_scatter_planned(const void *plan, const float *V, double *hist, int64_t N, int J, int G, int T, void *workspace,
                             size_t workspace_bytes, void *stream) {
  if (!plan || !V || !hist || !plan_args_ok(N, J, G) || T <= 0 || T > 12) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  if (!scratch_fits(J, G)) return RPGP_EWORKSPACE;
  const PlanView pv = plan_view(const_cast<void *>(plan), N, J, G);
  float *partial = reinterpret_cast<float *>(workspace) + rpgp_internal::ski_scratch_offset_floats(J, G, T);
  return scatter_planned(pv, V, hist, partial, N, J, G, T, T, 0, reinterpret_cast<hipStream_t>(stream));
}

int rpgp_ski_bilinear_scatter_planned(const void *plan, const float *L, const float *R, double *hist2, int64_t N, int J, int G,
                                      int T, void *workspace, size_t workspace_bytes, void *stream) {
  if (!plan || !L || !R || !hist2 || !plan_args_ok(N, J, G) || T <= 0 || T > 12) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  if (!scratch_fits(J, G)) return RPGP_EWORKSPACE;
  const PlanView pv = plan_view(const_cast<void *>(plan), N, J, G);
  float *partial = reinterpret_cast<float *>(workspace) + rpgp_internal::ski_scratch_offset_floats(J, G, T);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // two cell-sorted scatters into column offsets 0 and T of the [J][G][2T] histogram
  int rc = scatter_planned(pv, L, hist2, partial, N, J, G, T, 2 * T, 0, st);
  if (rc) return rc;
  return scatter_planned(pv, R, hist2, partial, N, J, G, T, 2 * T, T, st);
}

int rpgp_ski_mvm_planned(const void *plan, const float *Z, const float *grid_params, const float *V, float *out, int64_t N,
                         int ldz, int J, int G, int T, float scale, float noise, void *workspace, size_t workspace_bytes,
                         void *stream) {
  if (!plan || !Z || !grid_params || !V || !out || !plan_args_ok(N, J, G) || T <= 0 || T > 12 || ldz < J) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  if (!scratch_fits(J, G)) return RPGP_EWORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const PlanView pv = plan_view(const_cast<void *>(plan), N, J, G);
  double *hist = reinterpret_cast<double *>(workspace);
  float *H = reinterpret_cast<float *>(workspace) + 2 * (size_t)J * G * (2 * T);
  float *partial = reinterpret_cast<float *>(workspace) + rpgp_internal::ski_scratch_offset_floats(J, G, T);
  int rc = scatter_planned(pv, V, hist, partial, N, J, G, T, T, 0, st);
  if (rc) return rc;
  rc = rpgp_internal::ski_toeplitz_launch(hist, 1, grid_params, H, J, G, T, st, pv.tcol);
  if (rc) return rc;
  return gather_planned(Z, grid_params, H, V, out, N, ldz, J, G, T, scale, noise, st);
}

int rpgp_ski_gather_fast(const void *plan, const float *Z, const float *grid_params, const float *H, const float *V, float *out,
                         int64_t M, int ldz, int J, int G, int T, float scale, float noise, void *stream) {
  (void)plan;                                  // (kept in the signature: the gather reads the coordinates, not the plan)
  if (!Z || !grid_params || !H || !out || M <= 0 || J <= 0 || G < 8 || T <= 0 || T > 12 || ldz < J) return RPGP_EINVAL;
  if (noise != 0.f && !V) return RPGP_EINVAL;
  return gather_planned(Z, grid_params, H, V, out, (long long)M, ldz, J, G, T, scale, noise, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
