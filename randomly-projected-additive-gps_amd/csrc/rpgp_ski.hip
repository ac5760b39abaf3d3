// rpgp_ski.hip — the planned (cell-sorted) SKI product.
//
// Z is fixed for a whole CG solve (one hyper-parameter step = 20 .. 100 products), so the interpolation structure is
// prepared ONCE per step (`rpgp_ski_plan`): every projection's points are sorted by the grid cell of their first tap
// (stable radix sort: the order inside a cell is the point order, so everything downstream is deterministic), with the
// tap fraction kept in sorted order.  The scatter  hist_j = W_j^T V  is then a SEGMENTED REDUCTION over the sorted
// points — no atomics at all (round 2: fixed-point LDS atomics, 58 - 65 us for 22 MB at the C5 shape = 0.36 TB/s):
//   ski_scatter_sorted : one wave per (cell, segment of <= 64 points): the segment's point indices and fractions arrive in
//                        two coalesced loads, the V rows are gathered with all loads of the segment in flight, every
//                        lane accumulates its 4 tap sums in registers, lane groups are combined by shuffles;
//                        partial[item][tap][t]  (fp32, <= 64 terms each)
//   ski_cellsum        : hist[j][g][t] = sum over taps k and the items of cell g - k, fixed order, FLOAT64 — replaces the
//                        per-chunk slabs + slab sum
//   Toeplitz product   : ski_toeplitz_mfma_kernel (rpgp_kernels.hip), first column from the plan (no exp per product)
//   gather             : with the whole H (J x G x T floats) resident in LDS when it fits (C5: 135 KB of the 160 KB):
//                        the 12 tap rows per point are LDS reads instead of L2 requests that miss L1 four lines at a time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#include <cstring>
#include <mutex>
#include <unordered_set>
#include <stdlib.h>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/rpgp.h"
#include "rpgp_internal.h"
#include "rpgp_ski_common.h"

namespace {

constexpr int kSeg = 256;                      // points per scatter item: one workgroup, 64 points per wave

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
  int *item_start;     // [J * G + 1]  first scatter item of the cell (exclusive scan of ceil(count / kSeg))
  int2 *item_info;     // [max_items]  (first sorted entry, number of points) of every scatter item
  double *tcol;        // [J][G16]  first column of the Toeplitz matrix of every projection, exp(-0.5 (k h_j)^2)
  // chunked product (round 5; only when chunk_ok(N, J, G)) -- see "chunked product" below
  const void *base;    // the plan blob (key of chunked_plans())
  int CH, nch;         // rows per chunk, chunks
  int4 *winfo;         // [nch * J]       (first cell, cells spanned, first window row in the window table, window rows)
  float *uloc;         // [nch * J][CH]   clamped grid coordinate of every (chunk row, projection)
  uint16_t *lperm;     // [nch * J][CH]   the chunk's rows sorted by (cell, row)
  uint16_t *coff;      // [nch * J][G]    first sorted entry of every cell of the window (+ one closing entry)
  size_t bytes;
};

// ---- chunked product: geometry ---------------------------------------------------------------------------------------------
// The rows are cut into `nch` contiguous chunks of CH rows, one per workgroup and about one per CU; CH <= 2048 so that
// chunk-local row numbers and offsets are 16-bit and a chunk's right-hand-side rows fit in LDS.
constexpr int kChunkTarget = 256;              // chunks aimed at (the CUs of an MI355X)
constexpr int kChunkMaxRows = 2048;
inline int chunk_rows(long long N) {
  long long ch = (N + kChunkTarget - 1) / kChunkTarget;
  ch = (ch + 15) & ~15LL;
  if (ch < 256) ch = 256;
  if (ch > kChunkMaxRows) ch = kChunkMaxRows;
  return (int)ch;
}
// The chunked product is OPT-IN (RPGP_SKI_CHUNK=1 or rpgp_ski_chunk_mode(1)): measured at the C5 shape it does not beat the
// cell-sorted product of rounds 3 - 4.  A plan built while the mode is on carries the tables of BOTH
// products (the set below remembers which plans do), so the switch may be flipped between two products of such a plan.
inline int &chunk_mode_ref() {
  static int mode = [] {
    const char *e = getenv("RPGP_SKI_CHUNK");
    return (e && e[0] == '1') ? 1 : 0;
  }();
  return mode;
}
inline bool chunk_env_on() { return chunk_mode_ref() != 0; }
struct ChunkedPlans {
  std::mutex mu;
  std::unordered_set<const void *> set;
  void mark(const void *plan, bool chunked) {
    std::lock_guard<std::mutex> g(mu);
    if (chunked) set.insert(plan);
    else set.erase(plan);
  }
  bool has(const void *plan) {
    std::lock_guard<std::mutex> g(mu);
    return set.count(plan) != 0;
  }
};
inline ChunkedPlans &chunked_plans() {
  static ChunkedPlans p;
  return p;
}
// The window table lives in the scatter slabs of the SKI workspace (rpgp_ski_workspace_bytes: 1024 / J chunks of J x G x 12
// floats), which bounds chunks x projections by 1024; few projections, enough rows to fill the chip.
inline bool chunk_shape_ok(long long N, int J, int G) {
  if (J < 1 || J > 4 || G < 8 || G > 2048 || N < 16384) return false;
  const int ch = chunk_rows(N);
  const long long nch = (N + ch - 1) / ch;
  return nch * J <= 1024 && nch <= (1024 + J - 1) / J;
}

inline long long max_items(long long N, int J, int G);

inline PlanView plan_view(void *base, long long N, int J, int G) {
  PlanView v;
  v.base = base;
  char *p = reinterpret_cast<char *>(base);
  const size_t nj = (size_t)N * J, cells = (size_t)J * G + 1, G16 = (size_t)((G + 15) & ~15);
  v.rec = reinterpret_cast<int2 *>(p); p += align256(nj * sizeof(int2));
  v.fnat = reinterpret_cast<float *>(p); p += align256(nj * sizeof(float));
  v.cell_start = reinterpret_cast<int *>(p); p += align256(cells * sizeof(int));
  v.item_start = reinterpret_cast<int *>(p); p += align256(cells * sizeof(int));
  v.item_info = reinterpret_cast<int2 *>(p); p += align256((size_t)max_items(N, J, G) * sizeof(int2));
  v.tcol = reinterpret_cast<double *>(p); p += align256((size_t)J * G16 * sizeof(double));
  v.CH = v.nch = 0;
  v.winfo = nullptr; v.uloc = nullptr; v.lperm = nullptr; v.coff = nullptr;
  if (chunk_shape_ok(N, J, G)) {       // (the area is there whatever RPGP_SKI_CHUNK says: the size must not depend on it)
    v.CH = chunk_rows(N);
    v.nch = (int)((N + v.CH - 1) / v.CH);
    const size_t cj = (size_t)v.nch * J;
    v.winfo = reinterpret_cast<int4 *>(p); p += align256(cj * sizeof(int4));
    v.uloc = reinterpret_cast<float *>(p); p += align256(cj * v.CH * sizeof(float));
    v.lperm = reinterpret_cast<uint16_t *>(p); p += align256(cj * v.CH * sizeof(uint16_t));
    v.coff = reinterpret_cast<uint16_t *>(p); p += align256(cj * G * sizeof(uint16_t));
  }
  v.bytes = (size_t)(p - reinterpret_cast<char *>(base));
  return v;
}

inline long long max_items(long long N, int J, int G) { return (N * J + kSeg - 1) / kSeg + (long long)J * G; }

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
    tcol[c] = k < G ? exp(-0.5 * d * d) : 0.0;
  }
}

// item_start = exclusive scan over the cells of ceil(count / kSeg); one workgroup, 1024 threads, fixed order
__global__ __launch_bounds__(1024) void plan_items_kernel(const int *__restrict__ cell_start, int cells,
                                                          int *__restrict__ item_start, int2 *__restrict__ item_info) {
  __shared__ int ssum[1024];
  const int per = (cells + 1023) / 1024;
  const int c0 = threadIdx.x * per, c1 = (c0 + per < cells) ? c0 + per : cells;
  int loc = 0;
  for (int c = c0; c < c1; ++c) loc += (cell_start[c + 1] - cell_start[c] + kSeg - 1) / kSeg;
  ssum[threadIdx.x] = loc;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = 0;
    if ((int)threadIdx.x >= off) v = ssum[threadIdx.x - off];
    __syncthreads();
    ssum[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x == 0 ? 0 : ssum[threadIdx.x - 1];
  for (int c = c0; c < c1; ++c) {
    item_start[c] = run;
    const int b = cell_start[c], e = cell_start[c + 1];
    for (int s = b; s < e; s += kSeg) item_info[run++] = make_int2(s, e - s < kSeg ? e - s : kSeg);
  }
  if (threadIdx.x == 1023) item_start[cells] = ssum[1023];
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

// ---- scatter over the sorted points ---------------------------------------------------------------------------------------
// One workgroup per item (<= 256 points of one cell), one wave per 64 of them.  Lanes are (point, t) with LPP = 1 / 4 / 16
// lanes per point (TT = 1 / 4 / 12 columns per pass).  The four waves' sums are added in a fixed order.
// Lane layout: LPP lanes per point and CPL columns per lane (TT = LPP * CPL columns per pass): 1 x 1 for a single column,
// 4 x 1 for up to 4, 4 x 3 for up to 12.  What bounds this kernel is the L1 -> register return path (64 B/clk per CU:
// every lane of a wave receives its own copy of a broadcast load), so the per-point records (index, 4 weights) are read by
// 4 lanes per point, not 16: 8 KB of returned data per 64 points instead of 24 KB.
template <int LPP, int CPL>
__global__ __launch_bounds__(256) void ski_scatter_sorted_kernel(const int2 *__restrict__ rec,
                                                                 const int2 *__restrict__ item_info,
                                                                 const int *__restrict__ item_start, int cells,
                                                                 const float *__restrict__ V, float *__restrict__ partial,
                                                                 int T, int t0, int tcnt) {
  constexpr int TT = LPP * CPL;
  constexpr int PPW = 64 / LPP;              // points per wave step: 64 / 16
  constexpr int STEPS = 64 / PPW;            // 1 / 4
  __shared__ float sP[4][4][12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = blockIdx.x;
  if (item >= item_start[cells]) return;
  const int2 info = item_info[item];           // (first sorted entry, points) — one load instead of a binary search
  const int s0 = info.x + wave * 64;
  int cnt = info.y - wave * 64;
  cnt = cnt > 64 ? 64 : (cnt < 0 ? 0 : cnt);
  const int c = lane % LPP, pg = lane / LPP;
  float acc[4][CPL];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < CPL; ++r) acc[k][r] = 0.f;
  if (cnt > 0) {
    int2 pr[STEPS];
#pragma unroll
    for (int m = 0; m < STEPS; ++m) {
      const int src = m * PPW + pg;
      pr[m] = rec[s0 + (src < cnt ? src : 0)];
    }
    float v[STEPS][CPL];
    float4 w[STEPS];
#pragma unroll
    for (int m = 0; m < STEPS; ++m) {
      const int src = m * PPW + pg;
      const int pi_m = pr[m].x;
      w[m] = ski_weights_of_frac(__builtin_bit_cast(float, pr[m].y));
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        const int col = c + LPP * r;
        const bool ok = src < cnt && col < tcnt;
        const float x = V[ok ? (size_t)pi_m * T + t0 + col : 0];
        v[m][r] = ok ? x : 0.f;
      }
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
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < CPL; ++r) acc[k][r] += __shfl_xor(acc[k][r], off, 64);
    }
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
    if (tt < tcnt) partial[((size_t)item * 4 + k) * TT + tt] = ((sP[0][k][tt] + sP[1][k][tt]) + sP[2][k][tt]) + sP[3][k][tt];
  }
}

// ---- scatter, one workgroup per CELL (round 4) ------------------------------------------------------------------------------
// The item form above starts ~7 600 short workgroups (<= 256 points each) whose lifetime is a chain of dependent loads (item
// record -> point indices -> V rows), and leaves the per-cell sums to a second kernel that walks the items of four neighbouring
// cells per output.  Here a workgroup owns one interpolation cell of one projection and walks ALL its points, 256 per round
// (wave w takes 64 of them), with the next round's point indices requested before the current round's V rows are consumed:
// cell bounds -> indices -> rows is paid once per cell, not once per 256 points; the four taps' sums of the whole cell leave in
// one record (cellpart[cell][tap][t]), so the histogram is four shifted reads per entry (ski_cellsum4_kernel) — no item
// lists.  Sums: per lane over the rounds, then the lanes of a wave (xor tree), then the four waves in order: fixed order,
// bitwise reproducible.  Same lane layout as the item kernel.
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

// hist[j][g][hoff + t] (row stride HT, float64) = sum_k sum_{items of cell (j, g - k)} partial[item][k][t]
__global__ __launch_bounds__(256) void ski_cellsum_kernel(const float *__restrict__ partial, const int *__restrict__ item_start,
                                                          double *__restrict__ hist, int J, int G, int TT, int tcnt, int HT,
                                                          int hoff) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)J * G * TT) return;
  const int t = (int)(e % TT);
  const long long jg = e / TT;
  const int g = (int)(jg % G), j = (int)(jg / G);
  if (t >= tcnt) return;
  // the four taps' item ranges are requested together (clamped cells, pinned by an empty asm), then the first item of every
  // tap; the sums run in the original (tap, item) order.  The plain nest was ~12 dependent round trips per thread.
  int i0[4], i1[4];
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = g - k;
    ok[k] = !(c < 0 || c > G - 4);
    const int cc = ok[k] ? c : 0;
    i0[k] = item_start[j * G + cc];
    i1[k] = item_start[j * G + cc + 1];
  }
  asm volatile("" ::"v"(i0[0]), "v"(i0[1]), "v"(i0[2]), "v"(i0[3]), "v"(i1[0]), "v"(i1[1]), "v"(i1[2]), "v"(i1[3]));
  float first[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)        // (an empty range reads item 0 — always there — and discards it)
    first[k] = partial[((size_t)(i0[k] < i1[k] ? i0[k] : 0) * 4 + k) * TT + t];
  asm volatile("" ::"v"(first[0]), "v"(first[1]), "v"(first[2]), "v"(first[3]));
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!ok[k] || i0[k] >= i1[k]) continue;
    acc += (double)first[k];
    for (int it = i0[k] + 1; it < i1[k]; ++it) acc += (double)partial[((size_t)it * 4 + k) * TT + t];
  }
  hist[jg * HT + hoff + t] = acc;
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

// PASSA (round 5): the executor's pass A — partial p . Ap per column and partial L^T (Ap) — accumulated here, where Ap is
// formed and p (= V) is already in registers for the noise term: the workgroup leaves one slab of the executor's reduction
// format (csrc/rpgp_cg.hip, k_pass_a: [16 column sums][16 zeros][16 x 16 L^T Ap]) and the separate pass over p, Ap and L
// (11 us per iteration at the C5 shape) is not launched.  Fixed summation order: per lane over its rows, the sixteen rows of a
// wave step by xor shuffles, the sixteen waves in order.
template <int CPL, int U, bool PASSA = false>
__global__ __launch_bounds__(1024) void ski_gather_lds_kernel(const float *__restrict__ Z, const float *__restrict__ gp,
                                                              const float *__restrict__ H, const float *__restrict__ V,
                                                              float *__restrict__ out, long long M, int ldz, int J, int G, int T,
                                                              float scale, float noise, const float *__restrict__ Lp = nullptr,
                                                              int K = 0, float *__restrict__ partA = nullptr) {
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
  constexpr int KA = PASSA ? rpgp_internal::kCgMaxK : 1;
  float pa_dot[CPL], pa_lt[KA][CPL];
#pragma unroll
  for (int r = 0; r < CPL; ++r) {
    pa_dot[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KA; ++kk) pa_lt[kk][r] = 0.f;
  }
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
      // pass A: the row of L is requested here, at the head of the group's tap reads (asked for where it is used, the four
      // requests of every group were a round trip with nothing to hide behind: 31 us for the kernel)
      float lrow[KA];
      if constexpr (PASSA) {
        const long long pl = live ? p : 0;
        if (K == 15) {                         // the reference's preconditioner rank: the row as four 16-byte requests
          typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
          const float *lp = Lp + pl * 15;
          const f4u a = *reinterpret_cast<const f4u *>(lp), b = *reinterpret_cast<const f4u *>(lp + 4);
          const f4u d = *reinterpret_cast<const f4u *>(lp + 8), e = *reinterpret_cast<const f4u *>(lp + 11);
          lrow[0] = a.x; lrow[1] = a.y; lrow[2] = a.z; lrow[3] = a.w; lrow[4] = b.x; lrow[5] = b.y; lrow[6] = b.z; lrow[7] = b.w;
          lrow[8] = d.x; lrow[9] = d.y; lrow[10] = d.z; lrow[11] = e.x; lrow[12] = e.y; lrow[13] = e.z; lrow[14] = e.w;
          if (KA > 15) lrow[KA - 1] = 0.f;
        } else {
#pragma unroll
          for (int kk = 0; kk < KA; ++kk) {
            const float lv = Lp[pl * K + (kk < K ? kk : 0)];
            lrow[kk] = kk < K ? lv : 0.f;
          }
        }
      }
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
        float o[CPL];
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
          o[r] = __builtin_fmaf(noise, vin[u][r], scale * (float)acc[r]);
          if (c + 4 * r < T) out[p * T + c + 4 * r] = o[r];
          else o[r] = 0.f;
        }
        if constexpr (PASSA) {
#pragma unroll
          for (int r = 0; r < CPL; ++r) pa_dot[r] = __builtin_fmaf(vin[u][r], o[r], pa_dot[r]);
#pragma unroll
          for (int kk = 0; kk < KA; ++kk)
#pragma unroll
            for (int r = 0; r < CPL; ++r) pa_lt[kk][r] = __builtin_fmaf(lrow[kk], o[r], pa_lt[kk][r]);
        }
      }
    }
  }
  if constexpr (PASSA) {
    // the 16 rows of a wave step (lanes with the same c), then the 16 waves through the (no longer needed) image of H
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) {
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        pa_dot[r] += __shfl_xor(pa_dot[r], off, 64);
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) pa_lt[kk][r] += __shfl_xor(pa_lt[kk][r], off, 64);
      }
    }
    __syncthreads();                           // (every wave is done with sH)
    float *sw = sH;                            // [16 waves][17][16]: row 0 = column sums, rows 1 .. 16 = L^T Ap
    const int wave = threadIdx.x >> 6;
    if (pg == 0) {
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        const int col = c + 4 * r;             // < 16
        sw[(wave * 17 + 0) * 16 + col] = pa_dot[r];
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) sw[(wave * 17 + 1 + kk) * 16 + col] = pa_lt[kk][r];
      }
    }
    __syncthreads();
    float *dst = partA + (size_t)blockIdx.x * rpgp_internal::kCgRedW;
    for (int e = threadIdx.x; e < rpgp_internal::kCgRedW; e += 1024) {
      float v = 0.f;
      const int row = e < 16 ? 0 : (e >= rpgp_internal::kCgRedLt ? 1 + (e - rpgp_internal::kCgRedLt) / 16 : -1);
      const int col = e & 15;
      if (row >= 0 && col < 4 * CPL && row <= KA) {
        float acc = sw[(0 * 17 + row) * 16 + col];
        for (int w = 1; w < 16; ++w) acc += sw[(w * 17 + row) * 16 + col];
        v = acc;
      }
      dst[e] = v;
    }
  }
}

// ============================================================================================================================
// Chunked product (round 5).  The cell-sorted scatter above reads, per projection, every row of V in grid-cell order: J random
// passes over a 17 MB array in 44-byte pieces (C5: 93 MB fetched for 51 MB useful) behind three dependent round trips.  Here a
// workgroup owns a contiguous CHUNK of rows (about N / 256 of them):
//   scatter : the chunk's rows of V are copied to LDS ONCE, in storage order (one coalesced block), and serve all J projections;
//             per projection the chunk's rows are walked in (cell, row) order through a 16-bit chunk-local permutation, a lane
//             group per cell — a segmented sum without atomics whose operands are LDS reads; the four taps of neighbouring
//             cells are combined by lane shuffles, and the chunk leaves one WINDOW of grid rows [first cell, last cell + 3]
//             per projection.  With the rows stored in a locality-preserving order (training.locality_order) a chunk's window
//             is ~100 - 150 of the 1024 grid rows; in any order the product is the same, only the windows grow.
//   combine : hist[j][g] = the windows that cover g, added in chunk order in float64 (fixed order: bitwise reproducible).
//   gather  : a chunk reads the window of H it needs into LDS (not all of H) and its rows' stencils once.
// The plan keeps, per (chunk, projection): the clamped grid coordinates, the permutation, the per-cell offsets and the window.
// ============================================================================================================================
constexpr int kPlanItems = 8;                  // rows per thread of the plan's block sort (256 threads x 8 = kChunkMaxRows)

// one workgroup per (chunk, projection): grid coordinates, (cell, row) block sort, per-cell offsets, window bounds
__global__ __launch_bounds__(256) void chunk_plan_kernel(const float *__restrict__ Z, const float *__restrict__ gp, long long N,
                                                         int ldz, int J, int G, int CH, int4 *__restrict__ winfo,
                                                         float *__restrict__ uloc, uint16_t *__restrict__ lperm,
                                                         uint16_t *__restrict__ coff) {
  using Sort = rocprim::block_radix_sort<unsigned, 256, kPlanItems>;
  __shared__ typename Sort::storage_type sort_storage;
  __shared__ int cnt[kChunkMaxRows + 8];       // points per cell (G <= 2048), then their exclusive scan
  __shared__ int red[2][4];
  __shared__ int wsum[4];
  const int chunk = blockIdx.x, j = blockIdx.y;
  const long long base = (long long)chunk * CH;
  const int n = (int)((N - base) < CH ? (N - base) : CH);
  const float *gj = ski_grid_of(gp, J, j);
  const float g0 = gj[0], inv_h = gj[2];
  const size_t cj = (size_t)chunk * J + j;
  for (int c = threadIdx.x; c < G + 8; c += 256) cnt[c] = 0;
  __syncthreads();
  unsigned keys[kPlanItems];
  int cmin = 0x7fffffff, cmax = -1;
#pragma unroll
  for (int i = 0; i < kPlanItems; ++i) {
    const int p = threadIdx.x * kPlanItems + i;
    keys[i] = 0xffffffffu;
    if (p < n) {
      const float u = ski_grid_coord(Z[(base + p) * ldz + j], g0, inv_h, G);
      float w[4], dw[4];
      const int cell = ski_taps_u<false>(u, inv_h, G, w, dw);
      uloc[cj * CH + p] = u;
      keys[i] = ((unsigned)cell << 11) | (unsigned)p;
      atomicAdd(&cnt[cell], 1);
      cmin = cell < cmin ? cell : cmin;
      cmax = cell > cmax ? cell : cmax;
    } else if (p < CH) {
      uloc[cj * CH + p] = 1.0f;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int a = __shfl_xor(cmin, off, 64), b = __shfl_xor(cmax, off, 64);
    cmin = a < cmin ? a : cmin;
    cmax = b > cmax ? b : cmax;
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = cmin;
    red[1][threadIdx.x >> 6] = cmax;
  }
  Sort().sort(keys, sort_storage, 0, 22);      // 11 bits of row, <= 11 bits of cell; padding keys sort last
  __syncthreads();
  cmin = min(min(red[0][0], red[0][1]), min(red[0][2], red[0][3]));
  cmax = max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3]));
#pragma unroll
  for (int i = 0; i < kPlanItems; ++i) {
    const int pos = threadIdx.x * kPlanItems + i;
    if (pos < CH) lperm[cj * CH + pos] = (uint16_t)(pos < n ? (keys[i] & 2047u) : 0u);
  }
  // exclusive scan of the counts of cells cmin .. cmax (+ the closing entry): 8 consecutive cells per thread
  const int ncell = cmax - cmin + 1;
  int loc[8], run = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x * 8 + i;
    loc[i] = run;
    run += (c < ncell) ? cnt[cmin + c] : 0;
  }
  int incl = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if ((int)(threadIdx.x & 63) >= off) incl += v;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int before = incl - run;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wsum[w];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x * 8 + i;
    if (c <= ncell && c < G) coff[cj * G + c] = (uint16_t)(before + loc[i]);
  }
  if (threadIdx.x == 0) winfo[cj] = make_int4(cmin, ncell, 0, ncell + 3);
}

// first window row of every (chunk, projection) in the window table: exclusive scan of the window lengths, one workgroup
__global__ __launch_bounds__(1024) void chunk_scan_kernel(int4 *__restrict__ winfo, int n) {
  __shared__ int ssum[1024];
  const int len = (int)threadIdx.x < n ? winfo[threadIdx.x].w : 0;
  ssum[threadIdx.x] = len;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = 0;
    if ((int)threadIdx.x >= off) v = ssum[threadIdx.x - off];
    __syncthreads();
    ssum[threadIdx.x] += v;
    __syncthreads();
  }
  if ((int)threadIdx.x < n) winfo[threadIdx.x].z = ssum[threadIdx.x] - len;
}

// LDS rows of P = 4 LPP floats: element e of the chunk's contiguous n x T block -> (row, column) without a division
// (e < 2048 * 12 < 2^15, T <= 12: floor(e / T) = (e * ceil(2^20 / T)) >> 20 exactly)
__device__ __forceinline__ unsigned div_small(unsigned e, unsigned magic) { return (e * magic) >> 20; }

constexpr int kChunkMaxJ = 4;                  // projections of the chunked product (chunk_shape_ok)
constexpr size_t kChunkLdsBytes = 160 * 1024 - 256;       // dynamic LDS a chunk workgroup may take (one workgroup per CU)

// The scatter of one chunk.  Every global load of the workgroup is requested before the first is consumed: the chunk's rows of
// V (<= 6 float4 per thread) and, for the projections of the first round, the rows' grid coordinates, their permutation, the
// per-cell offsets and the window record — one memory round trip, then LDS only.  NJB projections are resident at a time
// (the host takes the most that fit beside V: all three at the C5 shape), and their (projection, cell group) items are dealt
// to the 16 waves together.
template <int LPP>
__global__ __launch_bounds__(1024) void ski_chunk_scatter_kernel(const int4 *__restrict__ winfo, const float *__restrict__ uloc,
                                                                 const uint16_t *__restrict__ lperm,
                                                                 const uint16_t *__restrict__ coff,
                                                                 const float *__restrict__ V, float *__restrict__ win,
                                                                 long long N, int J, int G, int T, int CH, int NJB) {
  constexpr int P = 4 * LPP;                   // floats per LDS row of V / per window row
  constexpr int SLOTS = 64 / LPP;              // cells per wave step (21 / 32 / 64)
  constexpr int NOUT = SLOTS - 3;              // of which this many produce a window row (3 leading cells are re-done: halo)
  constexpr int VQ = kChunkMaxRows * 12 / 4 / 1024;      // float4 of V per thread (6)
  extern __shared__ float4 lds4[];
  float *Vs = reinterpret_cast<float *>(lds4);                       // [CH][P]
  float4 *Ws = lds4 + (size_t)CH * LPP;                               // [NJB][CH]     tap weights
  uint16_t *lp = reinterpret_cast<uint16_t *>(Ws + (size_t)NJB * CH); // [NJB][CH]     rows in (cell, row) order
  uint16_t *co = lp + (size_t)NJB * CH;                               // [NJB][G + 2]  per-cell offsets of the window
  __shared__ int4 swi[kChunkMaxJ];
  const int chunk = blockIdx.x;
  const long long base = (long long)chunk * CH;
  const int n = (int)((N - base) < CH ? (N - base) : CH);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPP, q = lane - slot * LPP;
  const unsigned magic = ((1u << 20) + T - 1) / T;
  const unsigned total = (unsigned)n * T, total4 = total >> 2;

  // ---- requests: V, then the tables of the first round ------------------------------------------------------------------
  const float4 *src = reinterpret_cast<const float4 *>(V + base * T);
  float4 vq[VQ];
#pragma unroll
  for (int k = 0; k < VQ; ++k) {
    const unsigned e4 = threadIdx.x + k * 1024;
    vq[k] = src[e4 < total4 ? e4 : 0];         // (unconditional, clamped: every thread has VQ loads in flight)
  }
  float vtail = 0.f;
  if (threadIdx.x < (total & 3u)) vtail = V[base * T + (total4 << 2) + threadIdx.x];
  float ru[kChunkMaxJ][2];
  unsigned rl[kChunkMaxJ][2], rc[kChunkMaxJ][2];
  auto request = [&](int j0) {
#pragma unroll
    for (int jl = 0; jl < kChunkMaxJ; ++jl) {
      const int j = j0 + jl;
      const bool live = jl < NJB && j < J;
      const size_t cj = (size_t)chunk * J + (live ? j : 0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int p = threadIdx.x + h * 1024;
        const int pc = p < CH ? p : 0, cc = p < G ? p : 0;
        ru[jl][h] = uloc[cj * CH + pc];
        rl[jl][h] = lperm[cj * CH + pc];
        rc[jl][h] = coff[cj * G + cc];
      }
      if (threadIdx.x == 0 && live) swi[jl] = winfo[cj];
    }
  };
  request(0);

  // ---- V into LDS rows of P floats ------------------------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < VQ; ++k) {
    const unsigned e4 = threadIdx.x + k * 1024;
    if (e4 < total4) {
      const float xs[4] = {vq[k].x, vq[k].y, vq[k].z, vq[k].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned e = 4 * e4 + i, row = div_small(e, magic);
        Vs[row * P + (e - row * T)] = xs[i];
      }
    }
  }
  if (threadIdx.x < (total & 3u)) {
    const unsigned e = (total4 << 2) + threadIdx.x, row = div_small(e, magic);
    Vs[row * P + (e - row * T)] = vtail;
  }
  if (T < P)
    for (unsigned r = threadIdx.x; r < (unsigned)n; r += 1024)
      for (int c = T; c < P; ++c) Vs[r * P + c] = 0.f;

  for (int j0 = 0; j0 < J; j0 += NJB) {
    if (j0 > 0) {
      __syncthreads();                         // (the previous round's readers are done)
      request(j0);
    }
    const int nj = J - j0 < NJB ? J - j0 : NJB;
#pragma unroll
    for (int jl = 0; jl < kChunkMaxJ; ++jl) {
      if (jl < nj) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int p = threadIdx.x + h * 1024;
          if (p < n) {
            float w[4], dw[4];
            (void)ski_taps_u<false>(ru[jl][h], 0.f, G, w, dw);
            Ws[(size_t)jl * CH + p] = make_float4(w[0], w[1], w[2], w[3]);
            lp[(size_t)jl * CH + p] = (uint16_t)rl[jl][h];
          }
          if (p < G) co[jl * (G + 2) + p] = (uint16_t)rc[jl][h];
        }
      }
    }
    __syncthreads();
    // items of the round: (projection jl, cell group grp), dealt to the waves in one sequence
    int gstart[kChunkMaxJ + 1];
    gstart[0] = 0;
#pragma unroll
    for (int jl = 0; jl < kChunkMaxJ; ++jl)
      gstart[jl + 1] = gstart[jl] + (jl < nj ? (swi[jl].y + 3 + NOUT - 1) / NOUT : 0);
    for (int item = wave; item < gstart[kChunkMaxJ]; item += 16) {
      int jl = 0;
#pragma unroll
      for (int t = 1; t < kChunkMaxJ; ++t) jl += (item >= gstart[t]) ? 1 : 0;
      const int grp = item - gstart[jl];
      const int4 wi = swi[jl];
      const int ncell = wi.y;
      const float4 *Wj = Ws + (size_t)jl * CH;
      const uint16_t *lj = lp + (size_t)jl * CH, *cj_ = co + jl * (G + 2);
      const int cl = grp * NOUT + slot - 3;    // cell of this lane group, relative to the window's first cell
      const bool valid = slot < SLOTS && cl >= 0 && cl < ncell;
      int it = valid ? cj_[cl] : 0;
      const int end = valid ? cj_[cl + 1] : 0;
      float acc[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[k][r] = 0.f;
      // four rows per trip: the four row numbers, then the eight 16-byte records, are requested together (one row per trip
      // is a chain of two dependent LDS round trips per row: measured 140 ns per row and wave); rows beyond the cell's end
      // repeat the last row with its V masked to zero — the sums keep the cell's row order
      for (; it < end; it += 4) {
        unsigned pr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pr[u] = lj[it + u < end ? it + u : end - 1];
        float4 w[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          w[u] = Wj[pr[u]];
          v[u] = *reinterpret_cast<const float4 *>(Vs + pr[u] * P + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool live = it + u < end;      // (masking V, not the weights: a repeated non-finite row must not add 0 x inf)
          const float ww[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
          const float vv[4] = {live ? v[u].x : 0.f, live ? v[u].y : 0.f, live ? v[u].z : 0.f, live ? v[u].w : 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[0][r] = __builtin_fmaf(ww[0], vv[r], acc[0][r]);
            acc[1][r] = __builtin_fmaf(ww[1], vv[r], acc[1][r]);
            acc[2][r] = __builtin_fmaf(ww[2], vv[r], acc[2][r]);
            acc[3][r] = __builtin_fmaf(ww[3], vv[r], acc[3][r]);
          }
        }
      }
      // window row cl = tap 0 of cell cl + tap 1 of cell cl - 1 + tap 2 of cell cl - 2 + tap 3 of cell cl - 3
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a1 = __shfl_up(acc[1][r], LPP, 64);
        const float a2 = __shfl_up(acc[2][r], 2 * LPP, 64);
        const float a3 = __shfl_up(acc[3][r], 3 * LPP, 64);
        o[r] = ((acc[0][r] + a1) + a2) + a3;
      }
      if (slot >= 3 && slot < SLOTS && cl < ncell + 3)
        *reinterpret_cast<float4 *>(win + ((size_t)(wi.z + cl) * LPP + q) * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// hist[j][g][hoff + t] (row stride HT, float64) = sum over the chunks whose window covers g, in chunk order.  One workgroup
// per (16 grid rows, projection): the covering chunks are listed in LDS with their window records (ordered compaction, one
// round trip), 8 lane groups take every eighth list entry each with all their window reads in flight together, and the eight
// float64 sums are added in a fixed order.
template <int LPP>
__global__ __launch_bounds__(512) void ski_chunk_combine_kernel(const int4 *__restrict__ winfo, const float *__restrict__ win,
                                                                double *__restrict__ hist, int nch, int J, int G, int tcnt,
                                                                int HT, int hoff) {
  constexpr int P = 4 * LPP;
  constexpr int NP = 8;                        // lane groups (parts of the list) per output
  constexpr int UN = 8;                        // window reads in flight per thread
  __shared__ int llo[1024], lhi[1024], lrow[1024];       // window [llo, lhi) and (first table row - llo) of the listed chunks
  __shared__ int wcount[8];
  __shared__ double part[NP][16][12];
  const int j = blockIdx.y, g0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int total = 0;
  for (int c0 = 0; c0 < nch; c0 += 512) {      // ordered compaction of the covering chunks, 512 candidates per round
    const int chunk = c0 + threadIdx.x;
    const int4 wi = winfo[(size_t)(chunk < nch ? chunk : 0) * J + j];
    const bool cov = chunk < nch && wi.x < g0 + 16 && wi.x + wi.w > g0;
    const unsigned long long m = __ballot(cov);
    if (lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    int off = total, all = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      off += w < wave ? wcount[w] : 0;
      all += wcount[w];
    }
    if (cov) {
      const int i = off + __popcll(m & ((1ull << lane) - 1ull));
      llo[i] = wi.x;
      lhi[i] = wi.x + wi.w;
      lrow[i] = wi.z - wi.x;
    }
    total += all;
    __syncthreads();
  }
  // thread = (part r of NP, grid row gl of 16, column group q of LPP); threads beyond 16 NP LPP idle
  const int q = threadIdx.x % LPP, gl = (threadIdx.x / LPP) % 16, r = threadIdx.x / (16 * LPP);
  const int g = g0 + gl;
  if (r < NP) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = r; i0 < total; i0 += NP * UN) {
      float4 x[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {           // unconditional reads of a clamped, valid table row; masked below
        const int i = i0 + u * NP;
        const int ii = i < total ? i : 0;
        const bool ok = i < total && g >= llo[ii] && g < lhi[ii];
        const int row = ok ? lrow[ii] + g : lrow[ii] + llo[ii];
        const float4 y = *reinterpret_cast<const float4 *>(win + ((size_t)row * LPP + q) * 4);
        x[u] = ok ? y : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        acc[0] += (double)x[u].x;
        acc[1] += (double)x[u].y;
        acc[2] += (double)x[u].z;
        acc[3] += (double)x[u].w;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) part[r][gl][4 * q + c] = acc[c];
  }
  __syncthreads();
  if (threadIdx.x < 16 * P) {
    const int gl2 = threadIdx.x / P, c = threadIdx.x % P;
    if (g0 + gl2 < G && c < tcnt) {
      double sum = part[0][gl2][c];
#pragma unroll
      for (int k = 1; k < NP; ++k) sum += part[k][gl2][c];
      hist[((size_t)j * G + g0 + gl2) * HT + hoff + c] = sum;
    }
  }
}

// out[i][t] = scale * sum_j sum_k w_k(z_ij) H[j][idx0 + k][t] + noise V[i][t] for the chunk's rows.  The windows of H the
// chunk touches (all projections': ~3 x 160 rows at the C5 shape) and the rows' grid coordinates are staged in LDS behind ONE
// round of requests; a row is then finished in one go — its stencils derived in the lane (the LDS pipe, which bounds this
// kernel, only serves the coordinate and the four tap rows), the sum over j in float64 in projection order, every product
// formed exactly as in ski_gather_lds_kernel (same bits) — with its V values in flight across the row sums.  A chunk whose
// windows exceed the LDS rows (HCAP: more than ~98 % of the grid in every projection) reads H from memory instead: same
// arithmetic.  STEPS wave steps of 64 / LPP rows per wave make a pass of RPP rows.
template <int LPP, int STEPS>
__global__ __launch_bounds__(1024) void ski_chunk_gather_kernel(const int4 *__restrict__ winfo, const float *__restrict__ uloc,
                                                                const float *__restrict__ H, const float *__restrict__ V,
                                                                float *__restrict__ out, long long N, int J, int G, int T, int CH,
                                                                float scale, float noise, int HCAP, int RPA) {
  constexpr int P = 4 * LPP;
  constexpr int SLOTS = 64 / LPP;              // rows per wave step
  constexpr int RPP = 16 * SLOTS * STEPS;      // rows per pass
  extern __shared__ float4 lds4[];
  float *Hs = reinterpret_cast<float *>(lds4);                        // [HCAP][P]  window rows of all projections
  float *Us = Hs + (size_t)HCAP * P;                                  // [J][RPA]   grid coordinates of the pass's rows
  __shared__ int4 swi[kChunkMaxJ];
  const int chunk = blockIdx.x;
  const long long base = (long long)chunk * CH;
  const int n = (int)((N - base) < CH ? (N - base) : CH);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPP, q = lane - slot * LPP;
  const unsigned magic = ((1u << 20) + T - 1) / T;
  if ((int)threadIdx.x < J) swi[threadIdx.x] = winfo[(size_t)chunk * J + threadIdx.x];
  __syncthreads();
  int hrow[kChunkMaxJ], wlo[kChunkMaxJ], rows = 0;      // (uniform: kept in scalar registers)
#pragma unroll
  for (int j = 0; j < kChunkMaxJ; ++j) {
    hrow[j] = rows;
    wlo[j] = __builtin_amdgcn_readfirstlane(j < J ? swi[j].x : 0);
    rows += __builtin_amdgcn_readfirstlane(j < J ? swi[j].w : 0);
  }
  const bool resident = rows <= HCAP;          // (uniform over the workgroup)
  const float *Vc = noise != 0.f ? V + base * T : H;      // (only element 0 is touched when noise == 0)
  float *outc = out + base * T;
  for (int p0 = 0; p0 < n; p0 += RPP) {
    const int np = n - p0 < RPP ? n - p0 : RPP;
    if (p0 > 0) __syncthreads();               // (the previous pass's readers of the coordinates are done)
    // ---- requests: the rows' grid coordinates and the first batch of window rows -------------------------------------------
    constexpr int HB = 8;                      // window floats per thread and batch
    float ru[kChunkMaxJ][2];
#pragma unroll
    for (int j = 0; j < kChunkMaxJ; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int p = threadIdx.x + h * 1024;
        ru[j][h] = uloc[((size_t)chunk * J + (j < J ? j : 0)) * CH + p0 + (p < np ? p : 0)];
      }
    const unsigned etot = (p0 == 0 && resident) ? (unsigned)rows * T : 0u;      // the windows serve every pass
    unsigned eoff[kChunkMaxJ];
    const float *hsrc[kChunkMaxJ];
#pragma unroll
    for (int j = 0; j < kChunkMaxJ; ++j) {
      eoff[j] = (unsigned)hrow[j] * T;
      hsrc[j] = H + ((size_t)(j < J ? j : 0) * G + wlo[j]) * T;
    }
    auto window_of = [&](unsigned e) {
      int j = 0;
#pragma unroll
      for (int t = 1; t < kChunkMaxJ; ++t) j += (t < J && e >= eoff[t]) ? 1 : 0;
      return j;
    };
    float hq[HB];
#pragma unroll
    for (int k = 0; k < HB; ++k) {
      const unsigned e = threadIdx.x + k * 1024;
      const unsigned ec = e < etot ? e : 0;
      const int j = window_of(ec);
      hq[k] = hsrc[j][ec - eoff[j]];
    }
    // ---- into LDS ---------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < kChunkMaxJ; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int p = threadIdx.x + h * 1024;
        if (j < J && p < np) Us[(size_t)j * RPA + p] = ru[j][h];      // (np, RPA: rows of the pass / coordinate slots)
      }
    auto window_store = [&](unsigned e, float x) {
      const int j = window_of(e);
      const unsigned rel = e - eoff[j], row = div_small(rel, magic);
      Hs[((unsigned)hrow[j] + row) * P + (rel - row * T)] = x;
    };
#pragma unroll
    for (int k = 0; k < HB; ++k) {
      const unsigned e = threadIdx.x + k * 1024;
      if (e < etot) window_store(e, hq[k]);
    }
    for (unsigned e = threadIdx.x + HB * 1024; e < etot; e += 1024) {      // (windows beyond 8192 floats: rare)
      const int j = window_of(e);
      window_store(e, hsrc[j][e - eoff[j]]);
    }
    if (T < P && etot > 0)
      for (unsigned r = threadIdx.x; r < (unsigned)rows; r += 1024)
        for (int c = T; c < P; ++c) Hs[r * P + c] = 0.f;
    // the rows' V values are requested two steps ahead of their use (a ring of three: all twenty at once cost the registers
    // the row sums need)
    float vring[3][4];
    auto load_v = [&](int s, float (&vin)[4]) {
      const int p = (s * 16 + wave) * SLOTS + slot;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 4 * q + r;
        const bool ok = s < STEPS && slot < SLOTS && p < np && col < T && noise != 0.f;
        const float x = Vc[ok ? (unsigned)(p0 + p) * T + col : 0u];      // (32-bit offsets from the chunk's first row)
        vin[r] = ok ? x : 0.f;
      }
    };
    load_v(0, vring[0]);
    load_v(1, vring[1]);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      load_v(s + 2, vring[(s + 2) % 3]);
      const int p = (s * 16 + wave) * SLOTS + slot;
      if (slot < SLOTS && p < np) {
        double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int j = 0; j < J; ++j) {
          float w[4], dw[4];
          const int idx0 = ski_taps_u<false>(Us[(size_t)j * RPA + p], 0.f, G, w, dw);
          float a0[4], a1[4], a2[4], a3[4];
          if (resident) {
            int hj = hrow[0] - wlo[0];         // (row of Hs of grid row 0 of projection j)
#pragma unroll
            for (int t = 1; t < kChunkMaxJ; ++t) hj = j == t ? hrow[t] - wlo[t] : hj;
            const float *hp = Hs + (unsigned)(idx0 + hj) * P + 4 * q;
            const float4 h0 = *reinterpret_cast<const float4 *>(hp);
            const float4 h1 = *reinterpret_cast<const float4 *>(hp + P);
            const float4 h2 = *reinterpret_cast<const float4 *>(hp + 2 * P);
            const float4 h3 = *reinterpret_cast<const float4 *>(hp + 3 * P);
            a0[0] = h0.x; a0[1] = h0.y; a0[2] = h0.z; a0[3] = h0.w;
            a1[0] = h1.x; a1[1] = h1.y; a1[2] = h1.z; a1[3] = h1.w;
            a2[0] = h2.x; a2[1] = h2.y; a2[2] = h2.z; a2[3] = h2.w;
            a3[0] = h3.x; a3[1] = h3.y; a3[2] = h3.z; a3[3] = h3.w;
          } else {
            const float *hp = H + ((size_t)j * G + idx0) * T;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int col = 4 * q + r < T ? 4 * q + r : 0;
              a0[r] = hp[col];
              a1[r] = hp[T + col];
              a2[r] = hp[2 * T + col];
              a3[r] = hp[3 * T + col];
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr = w[0] * a0[r];
            pr = __builtin_fmaf(w[1], a1[r], pr);
            pr = __builtin_fmaf(w[2], a2[r], pr);
            pr = __builtin_fmaf(w[3], a3[r], pr);
            a[r] += (double)pr;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = 4 * q + r;
          if (col < T) outc[(unsigned)(p0 + p) * T + col] = __builtin_fmaf(noise, vring[s % 3][r], scale * (float)a[r]);
        }
      }
    }
  }
}

// LDS of the scatter with `njb` projections resident
inline size_t chunk_scatter_lds(int CH, int G, int LPP, int njb) {
  return (size_t)CH * LPP * 16 + (size_t)njb * ((size_t)CH * 16 + (size_t)CH * 2 + (size_t)(G + 2) * 2);
}
template <int LPP> struct GatherGeom { static constexpr int STEPS = LPP == 3 ? 5 : 2; };      // passes of 1680 / 1024 / 2048 rows

template <class K>
inline int big_lds(K kernel, size_t bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// hist[j][g][hoff .. hoff + T) from the chunk tables of the plan; `win`: the scatter slabs of the SKI workspace
int scatter_chunked(const PlanView &pv, const float *V, double *hist, float *win, long long N, int J, int G, int T, int HT,
                    int hoff, hipStream_t st) {
  const int LPP = (T + 3) / 4;
  int njb = J;
  while (njb > 1 && chunk_scatter_lds(pv.CH, G, LPP, njb) > kChunkLdsBytes) --njb;
  const size_t lds = chunk_scatter_lds(pv.CH, G, LPP, njb);
  if (lds > kChunkLdsBytes) return RPGP_EWORKSPACE;
  static bool attr_set = false;
  if (!attr_set) {
    if (big_lds(ski_chunk_scatter_kernel<1>, kChunkLdsBytes) || big_lds(ski_chunk_scatter_kernel<2>, kChunkLdsBytes) ||
        big_lds(ski_chunk_scatter_kernel<3>, kChunkLdsBytes))
      return RPGP_EWORKSPACE;
    attr_set = true;
  }
  const dim3 cgrid((unsigned)((G + 15) / 16), (unsigned)J);
#define RPGP_CHUNK_SCATTER(L_)                                                                                                 \
  hipLaunchKernelGGL((ski_chunk_scatter_kernel<L_>), dim3((unsigned)pv.nch), dim3(1024), lds, st, pv.winfo, pv.uloc, pv.lperm,  \
                     pv.coff, V, win, N, J, G, T, pv.CH, njb);                                                                            \
  hipLaunchKernelGGL((ski_chunk_combine_kernel<L_>), cgrid, dim3(512), 0, st, pv.winfo, win, hist, pv.nch, J, G, T, HT, hoff)
  if (LPP == 1) {
    RPGP_CHUNK_SCATTER(1);
  } else if (LPP == 2) {
    RPGP_CHUNK_SCATTER(2);
  } else {
    RPGP_CHUNK_SCATTER(3);
  }
#undef RPGP_CHUNK_SCATTER
  return launch_status();
}

int gather_chunked(const PlanView &pv, const float *H, const float *V, float *out, long long N, int J, int G, int T, float scale,
                   float noise, hipStream_t st) {
  const int LPP = (T + 3) / 4;
  const int steps = LPP == 3 ? GatherGeom<3>::STEPS : (LPP == 2 ? GatherGeom<2>::STEPS : GatherGeom<1>::STEPS);
  const int rpp = 16 * (64 / LPP) * steps;
  const int rpa = pv.CH < rpp ? pv.CH : rpp;                         // coordinate slots per projection (rows of a pass)
  const size_t coord = (((size_t)J * rpa * 4) + 15) & ~(size_t)15;
  long long hcap = (long long)((kChunkLdsBytes - coord) / ((size_t)LPP * 16));     // window rows that fit beside the coordinates
  if (hcap > (long long)J * G) hcap = (long long)J * G;
  if (hcap < 64) return RPGP_EWORKSPACE;
  const size_t lds = (size_t)hcap * LPP * 16 + coord;
  static bool attr_set = false;
  if (!attr_set) {
    if (big_lds(ski_chunk_gather_kernel<1, GatherGeom<1>::STEPS>, kChunkLdsBytes) ||
        big_lds(ski_chunk_gather_kernel<2, GatherGeom<2>::STEPS>, kChunkLdsBytes) ||
        big_lds(ski_chunk_gather_kernel<3, GatherGeom<3>::STEPS>, kChunkLdsBytes))
      return RPGP_EWORKSPACE;
    attr_set = true;
  }
#define RPGP_CHUNK_GATHER(L_)                                                                                                  \
  hipLaunchKernelGGL((ski_chunk_gather_kernel<L_, GatherGeom<L_>::STEPS>), dim3((unsigned)pv.nch), dim3(1024), lds, st, pv.winfo, \
                     pv.uloc, H, V, out, N, J, G, T, pv.CH, scale, noise, (int)hcap, rpa)
  if (LPP == 1) {
    RPGP_CHUNK_GATHER(1);
  } else if (LPP == 2) {
    RPGP_CHUNK_GATHER(2);
  } else {
    RPGP_CHUNK_GATHER(3);
  }
#undef RPGP_CHUNK_GATHER
  return launch_status();
}

constexpr size_t kGatherLdsMax = 150 * 1024;

struct GatherPassA {                         // the executor's pass A folded into the gather (see ski_gather_lds_kernel<.., PASSA>)
  const float *L;
  int K;
  float *partA;
  int nparts;                                // out: slabs written (0: not folded)
};
int gather_planned(const PlanView *pv, const float *Z, const float *gp, const float *H, const float *V, float *out, long long M,
                   int ldz, int J, int G, int T, float scale, float noise, hipStream_t st, GatherPassA *pa = nullptr) {
  if (pa) pa->nparts = 0;
  if (pv && pv->nch > 0 && chunk_env_on() && T <= 12 && (V || noise == 0.f) && chunked_plans().has(pv->base))
    return gather_chunked(*pv, H, V, out, M, J, G, T, scale, noise, st);
  const size_t lds = (size_t)J * G * T * sizeof(float);
  static const int mode = [] { const char *e = getenv("RPGP_SKI_GATHER"); return e ? atoi(e) : 0; }();   // 3: never the LDS form
  if (T > 1 && T <= 12 && lds <= kGatherLdsMax && M >= 32768 && mode != 3 && ((size_t)J * G * T) % 4 == 0) {
    static bool attr_set = false;
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
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (pa && T > 8 && noise != 0.f && V && pa->K >= 0 && pa->K <= rpgp_internal::kCgMaxK &&
        lds >= (size_t)16 * 17 * 16 * sizeof(float)) {
      static bool pa_attr = false;
      if (!pa_attr)
        pa_attr = hipFuncSetAttribute(reinterpret_cast<const void *>(ski_gather_lds_kernel<3, 1, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGatherLdsMax) == hipSuccess;
      if (pa_attr) {
        hipLaunchKernelGGL((ski_gather_lds_kernel<3, 1, true>), dim3((unsigned)cus), dim3(1024), lds, st, Z, gp, H, V, out, M, ldz, J,
                           G, T, scale, noise, pa->L, pa->K, pa->partA);
        pa->nparts = cus;
        return launch_status();
      }
    }
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

// hist[j][g][hoff .. hoff + T) (row stride HT, float64) from the plan; `partial` holds max_items * 4 * 12 floats
// `cells_tt` != nullptr: the caller's Toeplitz stage forms the histogram from the tap records itself — when the whole block is
// ONE column piece of the per-cell form, only the scatter runs and *cells_tt = the floats per tap row of `partial`; else 0.
int scatter_planned(const PlanView &pv, const float *V, double *hist, float *partial, long long N, int J, int G, int T, int HT,
                    int hoff, hipStream_t st, int *cells_tt = nullptr) {
  if (cells_tt) *cells_tt = 0;
  if (pv.nch > 0 && chunk_env_on() && T <= 12 && chunked_plans().has(pv.base))
    return scatter_chunked(pv, V, hist, partial, N, J, G, T, HT, hoff, st);
  const int cells = J * G;
  const long long items = max_items(N, J, G);
  const unsigned nb = (unsigned)items;
  // one workgroup per cell + four shifted reads (round 4), or the item form + item-walking cell sums (RPGP_SKI_SCATTER=item)
  static const bool by_cell = [] {
    const char *e = getenv("RPGP_SKI_SCATTER");
    return !(e && e[0] == 'i');
  }();
  for (int t0 = 0; t0 < T;) {
    const int tt = ski_tpiece(T - t0);
    const int tcnt = (T - t0 < tt) ? T - t0 : tt;
    const long long n = (long long)J * G * tt;
    if (by_cell) {
      const char *env_co = getenv("RPGP_SKI_CELL_ORDER");          // =0: storage order (read per call: A/B)
      const int go = (env_co && env_co[0] == '0') ? 0 : G;
      if (tt == 1)
        hipLaunchKernelGGL((ski_scatter_cell_kernel<1, 1>), dim3((unsigned)cells), dim3(256), 0, st, pv.rec, pv.cell_start, V, partial, T, t0, tcnt, go);
      else if (tt == 4)
        hipLaunchKernelGGL((ski_scatter_cell_kernel<4, 1>), dim3((unsigned)cells), dim3(256), 0, st, pv.rec, pv.cell_start, V, partial, T, t0, tcnt, go);
      else
        hipLaunchKernelGGL((ski_scatter_cell_kernel<4, 3>), dim3((unsigned)cells), dim3(256), 0, st, pv.rec, pv.cell_start, V, partial, T, t0, tcnt, go);
      int rc = launch_status();
      if (rc) return rc;
      if (cells_tt && t0 == 0 && tcnt == T) {              // one piece: the tap records go to the Toeplitz stage as they are
        *cells_tt = tt;
        return 0;
      }
      hipLaunchKernelGGL(ski_cellsum4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, hist, J, G, tt, tcnt,
                         HT, hoff + t0);
      rc = launch_status();
      if (rc) return rc;
      t0 += tcnt;
      continue;
    }
    if (tt == 1)
      hipLaunchKernelGGL((ski_scatter_sorted_kernel<1, 1>), dim3(nb), dim3(256), 0, st, pv.rec, pv.item_info, pv.item_start, cells, V, partial, T, t0, tcnt);
    else if (tt == 4)
      hipLaunchKernelGGL((ski_scatter_sorted_kernel<4, 1>), dim3(nb), dim3(256), 0, st, pv.rec, pv.item_info, pv.item_start, cells, V, partial, T, t0, tcnt);
    else
      hipLaunchKernelGGL((ski_scatter_sorted_kernel<4, 3>), dim3(nb), dim3(256), 0, st, pv.rec, pv.item_info, pv.item_start, cells, V, partial, T, t0, tcnt);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(ski_cellsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, pv.item_start, hist, J,
                       G, tt, tcnt, HT, hoff + t0);
    rc = launch_status();
    if (rc) return rc;
    t0 += tcnt;
  }
  return 0;
}

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
  hipLaunchKernelGGL(plan_items_kernel, dim3(1), dim3(1024), 0, st, pv.cell_start, cells, pv.item_start, pv.item_info);
  hipLaunchKernelGGL(plan_finish_kernel, dim3((unsigned)nblk), dim3(256), 0, st, keys_out, vals_out, pv.fnat, (long long)N, J, G,
                     (long long)nj, pv.rec);
  const bool chunked = pv.nch > 0 && chunk_env_on();
  chunked_plans().mark(plan, chunked);         // (a blob reused for a plan without the tables must not look chunked)
  if (chunked) {                               // chunk tables of the round-5 product (its kernels read nothing of the above)
    hipLaunchKernelGGL(chunk_plan_kernel, dim3((unsigned)pv.nch, (unsigned)J), dim3(256), 0, st, Z, grid_params, (long long)N, ldz,
                       J, G, pv.CH, pv.winfo, pv.uloc, pv.lperm, pv.coff);
    hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(1024), 0, st, pv.winfo, pv.nch * J);
  }
  return launch_status();
}

int rpgp_ski_chunk_mode(int mode) {
  const int prev = chunk_mode_ref();
  if (mode == 0 || mode == 1) chunk_mode_ref() = mode;
  return prev;
}

int rpgp_ski_plan_is_chunked(const void *plan) { return plan && chunked_plans().has(plan) ? 1 : 0; }

int rpgp_ski_scatter_planned(const void *plan, const float *V, double *hist, int64_t N, int J, int G, int T, void *workspace,
                             size_t workspace_bytes, void *stream) {
  if (!plan || !V || !hist || !plan_args_ok(N, J, G) || T <= 0 || T > 12) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  if ((size_t)max_items(N, J, G) * 48 > rpgp_internal::ski_scratch_floats(J, G)) return RPGP_EWORKSPACE;
  const PlanView pv = plan_view(const_cast<void *>(plan), N, J, G);
  float *partial = reinterpret_cast<float *>(workspace) + rpgp_internal::ski_scratch_offset_floats(J, G, T);
  return scatter_planned(pv, V, hist, partial, N, J, G, T, T, 0, reinterpret_cast<hipStream_t>(stream));
}

int rpgp_ski_bilinear_scatter_planned(const void *plan, const float *L, const float *R, double *hist2, int64_t N, int J, int G,
                                      int T, void *workspace, size_t workspace_bytes, void *stream) {
  if (!plan || !L || !R || !hist2 || !plan_args_ok(N, J, G) || T <= 0 || T > 12) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  if ((size_t)max_items(N, J, G) * 48 > rpgp_internal::ski_scratch_floats(J, G)) return RPGP_EWORKSPACE;
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
  if ((size_t)max_items(N, J, G) * 48 > rpgp_internal::ski_scratch_floats(J, G)) return RPGP_EWORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const PlanView pv = plan_view(const_cast<void *>(plan), N, J, G);
  double *hist = reinterpret_cast<double *>(workspace);
  float *H = reinterpret_cast<float *>(workspace) + 2 * (size_t)J * G * (2 * T);
  float *partial = reinterpret_cast<float *>(workspace) + rpgp_internal::ski_scratch_offset_floats(J, G, T);
  // RPGP_SKI_CELLSUM=0 (opt-in): the four-tap cell sums formed in the Toeplitz stage's operand load instead of a histogram pass
  // of their own (same bits).  Measured at the C5 shape ON THE DEVICE the folded Toeplitz stage takes 16.6 us against 8.1 + 4.8
  // (T = 11) and 8.1 + ~3 (T = 1): every one of the 64 row tiles of a projection repeats the sums of the whole histogram.  Through
  // the Python wrappers the T = 1 product is faster folded (49 -> 33 us: three launches instead of four on a host-bound path),
  // inside the executor — where the products of a solve run — it is not; the default keeps the histogram pass.
  static const bool fold_cells = [] { const char *e = getenv("RPGP_SKI_CELLSUM"); return e && e[0] == '0'; }();
  int cells_tt = 0;
  const bool fold = fold_cells && rpgp_internal::ski_toeplitz_takes_cells(J, G, T);
  int rc = scatter_planned(pv, V, hist, partial, N, J, G, T, T, 0, st, fold ? &cells_tt : nullptr);
  if (rc) return rc;
  rc = cells_tt ? rpgp_internal::ski_toeplitz_cells_launch(partial, cells_tt, grid_params, H, J, G, T, st, pv.tcol)
                : rpgp_internal::ski_toeplitz_launch(hist, 1, grid_params, H, J, G, T, st, pv.tcol);
  if (rc) return rc;
  return gather_planned(&pv, Z, grid_params, H, V, out, N, ldz, J, G, T, scale, noise, st);
}

}  // extern "C"

namespace rpgp_internal {
int ski_mvm_planned_passa(const void *plan, const float *Z, const float *grid_params, const float *V, float *out, long long N,
                          int ldz, int J, int G, int T, float scale, float noise, void *workspace, size_t workspace_bytes,
                          hipStream_t st, const float *L, int K, float *partA, int *nparts) {
  *nparts = 0;
  if (!plan || !Z || !grid_params || !V || !out || !plan_args_ok(N, J, G) || T <= 0 || T > 12 || ldz < J) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_ski_workspace_bytes(J, G, T)) return RPGP_EWORKSPACE;
  if ((size_t)max_items(N, J, G) * 48 > ski_scratch_floats(J, G)) return RPGP_EWORKSPACE;
  const PlanView pv = plan_view(const_cast<void *>(plan), N, J, G);
  double *hist = reinterpret_cast<double *>(workspace);
  float *H = reinterpret_cast<float *>(workspace) + 2 * (size_t)J * G * (2 * T);
  float *partial = reinterpret_cast<float *>(workspace) + ski_scratch_offset_floats(J, G, T);
  int cells_tt = 0;
  int rc = scatter_planned(pv, V, hist, partial, N, J, G, T, T, 0, st, nullptr);      // (wide blocks: the histogram pass pays)
  if (rc) return rc;
  rc = cells_tt ? ski_toeplitz_cells_launch(partial, cells_tt, grid_params, H, J, G, T, st, pv.tcol)
                : ski_toeplitz_launch(hist, 1, grid_params, H, J, G, T, st, pv.tcol);
  if (rc) return rc;
  GatherPassA pa{L, K, partA, 0};
  rc = gather_planned(&pv, Z, grid_params, H, V, out, N, ldz, J, G, T, scale, noise, st, (L || K == 0) ? &pa : nullptr);
  *nparts = pa.nparts;
  return rc;
}
}  // namespace rpgp_internal

extern "C" {

int rpgp_ski_gather_fast(const void *plan, const float *Z, const float *grid_params, const float *H, const float *V, float *out,
                         int64_t M, int ldz, int J, int G, int T, float scale, float noise, void *stream) {
  if (!Z || !grid_params || !H || !out || M <= 0 || J <= 0 || G < 8 || T <= 0 || T > 12 || ldz < J) return RPGP_EINVAL;
  if (noise != 0.f && !V) return RPGP_EINVAL;
  if (plan && plan_args_ok(M, J, G)) {
    const PlanView pv = plan_view(const_cast<void *>(plan), M, J, G);
    return gather_planned(&pv, Z, grid_params, H, V, out, (long long)M, ldz, J, G, T, scale, noise,
                          reinterpret_cast<hipStream_t>(stream));
  }
  return gather_planned(nullptr, Z, grid_params, H, V, out, (long long)M, ldz, J, G, T, scale, noise,
                        reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
