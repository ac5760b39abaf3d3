// rpgp_cg.hip — native mBCG executor: preconditioned batched conjugate gradients with Lanczos coefficients
//  as ONE host call, on one GPU or sharded over ranks.
//
// Per iteration (T <= 16 right-hand sides, row-major N x T; Woodbury preconditioner M = L L^T + sigma^2 I, K <= 16):
//   Ap = A p                         operator kernels (fused / prepared / SKI / cached-K / family)      [+ all-reduce]
//   k_pass_a : partial p.Ap per column and partial L^T(Ap)                                  reads p, Ap, L
//   k_reduce : fixed-order float64 sums of the per-workgroup partials                       [+ all-reduce, row mode]
//   k_pass_b : alpha = rz / pAp ; x += alpha p ; r -= alpha Ap ;
//              w = L^T r_old - alpha L^T(Ap) ; z = (r - L Cinv w) / sigma^2 ;
//              partial |r|^2, r.z and (directly, from the r just formed) L^T r              reads p, Ap, x, r, L
//   k_reduce                                                                                [+ all-reduce, row mode]
//   k_pass_c : beta = rz' / rz ; p = z + beta p ; history ; convergence decision            reads z, p
// Three streaming passes instead of five (round 2: k_coldot, k_update, k_wsolve, k_precond, k_direction).  The z of
// iteration k needs L^T r_k, a global reduction over the r_k that the same pass is still writing; it is formed from the
// DIRECT L^T r_{k-1} of the previous pass and the L^T(Ap) partials of pass A — one recurrence step from a freshly
// computed value, never an accumulated one.
//
// Data layout of the passes: a workgroup takes 256-row tiles, a wave 16-row blocks; lane l of the wave holds, for
// r = 0..3, element (row R0 + l/16 + 4r, column l%16).  That is (a) one contiguous 4-row segment per load instruction
// (the round-2 "thread = row" layout touched 22 cache lines per instruction at T = 11 and ran at 0.22 of HBM peak),
// (b) a fixed column per lane, so alpha / beta and the column sums are registers, (c) the B-operand layout of
// v_mfma_f32_16x16x4_f32, so L^T X (16 x 16 per block) is four exact-fp32 matrix instructions per block, and (d) the
// result layout of v_mfma_f64_16x16x4_f64, so the Woodbury correction L (Cinv w) is four float64 matrix instructions
// and the cancelling subtraction r - L (Cinv w) stays float64 (it shrinks the range-of-L part of r by ~1e-6).
//
// Sharding (struct rpgp_reducer): RPGP_SHARD_PARTIAL — vectors replicated, the operator returns this rank's partial
// product, summed by ONE all-reduce hook call on the launch stream per iteration (pair- / J-sharded exact operator,
// pair-sharded packed cache); RPGP_SHARD_ROWS — the SKI operator on this rank's rows: the J x G x T float64 histogram and
// the two small reduction vectors per iteration are all-reduced.  The hook only enqueues; the convergence flag stays on
// the device and the host still polls it one iteration late, so a sharded solve has no host synchronisation per iteration.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/rpgp.h"
#include "rpgp_internal.h"

namespace {

constexpr int kMaxT = 16;
constexpr int kMaxK = 16;          // preconditioner rank
constexpr int kMaxBlocks = 1024;   // workgroups of the streaming passes (= partial-sum slabs)
constexpr int kMaxHist = 64;       // Lanczos coefficients kept for at most this many iterations
constexpr int kRedW = 288;         // reduction vector: [16 col sums a][16 col sums b][16 x 16 L^T X]
constexpr int kRedLt = 32;         // offset of the L^T X block
static_assert(kRedW == rpgp_internal::kCgRedW && kRedLt == rpgp_internal::kCgRedLt && kMaxK == rpgp_internal::kCgMaxK,
              "slab format constants of rpgp_internal.h");

typedef float floatx4m __attribute__((ext_vector_type(4)));
typedef double doublex4m __attribute__((ext_vector_type(4)));

// launch STMT with TT = the exact number of right-hand sides (1 .. kMaxT)
#define CG_DISPATCH_T(T_, STMT)                                                                          \
  switch (T_) {                                                                                          \
    case 1: { constexpr int TT = 1; STMT; } break;    case 2: { constexpr int TT = 2; STMT; } break;    \
    case 3: { constexpr int TT = 3; STMT; } break;    case 4: { constexpr int TT = 4; STMT; } break;    \
    case 5: { constexpr int TT = 5; STMT; } break;    case 6: { constexpr int TT = 6; STMT; } break;    \
    case 7: { constexpr int TT = 7; STMT; } break;    case 8: { constexpr int TT = 8; STMT; } break;    \
    case 9: { constexpr int TT = 9; STMT; } break;    case 10: { constexpr int TT = 10; STMT; } break;  \
    case 11: { constexpr int TT = 11; STMT; } break;  case 12: { constexpr int TT = 12; STMT; } break;  \
    case 13: { constexpr int TT = 13; STMT; } break;  case 14: { constexpr int TT = 14; STMT; } break;  \
    case 15: { constexpr int TT = 15; STMT; } break;  default: { constexpr int TT = 16; STMT; } break;  \
  }

#define CG_CHECK(expr)                            \
  do {                                            \
    hipError_t _e = (expr);                       \
    if (_e != hipSuccess) return (int)_e;         \
  } while (0)

// device-resident scalar state
struct CgPoll {             // copied to pinned host memory after the iterations that test convergence
  float mean_resid;
  int done;                 // 0 running, 1 tolerance reached (later iterations are no-ops), 2 non-finite residual,
                            // 3 stagnated (no 1 % improvement of the best residual over `stagnation_window` tests)
  int iters;                // iterations performed when `done` was set
  float best_resid;         // stagnation bookkeeping
  int since_best;
  int snap_slot;            // which snap_resid entry is current
  float snap_cur;           // its value (mean residual of the iterate saved in x_best)
  int seq;                  // host copies only: stamped LAST (the host spins on it, see wait_record)
};
// the record a kernel hands to the host: the fields, a system-scope fence, then the stamp
__device__ __forceinline__ void publish_poll(CgPoll *__restrict__ host, const CgPoll &v, int stamp) {
  host->mean_resid = v.mean_resid;
  host->done = v.done;
  host->iters = v.iters;
  host->best_resid = v.best_resid;
  host->since_best = v.since_best;
  host->snap_slot = v.snap_slot;
  host->snap_cur = v.snap_cur;
  __threadfence_system();
  *reinterpret_cast<volatile int *>(&host->seq) = stamp;
  __threadfence_system();
}
struct CgState {
  float rz[2][kMaxT];       // r.z of the current iterate, ping-pong by iteration parity (no intra-kernel race)
  float rhs_norm[kMaxT];
  float resid[kMaxT];       // residual norms of the current iterate
  int rhs_zero[kMaxT];
  float snap_resid[2];      // mean residual of the iterate saved in x_best (ping-pong by iteration parity)
  int done_pp[2];           // poll.done as pass C reads it at its head (ping-pong: workgroup 0 sets poll.done at its tail while
                            // late workgroups of the same launch may only just be starting)
  CgPoll poll;
};
// ---- the lane <-> element map of the streaming passes -------------------------------------------------------------
// wave `wave` of the workgroup, 16-row block `blk` (0..3) of the 256-row tile starting at row0:
//   rows R0 .. R0 + 15 with R0 = row0 + 64 wave + 16 blk; lane (c = l % 16, q = l / 16), register r: row R0 + q + 4 r.
struct Lane {
  int c, q, wave;
  __device__ __forceinline__ Lane() : c(threadIdx.x & 15), q((threadIdx.x & 63) >> 4), wave(threadIdx.x >> 6) {}
};

// 32-bit element offsets (N * 16 < 2^31, checked by the host): the address of every load is base + 32-bit offset + constant;
// with 64-bit row arithmetic the passes spent more issue slots on addresses than on data (PMC, round 3: 1700 VALU + 700
// SALU instructions per wave for ~290 memory instructions; SQ_WAIT_INST_ANY 46 % of the wave cycles).  FULL = the whole
// 256-row tile is inside the vector: no row tests, and lanes whose column / rank index is out of range read a clamped,
// valid element instead of being masked — what they compute lands in slots nobody reads (column sums and L^T X entries
// >= T or >= K; the matching Cinv / tv entries are zero) and their stores are suppressed.
template <int TT, bool FULL>
__device__ __forceinline__ void load_block(const float *__restrict__ a, unsigned R0, unsigned N, const Lane &ln,
                                           float (&v)[4]) {
  const unsigned cc = ln.c < TT ? ln.c : TT - 1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned row = R0 + ln.q + 4 * r;
    if constexpr (FULL) {
      v[r] = a[row * TT + cc];
    } else {
      const bool ok = row < N && ln.c < TT;
      const float x = a[ok ? row * TT + cc : 0];          // unconditional load of a clamped address
      v[r] = ok ? x : 0.f;
    }
  }
}

template <int TT, bool FULL>
__device__ __forceinline__ void store_block(float *__restrict__ a, unsigned R0, unsigned N, const Lane &ln,
                                            const float (&v)[4]) {
  if (ln.c < TT) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned row = R0 + ln.q + 4 * r;
      if (FULL || row < N) a[row * TT + ln.c] = v[r];
    }
  }
}

// A operand of L^T X: lane (c = kk, q) holds L[R0 + q + 4 r][kk]
template <bool FULL>
__device__ __forceinline__ void load_L_rows(const float *__restrict__ L, unsigned R0, unsigned N, int K, const Lane &ln,
                                            float (&v)[4]) {
  const unsigned cc = ln.c < K ? ln.c : K - 1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned row = R0 + ln.q + 4 * r;
    if constexpr (FULL) {
      v[r] = L[row * (unsigned)K + cc];
    } else {
      const bool ok = row < N && ln.c < K;
      const float x = L[ok ? row * (unsigned)K + cc : 0];
      v[r] = ok ? x : 0.f;
    }
  }
}

// A operand of L (Cinv w) in float64: lane (c = row in block, q) holds L[R0 + c][4 s + q]
template <bool FULL>
__device__ __forceinline__ void load_L_cols(const float *__restrict__ L, unsigned R0, unsigned N, int K, const Lane &ln,
                                            float (&v)[4]) {
  const unsigned row = R0 + ln.c;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int b = 4 * s + ln.q;
    const unsigned bc = b < K ? b : K - 1;
    if constexpr (FULL) {
      v[s] = L[row * (unsigned)K + bc];
    } else {
      const bool ok = row < N && b < K;
      const float x = L[ok ? row * (unsigned)K + bc : 0];
      v[s] = ok ? x : 0.f;
    }
  }
}

// column sum over the workgroup of a per-lane partial (fixed order): every thread gets nothing; thread c < 16 of wave 0
// writes dst[c].  `sh` holds 64 floats.  Ends with a barrier.
__device__ __forceinline__ void block_colsum(float v, float *__restrict__ sh, float *__restrict__ dst, const Lane &ln) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  __syncthreads();
  if (ln.q == 0) sh[ln.wave * 16 + ln.c] = v;
  __syncthreads();
  if (threadIdx.x < 16) dst[threadIdx.x] = ((sh[threadIdx.x] + sh[16 + threadIdx.x]) + sh[32 + threadIdx.x]) + sh[48 + threadIdx.x];
}

// the four waves' 16 x 16 L^T X accumulators (fp32 MFMA result layout: lane reg r = D[4 q + r][c]) added in a fixed order
// into dst[kk * 16 + t].  `sh` holds 4 * 256 floats.  Ends with a barrier.
__device__ __forceinline__ void block_ltsum(const floatx4m &acc, float *__restrict__ sh, float *__restrict__ dst,
                                            const Lane &ln) {
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) sh[ln.wave * 256 + (4 * ln.q + r) * 16 + ln.c] = acc[r];
  __syncthreads();
  const int e = threadIdx.x;
  dst[e] = ((sh[e] + sh[256 + e]) + sh[512 + e]) + sh[768 + e];
}

// ---- the reduction on the CONSUMER side ------------------------------------------------------------------------------
// With few partial slabs (N <= 16k: at most kDirectParts workgroups per pass: 29 at C2, 59 at C3) a k_reduce launch between
// two passes is 5 us of kernel plus two launch boundaries, 45 times per optimiser step — a tenth of the step's device time.
// Here the pass that NEEDS a reduced entry adds the slabs up itself at its head: the same float64 additions in the same order
// as k_reduce (group g = slab g, then slab g + 32; the 32 group sums in order), so the numbers are bitwise those of the
// separate launch, every workgroup forms the same values, and nothing but the kernel boundary orders producer and consumer
// (no atomics, no fences: the round-4 "last workgroup folds" variant paid an agent-scope release per workgroup and was
// slower than the launch it replaced).  All loads of a batch are in flight together: one or two L2 round trips.
constexpr int kDirectParts = 64;
template <int NQ>
__device__ __forceinline__ void slab_sums(const float *const (&src)[NQ], const int (&np)[NQ], const int (&idx)[NQ],
                                          double (&out)[NQ]) {
  int nmax = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    out[q] = 0.0;
    nmax = np[q] > nmax ? np[q] : nmax;
  }
#pragma unroll 1
  for (int g0 = 0; g0 < 32 && g0 < nmax; g0 += 16) {
    float v[NQ][16], w[NQ][16];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        // (unconditional loads from clamped slabs, masked afterwards: a load under a condition becomes a branch and a wait
        //  of its own, and the 32 - 96 round trips of a batch then run one after the other)
        const int g = g0 + u, last = np[q] - 1;
        const int ga = g < last ? g : last, gb = g + 32 < last ? g + 32 : last;
        const float x = src[q][(unsigned)ga * (unsigned)kRedW + (unsigned)idx[q]];
        const float y = src[q][(unsigned)gb * (unsigned)kRedW + (unsigned)idx[q]];
        v[q][u] = g <= last ? x : 0.f;
        w[q][u] = g + 32 <= last ? y : 0.f;
      }
    // (nothing moves across: every load of the batch is issued before the first addition waits — left to itself the
    //  scheduler interleaves six loads at a time with their additions to save registers)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int u = 0; u < 16; ++u) out[q] += (0.0 + (double)v[q][u]) + (double)w[q][u];
  }
}

// ---- pass A: partial p.Ap per column, partial L^T (Ap) ------------------------------------------------------------
// part[blk][0..15] = sum over the workgroup's rows of a*b per column; part[blk][32 + kk*16 + t] = sum L[row][kk] b[row][t]
template <int TT>
__global__ __launch_bounds__(256) void k_pass_a(const float *__restrict__ a, const float *__restrict__ b,
                                                const float *__restrict__ L, float *__restrict__ part, long long N, int K) {
  __shared__ float sh[1024];
  const Lane ln;
  float dot = 0.f;
  // four independent accumulators (one per 16-row block): a single one is a chain of 16 DEPENDENT matrix instructions per
  // tile, and with 3 - 4 waves per SIMD nothing hides their latency (PMC: SQ_WAIT_INST_ANY = 40 % of the wave cycles)
  floatx4m lt4[4];
#pragma unroll
  for (int bk = 0; bk < 4; ++bk) lt4[bk] = floatx4m{0.f, 0.f, 0.f, 0.f};
  const unsigned Nu = (unsigned)N;
  auto do_tile = [&](unsigned tile, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const unsigned w0 = tile * 256u + 64u * ln.wave;
    float av[4][4], bv[4][4], lv[4][4];
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) {
      load_block<TT, FULL>(a, w0 + 16 * bk, Nu, ln, av[bk]);
      load_block<TT, FULL>(b, w0 + 16 * bk, Nu, ln, bv[bk]);
      if (K > 0) load_L_rows<FULL>(L, w0 + 16 * bk, Nu, K, ln, lv[bk]);
    }
#pragma unroll
    for (int bk = 0; bk < 4; ++bk)
#pragma unroll
      for (int r = 0; r < 4; ++r) dot = __builtin_fmaf(av[bk][r], bv[bk][r], dot);
    if (K > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int bk = 0; bk < 4; ++bk)
          lt4[bk] = __builtin_amdgcn_mfma_f32_16x16x4f32(lv[bk][r], bv[bk][r], lt4[bk], 0, 0, 0);
    }
  };
  const unsigned ntiles = (Nu + 255u) / 256u;
  for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if ((tile + 1u) * 256u <= Nu) do_tile(tile, std::true_type{});
    else do_tile(tile, std::false_type{});
  }
  const floatx4m lt = (lt4[0] + lt4[1]) + (lt4[2] + lt4[3]);
  float *dst = part + (size_t)blockIdx.x * kRedW;
  block_colsum(dot, sh, dst, ln);
  if (threadIdx.x < 16) dst[16 + threadIdx.x] = 0.f;
  block_ltsum(lt, sh, dst + kRedLt, ln);
}

// red[e] = sum over the slabs of part[.][e] in float64, fixed order: workgroup b owns entries 8 b .. 8 b + 7, its 32
// thread groups take every 32nd slab (8 independent loads in flight) and the 32 group sums are added in order.
__global__ __launch_bounds__(256) void k_reduce(const float *__restrict__ part, int nparts, double *__restrict__ red) {
  __shared__ double sh[256];
  const int e = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int idx = blockIdx.x * 8 + e;
  double s = 0.0;
  int p = g;
  // (round 6: ALL of a thread group's slabs requested before the first addition — 32 loads in flight at kMaxBlocks = 1 024
  //  slabs, one L2 round trip instead of four batches of eight; the additions keep their order: same bits)
  for (; p + 31 * 32 < nparts; p += 32 * 32) {
    float v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = part[(size_t)(p + 32 * u) * kRedW + idx];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 32; ++u) s += (double)v[u];
  }
  for (; p + 7 * 32 < nparts; p += 8 * 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(p + 32 * u) * kRedW + idx];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u];
  }
  for (; p < nparts; p += 32) s += (double)part[(size_t)p * kRedW + idx];
  sh[g * 8 + e] = s;
  __syncthreads();
  if (threadIdx.x < 8) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) tot += sh[q * 8 + threadIdx.x];
    red[idx] = tot;
  }
}

// ---- set-up: r = rhs / |rhs| (columns with |rhs| < 1e-10 are flagged zero and left unscaled), x = p = Ap = 0,
// ---- partial L^T r ------------------------------------------------------------------------------------------------
template <int TT, bool DIRECT = false>
__global__ __launch_bounds__(256) void k_init(const float *__restrict__ rhs, const double *__restrict__ red_sq,
                                              float *__restrict__ r, float *__restrict__ x, float *__restrict__ p,
                                              float *__restrict__ Ap, const float *__restrict__ L,
                                              float *__restrict__ part, CgState *__restrict__ st, long long N, int K,
                                              const float *__restrict__ dir_sq = nullptr, int n_sq = 0) {
  __shared__ float sh[1024];
  __shared__ float snrm[kMaxT];
  const Lane ln;
  // (dir_sq != nullptr: the |rhs|^2 sums are added up here from the slabs of the norm pass — see slab_sums)
  double q_sq;
  {
    const int tcl = (int)threadIdx.x < TT ? (int)threadIdx.x : TT - 1;
    if constexpr (DIRECT) {
      const float *const src[1] = {dir_sq};
      const int np[1] = {n_sq}, ix[1] = {tcl};
      double o[1];
      slab_sums<1>(src, np, ix, o);
      q_sq = o[0];
    } else {
      q_sq = red_sq[tcl];
    }
  }
  if (threadIdx.x < kMaxT) {
    float nrm = threadIdx.x < TT ? sqrtf((float)q_sq) : 1.0f;
    const int zero = nrm < 1e-10f;
    if (zero) nrm = 1.0f;
    snrm[threadIdx.x] = nrm;
    if (blockIdx.x == 0) {
      st->rhs_norm[threadIdx.x] = nrm;
      st->rhs_zero[threadIdx.x] = zero;
      st->resid[threadIdx.x] = 1.0f;
      st->rz[0][threadIdx.x] = 0.f;
      st->rz[1][threadIdx.x] = 0.f;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->poll.mean_resid = 1.0f;
    st->poll.done = 0;
    st->poll.iters = 0;
    st->poll.best_resid = 3.0e38f;
    st->poll.since_best = 0;
    st->poll.snap_slot = 0;
    st->poll.snap_cur = 3.0e38f;
    st->snap_resid[0] = 3.0e38f;
    st->snap_resid[1] = 3.0e38f;
    st->done_pp[0] = 0;
    st->done_pp[1] = 0;
  }
  __syncthreads();
  const float nrm = snrm[ln.c];
  floatx4m lt = {0.f, 0.f, 0.f, 0.f};
  const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
  const unsigned Nu = (unsigned)N;
  auto do_tile = [&](unsigned tile, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const unsigned w0 = tile * 256u + 64u * ln.wave;
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) {
      const unsigned R0 = w0 + 16 * bk;
      float rv[4], lv[4];
      load_block<TT, FULL>(rhs, R0, Nu, ln, rv);
      if (K > 0) load_L_rows<FULL>(L, R0, Nu, K, ln, lv);
#pragma unroll
      for (int q = 0; q < 4; ++q) rv[q] = rv[q] / nrm;
      store_block<TT, FULL>(r, R0, Nu, ln, rv);
      store_block<TT, FULL>(x, R0, Nu, ln, zero4);
      store_block<TT, FULL>(p, R0, Nu, ln, zero4);
      store_block<TT, FULL>(Ap, R0, Nu, ln, zero4);
      if (K > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) lt = __builtin_amdgcn_mfma_f32_16x16x4f32(lv[q], rv[q], lt, 0, 0, 0);
      }
    }
  };
  const unsigned ntiles = (Nu + 255u) / 256u;
  for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if ((tile + 1u) * 256u <= Nu) do_tile(tile, std::true_type{});
    else do_tile(tile, std::false_type{});
  }
  float *dst = part + (size_t)blockIdx.x * kRedW;
  __syncthreads();
  if (threadIdx.x < 32) dst[threadIdx.x] = 0.f;
  block_ltsum(lt, sh, dst + kRedLt, ln);
}

// ---- pass B -----------------------------------------------------------------------------------------------------------
// alpha = rz / pAp (guarded; 0 after convergence, so iterations the host had already enqueued leave x and r untouched);
// x += alpha p; r -= alpha Ap; z = M^-1 r (Woodbury, float64 correction); partial |r|^2, r.z, L^T r.
// first != 0: the set-up call (alpha = 0, w = L^T r0 as reduced by k_init).
template <int TT, int NB = 1, bool DIRECT = false>
__global__ __launch_bounds__(256) void k_pass_b(const float *__restrict__ p, const float *__restrict__ Ap,
                                                float *__restrict__ x, float *__restrict__ r, float *__restrict__ z,
                                                const float *__restrict__ L, const double *__restrict__ Cinv,
                                                const double *__restrict__ redA, const double *__restrict__ redB,
                                                float *__restrict__ part, const CgState *__restrict__ st,
                                                float *__restrict__ alpha_out, long long N, int K, float sigma2,
                                                float eps, float stop_after, int cur, int first,
                                                const float *__restrict__ dirA = nullptr, int nA = 0,
                                                const float *__restrict__ dirB = nullptr, int nB = 0) {
  __shared__ float sh[1024];
  __shared__ double sW[256];
  __shared__ double sTv[256];
  __shared__ float salpha[kMaxT];
  const Lane ln;
  // Every scalar this prologue needs is requested up front and pinned by one empty asm (hipcc otherwise sinks each load under
  // the condition that uses it — eight dependent round trips, about half of this kernel's time at small N).  (`redA` of
  // the set-up call holds the |rhs|^2 sums — valid memory, value unused; `Cinv` without a preconditioner is not a valid
  // pointer: its load stays behind that condition.)
  const int tcl = (int)threadIdx.x < TT ? (int)threadIdx.x : TT - 1;
  const float q_rz = st->rz[cur][tcl], q_res = st->resid[tcl];
  const int q_zero = st->rhs_zero[tcl], q_done = st->poll.done;
  double q_pap, q_wb, q_wa;
  if constexpr (DIRECT) {       // consumer-side reduction (slab_sums): pass A's slabs; L^T r of the previous pass B from the
    if (dirB) {                 // slabs of the set-up, afterwards from redB (written by workgroup 0 of the pass C in between)
      const float *const src[3] = {dirA, dirB, dirA};
      const int np[3] = {nA, nB, nA}, ix[3] = {tcl, kRedLt + (int)threadIdx.x, kRedLt + (int)threadIdx.x};
      double o[3];
      slab_sums<3>(src, np, ix, o);
      q_pap = o[0];
      q_wb = o[1];
      q_wa = o[2];
    } else {
      const float *const src[2] = {dirA, dirA};
      const int np[2] = {nA, nA}, ix[2] = {tcl, kRedLt + (int)threadIdx.x};
      double o[2];
      q_wb = redB[kRedLt + threadIdx.x];
      slab_sums<2>(src, np, ix, o);
      q_pap = o[0];
      q_wa = o[1];
    }
  } else {
    q_pap = redA[tcl];
    q_wb = redB[kRedLt + threadIdx.x];
    q_wa = redA[kRedLt + threadIdx.x];
  }
  double q_ci = 0.0;
  {
    const int kk0 = threadIdx.x >> 4, t0c = threadIdx.x & 15;
    if (K > 0) q_ci = Cinv[(kk0 < K ? kk0 : K - 1) * K + (t0c < K ? t0c : K - 1)];
  }
  asm volatile("" ::"v"(q_rz), "v"(q_res), "v"(q_zero), "v"(q_done), "v"(q_pap), "v"(q_wb), "v"(q_wa), "v"(q_ci));
  if (threadIdx.x < kMaxT) {
    float a = 0.f;
    if (!first && threadIdx.x < TT) {
      const float s = (float)q_pap;
      a = (fabsf(s) > eps) ? q_rz / s : 0.f;
      if (q_res < stop_after || q_zero || q_done) a = 0.f;
    }
    salpha[threadIdx.x] = a;
    if (blockIdx.x == 0 && !first && threadIdx.x < TT) alpha_out[threadIdx.x] = a;
  }
  __syncthreads();
  double tvB[4] = {0.0, 0.0, 0.0, 0.0};
  if (K > 0) {
    const int kk = threadIdx.x >> 4, t = threadIdx.x & 15;
    // w = L^T r of the residual this pass is about to form: direct value of the previous pass, one recurrence step
    double w = q_wb;
    if (!first) w -= (double)salpha[t] * q_wa;
    sW[threadIdx.x] = w;
    // Cinv through LDS: one load per thread (a loop of K dependent global loads cost ~10 us at the head of every workgroup)
    sTv[threadIdx.x] = (kk < K && t < K) ? q_ci : 0.0;
    __syncthreads();
    double tv = 0.0;
    if (kk < K && t < TT) {
#pragma unroll
      for (int cc = 0; cc < kMaxK; ++cc) tv = fma(sTv[kk * 16 + cc], sW[cc * 16 + t], tv);
    }
    __syncthreads();
    sTv[threadIdx.x] = tv;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) tvB[s] = sTv[(4 * s + ln.q) * 16 + ln.c];
  }
  const float a = salpha[ln.c];
  const double inv_s = K > 0 ? 1.0 / (double)sigma2 : 1.0;
  float acc_rr = 0.f, acc_rz = 0.f;
  floatx4m lt2[2] = {floatx4m{0.f, 0.f, 0.f, 0.f}, floatx4m{0.f, 0.f, 0.f, 0.f}};
  const unsigned Nu = (unsigned)N;
  auto do_tile = [&](unsigned tile, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const unsigned w0 = tile * 256u + 64u * ln.wave;
#pragma unroll
    for (int half = 0; half < 4 / NB; ++half) {
      float pv[NB][4], av[NB][4], xv[NB][4], rv[NB][4], lr[NB][4], lc[NB][4];
#pragma unroll
      for (int h = 0; h < NB; ++h) {
        const unsigned R0 = w0 + 16 * (NB * half + h);
        load_block<TT, FULL>(p, R0, Nu, ln, pv[h]);
        load_block<TT, FULL>(Ap, R0, Nu, ln, av[h]);
        load_block<TT, FULL>(x, R0, Nu, ln, xv[h]);
        load_block<TT, FULL>(r, R0, Nu, ln, rv[h]);
        if (K > 0) {
          load_L_rows<FULL>(L, R0, Nu, K, ln, lr[h]);
          load_L_cols<FULL>(L, R0, Nu, K, ln, lc[h]);
        }
      }
      float rn[NB][4], xn[NB][4], zn[NB][4];
#pragma unroll
      for (int h = 0; h < NB; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          xn[h][q] = __builtin_fmaf(a, pv[h][q], xv[h][q]);
          rn[h][q] = __builtin_fmaf(-a, av[h][q], rv[h][q]);
        }
      if (K > 0) {
        // the two blocks' float64 chains (4 dependent matrix instructions each) and their fp32 L^T r products interleaved
        doublex4m corr[NB];
#pragma unroll
        for (int h = 0; h < NB; ++h) corr[h] = doublex4m{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int h = 0; h < NB; ++h) {
            corr[h] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)lc[h][s], tvB[s], corr[h], 0, 0, 0);
            lt2[h & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(lr[h][s], rn[h][s], lt2[h & 1], 0, 0, 0);
          }
#pragma unroll
        for (int h = 0; h < NB; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q) zn[h][q] = (float)(((double)rn[h][q] - corr[h][q]) * inv_s);
      } else {
#pragma unroll
        for (int h = 0; h < NB; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q) zn[h][q] = rn[h][q];
      }
#pragma unroll
      for (int h = 0; h < NB; ++h) {
        const unsigned R0 = w0 + 16 * (NB * half + h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc_rr = __builtin_fmaf(rn[h][q], rn[h][q], acc_rr);
          acc_rz = __builtin_fmaf(rn[h][q], zn[h][q], acc_rz);
        }
        if (!first) {
          store_block<TT, FULL>(x, R0, Nu, ln, xn[h]);
          store_block<TT, FULL>(r, R0, Nu, ln, rn[h]);
        }
        if (K > 0 || first) store_block<TT, FULL>(z, R0, Nu, ln, zn[h]);     // identity preconditioner: pass C reads r as z
      }
    }
  };
  const unsigned ntiles = (Nu + 255u) / 256u;
  for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if ((tile + 1u) * 256u <= Nu) do_tile(tile, std::true_type{});
    else do_tile(tile, std::false_type{});
  }
  float *dst = part + (size_t)blockIdx.x * kRedW;
  block_colsum(acc_rr, sh, dst, ln);
  block_colsum(acc_rz, sh, dst + 16, ln);
  block_ltsum(lt2[0] + lt2[1], sh, dst + kRedLt, ln);
}

// ---- pass C -----------------------------------------------------------------------------------------------------------
// beta = rz' / rz; p = z + beta p; bookkeeping (workgroup 0): rz <- rz', resid, mean residual, beta history, and — on the
// iterations the host marks with check_now — the convergence decision (st->poll.done), which freezes later iterations.
// Best-iterate safeguard: fp32 CG on a system with cond(Khat) * 1e-6 >~ 1 (N s / sigma^2 beyond a few million) does not
// merely stall, its recurrence residual can GROW; the iterate with the smallest tested residual is kept in x_best (every
// workgroup takes the same decision from the same reduced numbers) and returned when the tolerance is never reached.
// first != 0: the set-up call (p = z, rz[0] = rz').
template <int TT, bool DIRECT = false>
__global__ __launch_bounds__(256) void k_pass_c(const float *__restrict__ z, float *__restrict__ p,
                                                const double *__restrict__ redB, CgState *__restrict__ st,
                                                float *__restrict__ beta_out, long long N, float eps, int cur, int first,
                                                int check_now, float tolerance, int iter_count, int stagnation_window,
                                                const float *__restrict__ x, float *__restrict__ x_best,
                                                CgPoll *__restrict__ poll_host, const float *__restrict__ dirB = nullptr,
                                                int nB = 0, double *__restrict__ lt_out = nullptr) {
  __shared__ float sbeta[kMaxT];
  __shared__ float sres[kMaxT];
  __shared__ float srzn[kMaxT];
  const Lane ln;
  // Every scalar this prologue needs is requested up front and pinned by one empty asm (hipcc otherwise sinks each load under
  // the condition that uses it: six dependent round trips — most of the kernel's time at small N)
  const int tcl = (int)threadIdx.x < TT ? (int)threadIdx.x : TT - 1;
  const int done_raw = st->done_pp[cur];
  const float snap_prev = st->snap_resid[cur];
  double q_rzn, q_rr;
  if constexpr (DIRECT) {
    if (blockIdx.x == 0) {
      // workgroup 0 also adds up the L^T r entries of this pass B for the NEXT pass B (its loads ride in the same batch as
      // the two every workgroup needs: no extra round trip here, one slab sum fewer at the head of every pass B)
      const float *const src[3] = {dirB, dirB, dirB};
      const int np[3] = {nB, nB, nB}, ix[3] = {16 + tcl, tcl, kRedLt + (int)threadIdx.x};
      double o[3];
      slab_sums<3>(src, np, ix, o);
      q_rzn = o[0];
      q_rr = o[1];
      lt_out[kRedLt + threadIdx.x] = o[2];
    } else {
      const float *const src[2] = {dirB, dirB};
      const int np[2] = {nB, nB}, ix[2] = {16 + tcl, tcl};
      double o[2];
      slab_sums<2>(src, np, ix, o);
      q_rzn = o[0];
      q_rr = o[1];
    }
  } else {
    q_rzn = redB[16 + tcl];
    q_rr = redB[tcl];
  }
  const float q_rz = st->rz[cur][tcl];
  const int q_zero = st->rhs_zero[tcl];
  asm volatile("" ::"v"(done_raw), "v"(snap_prev), "v"(q_rzn), "v"(q_rr), "v"(q_rz), "v"(q_zero));
  const int was_done = first ? 0 : done_raw;
  // the poll record of a tested iteration goes straight to pinned host memory (a D2H copy per poll is a 4 us copy kernel)
  if (was_done && blockIdx.x == 0 && threadIdx.x == 0) {
    st->done_pp[cur ^ 1] = was_done;
    if (check_now && poll_host) publish_poll(poll_host, st->poll, iter_count);
  }
  if (threadIdx.x < kMaxT) {
    float beta = 0.f, res = 0.f, rzn = 0.f;
    if (threadIdx.x < TT) {
      rzn = (float)q_rzn;
      beta = (!first && fabsf(q_rz) > eps) ? rzn / q_rz : 0.f;
      res = q_zero ? 0.f : sqrtf((float)q_rr);
    }
    sbeta[threadIdx.x] = beta;
    sres[threadIdx.x] = res;
    srzn[threadIdx.x] = rzn;
  }
  __syncthreads();
  if (was_done && !first) return;
  float mres = 0.f;
  for (int t = 0; t < TT; ++t) mres += sres[t];
  mres /= (float)TT;
  const bool improved = !first && check_now && mres == mres && mres < snap_prev;
  const float beta = sbeta[ln.c];
  const unsigned Nu = (unsigned)N;
  auto do_tile = [&](unsigned tile, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const unsigned w0 = tile * 256u + 64u * ln.wave;
    float pv[4][4], zv[4][4];
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) {
      load_block<TT, FULL>(p, w0 + 16 * bk, Nu, ln, pv[bk]);
      load_block<TT, FULL>(z, w0 + 16 * bk, Nu, ln, zv[bk]);
    }
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) {
#pragma unroll
      for (int q = 0; q < 4; ++q) pv[bk][q] = __builtin_fmaf(beta, pv[bk][q], zv[bk][q]);
      store_block<TT, FULL>(p, w0 + 16 * bk, Nu, ln, pv[bk]);
    }
    if (improved) {
#pragma unroll
      for (int bk = 0; bk < 4; ++bk) {
        float xv[4];
        load_block<TT, FULL>(x, w0 + 16 * bk, Nu, ln, xv);
        store_block<TT, FULL>(x_best, w0 + 16 * bk, Nu, ln, xv);
      }
    }
  };
  const unsigned ntiles = (Nu + 255u) / 256u;
  for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if ((tile + 1u) * 256u <= Nu) do_tile(tile, std::true_type{});
    else do_tile(tile, std::false_type{});
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0 && !first) {
      st->snap_resid[cur ^ 1] = improved ? mres : snap_prev;
      st->poll.snap_slot = cur ^ 1;
      st->poll.snap_cur = improved ? mres : snap_prev;
    }
    if (threadIdx.x < TT) {
      st->rz[first ? 0 : (cur ^ 1)][threadIdx.x] = srzn[threadIdx.x];
      if (!first) {
        st->resid[threadIdx.x] = sres[threadIdx.x];
        beta_out[threadIdx.x] = sbeta[threadIdx.x];
      }
    }
    if (threadIdx.x == 0 && !first) {
      const float m = mres;
      st->poll.mean_resid = m;
      if (check_now) {
        if (m != m) {
          st->poll.done = 2;
          st->poll.iters = iter_count;
        } else if (m < tolerance) {
          st->poll.done = 1;
          st->poll.iters = iter_count;
        } else if (snap_prev < 1.0f && m > 100.0f * snap_prev) {
          st->poll.done = 3;                      // diverging: give up, the saved iterate is returned
          st->poll.iters = iter_count;
        } else if (stagnation_window > 0) {
          if (m < 0.99f * st->poll.best_resid) {
            st->poll.best_resid = m;
            st->poll.since_best = 0;
          } else if (++st->poll.since_best >= stagnation_window) {
            st->poll.done = 3;
            st->poll.iters = iter_count;
          }
        }
        if (poll_host) publish_poll(poll_host, st->poll, iter_count);
      }
      st->done_pp[cur ^ 1] = st->poll.done;
    }
  }
}

// x *= |rhs| per column; when the tolerance was never reached and a better tested iterate was saved, return that one
// (the reported mean residual is updated by the host from the same two numbers)
__global__ __launch_bounds__(256) void k_unnormalise(float *__restrict__ x, const float *__restrict__ x_best,
                                                     const CgState *__restrict__ st, long long N, int T,
                                                     CgPoll *__restrict__ poll_host) {
  const float snap = st->snap_resid[st->poll.snap_slot];
  const bool use_best = st->poll.done != 1 && snap < st->poll.mean_resid;
  if (blockIdx.x == 0 && threadIdx.x == 0) publish_poll(poll_host, st->poll, 1);   // the final record: no copy behind the solve
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < N * T; e += (long long)gridDim.x * 256)
    x[e] = (use_best ? x_best[e] : x[e]) * st->rhs_norm[e % T];
}

// Workgroups of a streaming pass: ONE resident round of the kernel (occupancy x CUs; a second, partial round is a tail
// that costs a full tile time), and a tile count per workgroup that divides evenly.
template <typename K>
int resident_blocks(K kernel) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0) != hipSuccess || n < 1) n = 1;
  return n;
}
inline int device_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
      v = 256;
    return v;
  }();
  return n;
}
inline int blocks_for(long long N, int resident_per_cu) {
  const long long tiles = (N + 255) / 256;
  long long cap = (long long)resident_per_cu * device_cus();
  if (cap > kMaxBlocks) cap = kMaxBlocks;
  if (tiles <= cap) return tiles < 1 ? 1 : (int)tiles;
  const long long per = (tiles + cap - 1) / cap;       // tiles per workgroup
  return (int)((tiles + per - 1) / per);
}
struct Grids {
  int a, b, c;
};
template <int TT>
Grids grids_for(long long N) {
  static const int ra = resident_blocks(k_pass_a<TT>), rb = resident_blocks(k_pass_b<TT>), rc = resident_blocks(k_pass_c<TT>);
  return Grids{blocks_for(N, ra), blocks_for(N, rb), blocks_for(N, rc)};
}

// Pinned host landing zone for the lagged convergence polls (one per host thread; the executor keeps no other
// state between calls).
constexpr int kPollRing = 4;
constexpr size_t kHistFloats = (size_t)(kMaxHist + 1) * kMaxT;       // one coefficient history (the last row is a scratch slot)
struct PollCtx {
  CgPoll *host = nullptr;
  CgPoll *host_dev = nullptr;       // the same memory as the device sees it (pass C writes its poll record there)
  float *hist = nullptr;            // alpha history, then beta history: written by passes B / C straight into pinned memory
  float *hist_dev = nullptr;        //   (two D2H copies into pageable memory and a second synchronisation per solve otherwise)
  hipEvent_t ev[kPollRing];
  bool ok = false;
  int init() {
    if (ok) return 0;
    const size_t poll_bytes = ((kPollRing + 1) * sizeof(CgPoll) + 255) & ~(size_t)255;
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&host), poll_bytes + 2 * kHistFloats * sizeof(float),
                                 hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) return (int)e;
    e = hipHostGetDevicePointer(reinterpret_cast<void **>(&host_dev), host, 0);
    if (e != hipSuccess) return (int)e;
    hist = reinterpret_cast<float *>(reinterpret_cast<char *>(host) + poll_bytes);
    hist_dev = reinterpret_cast<float *>(reinterpret_cast<char *>(host_dev) + poll_bytes);
    for (int i = 0; i < kPollRing; ++i) {
      e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
      if (e != hipSuccess) return (int)e;
    }
    ok = true;
    return 0;
  }
};
thread_local PollCtx g_poll;
// Wait for a record a kernel publishes into pinned memory (publish_poll): the host SPINS on the stamp — the record of a
// 5 us pass arrives within microseconds of the pass finishing, where hipEventSynchronize / hipStreamSynchronize wake the
// thread up ~50 - 150 us later (measured as an idle device in front of k_unnormalise, once per solve, and again behind it).
// The spin is bounded by wall time (`budget_us`): a poll record of an EARLIER iteration is at most a few passes away
// (kPollSpinUs); the final record sits behind everything the host queued ahead, so after kFinalSpinUs the caller blocks in
// the runtime's own synchronisation instead of burning a core for the rest of a long solve (its late wake-up is then a
// negligible part of that solve) — which is also what reports a failed launch or a hung device.
constexpr long kPollSpinUs = 2000, kFinalSpinUs = 20000;
inline bool wait_record(const CgPoll *rec, int stamp, long budget_us) {
  const volatile int *p = &rec->seq;
  return rpgp_internal::spin_until([p, stamp] { return *p == stamp; }, budget_us);
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct ShardCtx {
  int mode = RPGP_SHARD_NONE, world = 1, rank = 0;
  rpgp_allreduce_fn fn = nullptr;
  void *ctx = nullptr;
  double *hist = nullptr;        // row mode: J x G x T float64 grid histogram
  float *H = nullptr;            // row mode: J x G x T Toeplitz product
  int reduce(void *buf, size_t count, int dtype, void *stream) const {
    if (!fn) return 0;
    rpgp_internal::TraceRange tr("rpgp:allreduce");
    return fn(ctx, buf, count, dtype, stream);
  }
};

// one application of the operator.  RPGP_SHARD_PARTIAL: this rank's partial product (noise on rank 0 only), summed over
// the ranks; RPGP_SHARD_ROWS (SKI): scatter the local rows, all-reduce the histogram, replicated Toeplitz product, gather
// the local rows.
size_t operator_workspace(const rpgp_operator *op, int T);

// out += add (N x T contiguous): the accumulation of a composite operator's parts
__global__ __launch_bounds__(256) void k_accumulate(float *__restrict__ out, const float *__restrict__ add, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] += add[i];
}

int apply_operator(const rpgp_operator *op, const ShardCtx &sh, const float *V, float *out, int T, void *ws, size_t ws_bytes,
                   void *stream) {
  if (op->kind == RPGP_OP_SUM) {
    // A = sum of `G` unsharded operators on the same N rows (prep -> rpgp_operator[G] in HOST memory; each part carries its
    // own scale and noise): part 0 straight into `out`, the others through an N x T scratch block at the head of the workspace
    const rpgp_operator *parts = reinterpret_cast<const rpgp_operator *>(op->prep);
    if (!parts || op->G <= 0 || sh.mode != RPGP_SHARD_NONE) return RPGP_EINVAL;
    const size_t nt = align256((size_t)op->N * T * sizeof(float));
    if (ws_bytes < nt) return RPGP_EWORKSPACE;
    float *tmp = reinterpret_cast<float *>(ws);
    void *pws = reinterpret_cast<char *>(ws) + nt;
    const long long n = (long long)op->N * T;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    for (int i = 0; i < op->G; ++i) {
      if (parts[i].kind == RPGP_OP_SUM || parts[i].N != op->N) return RPGP_EINVAL;
      const int rc = apply_operator(&parts[i], sh, V, i == 0 ? out : tmp, T, pws, ws_bytes - nt, stream);
      if (rc) return rc;
      if (i > 0) hipLaunchKernelGGL(k_accumulate, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, tmp, n);
    }
    return (int)hipGetLastError();
  }
  const bool partial = sh.mode == RPGP_SHARD_PARTIAL;
  const float noise = (partial && sh.rank != 0) ? 0.f : op->noise;
  const int world = op->world > 0 ? op->world : 1, rank = op->world > 0 ? op->rank : 0;
  int rc;
  switch (op->kind) {
    case RPGP_OP_FUSED:
      if (op->j1 <= op->j0) {       // J-sharding with more ranks than projections: an empty slice
        CG_CHECK(hipMemsetAsync(out, 0, (size_t)op->N * T * sizeof(float), reinterpret_cast<hipStream_t>(stream)));
        rc = 0;
      } else {
        rc = rpgp_mvm_sym_range(op->Z, V, out, op->N, op->ldz, T, op->j0, op->j1, world, rank, op->scale, noise, ws, ws_bytes,
                                stream);
      }
      break;
    case RPGP_OP_FUSED_PREPARED:
      if (op->j1 <= op->j0) {
        CG_CHECK(hipMemsetAsync(out, 0, (size_t)op->N * T * sizeof(float), reinterpret_cast<hipStream_t>(stream)));
        rc = 0;
      } else {
        rc = rpgp_mvm_sym_prepared_range(op->prep, V, out, op->N, op->J, T, op->j0, op->j1, world, rank, op->scale, noise,
                                         ws, ws_bytes, stream);
      }
      break;
    case RPGP_OP_SKI:
      if (sh.mode == RPGP_SHARD_ROWS) {
        const size_t nh = (size_t)op->J * op->G * T;
        if (op->N > 0) {
          // (the planned scatter answers RPGP_EWORKSPACE when its per-item scratch exceeds the plan's slab — rpgp.h: "fall
          //  back to rpgp_ski_scatter" — so the executor does exactly that instead of failing the solve)
          rc = op->prep ? rpgp_ski_scatter_planned(op->prep, V, sh.hist, op->N, op->J, op->G, T, ws, ws_bytes, stream)
                        : RPGP_EWORKSPACE;
          if (rc == RPGP_EWORKSPACE)
            rc = rpgp_ski_scatter(op->Z, op->grid_params, V, sh.hist, op->N, op->ldz, op->J, op->G, T, ws, ws_bytes, stream);
          if (rc) return rc;
        } else {
          CG_CHECK(hipMemsetAsync(sh.hist, 0, nh * sizeof(double), reinterpret_cast<hipStream_t>(stream)));
        }
        rc = sh.reduce(sh.hist, nh, RPGP_F64, stream);
        if (rc) return rc;
        if (op->N <= 0) return 0;
        rc = rpgp_ski_grid_product(sh.hist, op->grid_params, sh.H, op->J, op->G, T, stream);
        if (rc) return rc;
        return rpgp_ski_gather_fast(op->prep, op->Z, op->grid_params, sh.H, V, out, op->N, op->ldz, op->J, op->G, T, op->scale,
                                    op->noise, stream);
      }
      rc = RPGP_EWORKSPACE;
      if (op->prep && T <= 12)
        rc = rpgp_ski_mvm_planned(op->prep, op->Z, op->grid_params, V, out, op->N, op->ldz, op->J, op->G, T, op->scale, noise,
                                  ws, ws_bytes, stream);
      if (rc == RPGP_EWORKSPACE)    // no plan, a wide block, or a plan whose scratch bound is exceeded (large N J, small G)
        rc = rpgp_ski_mvm(op->Z, op->Z, op->grid_params, V, out, op->N, op->N, op->ldz, op->ldz, op->J, op->G, T, op->scale,
                          noise, ws, ws_bytes, stream);
      break;
    case RPGP_OP_DENSE:
      rc = rpgp_dense_mvm(op->Kd, V, out, op->N, op->ldk, T, noise, stream);
      break;
    case RPGP_OP_SYMCACHE:
      rc = rpgp_symcache_mvm(op->Kd, (size_t)op->ldk, op->G, V, out, op->N, T, op->scale, noise, world, rank, ws, ws_bytes,
                             stream);
      break;
    case RPGP_OP_FAMILY:
      rc = rpgp_family_mvm_sym(op->family, op->Z, V, out, op->N, op->ldz, T, op->scale, noise, ws, ws_bytes, stream);
      break;
    default:
      return RPGP_EINVAL;
  }
  if (rc) return rc;
  if (partial) return sh.reduce(out, (size_t)op->N * T, RPGP_F32, stream);
  return 0;
}

size_t operator_workspace(const rpgp_operator *op, int T) {
  const int world = op->world > 0 ? op->world : 1, rank = op->world > 0 ? op->rank : 0;
  switch (op->kind) {
    case RPGP_OP_FUSED:
    case RPGP_OP_FUSED_PREPARED:
      return rpgp_mvm_sym_range_workspace_bytes(op->N, T, world, rank);
    case RPGP_OP_SKI:
      return rpgp_ski_workspace_bytes(op->J, op->G, T);
    case RPGP_OP_DENSE:
      return 256;
    case RPGP_OP_SYMCACHE:
      return rpgp_symcache_workspace_bytes(op->N, T, world, rank);
    case RPGP_OP_FAMILY:
      return rpgp_family_mvm_workspace_bytes(op->N, op->N, T, 1);
    case RPGP_OP_SUM: {
      const rpgp_operator *parts = reinterpret_cast<const rpgp_operator *>(op->prep);
      size_t mx = 0;
      for (int i = 0; parts && i < op->G; ++i) {
        if (parts[i].kind == RPGP_OP_SUM) continue;
        const size_t b = operator_workspace(&parts[i], T);
        mx = b > mx ? b : mx;
      }
      return align256((size_t)(op->N > 0 ? op->N : 1) * T * sizeof(float)) + mx;
    }
    default:
      return 0;
  }
}

inline size_t ski_stage_bytes(const rpgp_operator *op, int T) {
  if (op->kind != RPGP_OP_SKI) return 0;
  const size_t nh = (size_t)op->J * op->G * T;
  return align256(nh * sizeof(double)) + align256(nh * sizeof(float));
}

}  // namespace

extern "C" {

size_t rpgp_mbcg_workspace_bytes(const rpgp_operator *op, int T, int precond_rank) {
  if (!op || T <= 0 || T > kMaxT || precond_rank < 0 || precond_rank > kMaxK || op->N < 0) return 0;
  const size_t nt = (size_t)(op->N > 0 ? op->N : 1) * T * sizeof(float);
  size_t total = 5 * align256(nt);                                             // r, p, z, Ap, x_best
  total += align256(sizeof(CgState));
  total += align256((size_t)kMaxBlocks * kRedW * sizeof(float));               // per-workgroup partials
  total += 2 * align256((size_t)kRedW * sizeof(double));                       // reduced vectors A / B
  total += ski_stage_bytes(op, T);
  total += align256(operator_workspace(op, T));
  return total;
}

int rpgp_mbcg_solve(const rpgp_operator *op, const float *rhs, float *x, int T, int max_iter, int min_iter,
                    int hist_len, int check_every, int stagnation_window, float tolerance, int precond_rank,
                    const float *L, const double *Cinv, float precond_sigma2, const rpgp_reducer *reducer,
                    float *alpha_hist_host, float *beta_hist_host, int *iterations_host, float *mean_resid_host,
                    void *workspace, size_t workspace_bytes, void *stream) {
  if (!op || !rhs || !x || T <= 0 || T > kMaxT || max_iter < 0 || hist_len < 0 || check_every <= 0 ||
      precond_rank < 0 || precond_rank > kMaxK || (precond_rank > 0 && (!L || !Cinv)) || op->N < 0)
    return RPGP_EINVAL;
  if (hist_len > kMaxHist || (hist_len > 0 && (!alpha_hist_host || !beta_hist_host))) return RPGP_EINVAL;
  ShardCtx sh;
  long long global_N = op->N;
  if (reducer && reducer->mode != RPGP_SHARD_NONE && reducer->fn) {      // (world == 1 still goes through the hook)
    if ((reducer->mode != RPGP_SHARD_PARTIAL && reducer->mode != RPGP_SHARD_ROWS) || reducer->world < 1 ||
        reducer->rank < 0 || reducer->rank >= reducer->world)
      return RPGP_EINVAL;
    if (reducer->mode == RPGP_SHARD_ROWS && (op->kind != RPGP_OP_SKI || T > 12 || reducer->global_N < op->N))
      return RPGP_EINVAL;
    if (reducer->mode == RPGP_SHARD_PARTIAL && op->kind != RPGP_OP_FUSED && op->kind != RPGP_OP_FUSED_PREPARED &&
        op->kind != RPGP_OP_SYMCACHE)
      return RPGP_EINVAL;
    sh.mode = reducer->mode;
    sh.world = reducer->world;
    sh.rank = reducer->rank;
    sh.fn = reducer->fn;
    sh.ctx = reducer->ctx;
    if (sh.mode == RPGP_SHARD_ROWS) global_N = reducer->global_N;
  }
  if (op->N <= 0 && sh.mode != RPGP_SHARD_ROWS) return RPGP_EINVAL;
  if (op->N >= (1LL << 27)) return RPGP_EINVAL;              // 32-bit element offsets in the streaming passes (N * 16 < 2^31)
  if (!workspace || workspace_bytes < rpgp_mbcg_workspace_bytes(op, T, precond_rank)) return RPGP_EWORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  rpgp_internal::TraceRange tr_solve("rpgp:mbcg_solve");
  const long long N = op->N;
  const int K = precond_rank;
  const size_t nt = (size_t)(N > 0 ? N : 1) * T * sizeof(float);

  char *w = reinterpret_cast<char *>(workspace);
  float *r = reinterpret_cast<float *>(w); w += align256(nt);
  float *p = reinterpret_cast<float *>(w); w += align256(nt);
  float *z = reinterpret_cast<float *>(w); w += align256(nt);
  float *Ap = reinterpret_cast<float *>(w); w += align256(nt);
  float *x_best = reinterpret_cast<float *>(w); w += align256(nt);
  CgState *state = reinterpret_cast<CgState *>(w); w += align256(sizeof(CgState));
  float *part = reinterpret_cast<float *>(w); w += align256((size_t)kMaxBlocks * kRedW * sizeof(float));
  double *redA = reinterpret_cast<double *>(w); w += align256((size_t)kRedW * sizeof(double));
  double *redB = reinterpret_cast<double *>(w); w += align256((size_t)kRedW * sizeof(double));
  if (op->kind == RPGP_OP_SKI) {
    const size_t nh = (size_t)op->J * op->G * T;
    sh.hist = reinterpret_cast<double *>(w); w += align256(nh * sizeof(double));
    sh.H = reinterpret_cast<float *>(w); w += align256(nh * sizeof(float));
  }
  void *op_ws = w;
  const size_t op_ws_bytes = operator_workspace(op, T);
  const bool rows = sh.mode == RPGP_SHARD_ROWS;

  Grids gr{1, 1, 1};
  CG_DISPATCH_T(T, gr = grids_for<TT>(N));
  const int nba = gr.a, nbb = gr.b, nbc = gr.c;
  const int nred = kRedW / 8;
  const float eps = 1e-30f, stop_after = 1e-10f;
  {
    const int prc = g_poll.init();
    if (prc) return prc;
  }
  CgPoll *hpoll = g_poll.host;
  float *alpha_d = g_poll.hist_dev, *beta_d = g_poll.hist_dev + kHistFloats;      // (pinned host memory, device view)
  // with the identity preconditioner z IS r: pass B skips the store and pass C reads r
  float *zsrc = K > 0 ? z : r;

  // Few slabs and no all-reduce between the sums and their consumer: the consuming pass adds the slabs up itself (slab_sums)
  // — 45 launches fewer per optimiser step at the C2 / C3 shapes.  Four slab areas: the A pass, the set-up pass, and the B
  // pass ping-pong (pass B of iteration i reads the L^T r slabs of iteration i - 1 while it writes its own).
  // RPGP_CG_DIRECT=0 keeps the separate k_reduce launches (A/B measurements).
  const char *env_direct = getenv("RPGP_CG_DIRECT");
  const bool direct = !rows && nba <= kDirectParts && nbb <= kDirectParts && !(env_direct && env_direct[0] == '0');
  float *partA = part, *partI = part + (size_t)kDirectParts * kRedW;
  float *partB[2] = {part + (size_t)2 * kDirectParts * kRedW, part + (size_t)3 * kDirectParts * kRedW};
  static_assert(kMaxBlocks >= 4 * kDirectParts, "slab areas of the consumer-side reduction");

#define CG_REDUCE(src_, dst_, nparts_)                                                               \
  do {                                                                                               \
    if (!direct) hipLaunchKernelGGL(k_reduce, dim3(nred), dim3(256), 0, st, src_, nparts_, dst_);    \
    if (rows) {                                                                                      \
      const int rrc_ = sh.reduce(dst_, kRedW, RPGP_F64, stream);                                     \
      if (rrc_) return rrc_;                                                                         \
    }                                                                                                \
  } while (0)
  const float *nil = nullptr;

  // set-up: |rhs| per column; r = rhs / |rhs|, x = p = Ap = 0, w0 = L^T r0; z0 = M^-1 r0, rz0; p0 = z0
  if (!direct) {
    partI = part;
    partB[0] = partB[1] = part;
  }
  CG_DISPATCH_T(T, hipLaunchKernelGGL((k_pass_a<TT>), dim3(nba), dim3(256), 0, st, rhs, rhs, L, partA, N, 0));
  CG_REDUCE(partA, redA, nba);
  if (direct) {
    CG_DISPATCH_T(T, hipLaunchKernelGGL((k_init<TT, true>), dim3(nba), dim3(256), 0, st, rhs, redA, r, x, p, Ap, L, partI, state,
                                        N, K, partA, nba));
  } else {
    CG_DISPATCH_T(T, hipLaunchKernelGGL((k_init<TT, false>), dim3(nba), dim3(256), 0, st, rhs, redA, r, x, p, Ap, L, partI, state,
                                        N, K, nil, 0));
  }
  CG_REDUCE(partI, redB, nba);
  // (the set-up pass B reads only the L^T r0 sums; its `A` source must merely be valid memory)
#define CG_PASS_B(DIR_, ...) \
  CG_DISPATCH_T(T, hipLaunchKernelGGL((k_pass_b<TT, 1, DIR_>), dim3(nbb), dim3(256), 0, st, __VA_ARGS__))
#define CG_PASS_C(DIR_, ...) \
  CG_DISPATCH_T(T, hipLaunchKernelGGL((k_pass_c<TT, DIR_>), dim3(nbc), dim3(256), 0, st, __VA_ARGS__))
  if (direct) {
    CG_PASS_B(true, p, Ap, x, r, z, L, Cinv, redA, redB, partB[1], state, alpha_d, N, K, precond_sigma2, eps, stop_after, 0, 1,
              partA, nba, partI, nba);
    CG_PASS_C(true, z, p, redB, state, beta_d, N, eps, 0, 1, 0, tolerance, 0, 0, x, x_best, (CgPoll *)nullptr, partB[1], nbb,
              redB);
  } else {
    CG_PASS_B(false, p, Ap, x, r, z, L, Cinv, redA, redB, partB[1], state, alpha_d, N, K, precond_sigma2, eps, stop_after, 0, 1,
              nil, 0, nil, 0);
    CG_REDUCE(partB[1], redB, nbb);
    CG_PASS_C(false, z, p, redB, state, beta_d, N, eps, 0, 1, 0, tolerance, 0, 0, x, x_best, (CgPoll *)nullptr, nil, 0);
  }
  CG_CHECK(hipGetLastError());

  int it = 0;
  const int n_iter = max_iter < global_N ? max_iter : (int)global_N;
  const int min_it = min_iter < n_iter - 1 ? min_iter : n_iter - 1;
  const int n_hist = hist_len < n_iter ? hist_len : n_iter;
  // The convergence decision is taken on the device (pass C), which writes its poll record straight into pinned host
  // memory; the host reads it one iteration late (after the event recorded behind that pass C), so the next iteration is
  // already queued while it waits and the GPU never idles on the round trip.  The iteration queued past convergence is a
  // no-op on x (alpha = 0).
  int polled_it = -1;                     // iteration whose poll is in flight (-1: none)
  CgPoll last = {1.0f, 0, 0, 0.f, 0, 0, 3.0e38f, 0};
  // One iteration's launches: the executor only ENQUEUES (the hipGraph form of this loop — one captured iteration replayed —
  // was built in round 5 and measured 6 - 17 % slower than this queue-ahead form, whose launches already run ahead of the
  // device: DESIGN.md §3.4, tools/experiments/r5_cg_graph_form.patch).
  auto enqueue_iteration = [&](int it) -> int {
    rpgp_internal::TraceRange tr_it(it % 8 == 0 ? "rpgp:cg_iteration" : nullptr);
    const int rc = apply_operator(op, sh, p, Ap, T, op_ws, op_ws_bytes, stream);
    if (rc) return rc;
    float *pb_new = partB[it & 1];   // (ping-pong kept: the set-up pass B's slabs live in partB[1] until iteration 0 has read redB)
    CG_DISPATCH_T(T, hipLaunchKernelGGL((k_pass_a<TT>), dim3(nba), dim3(256), 0, st, p, Ap, L, partA, N, K));
    CG_REDUCE(partA, redA, nba);
    const int slot = it < hist_len ? it : kMaxHist;      // history row (the last row is a scratch slot)
    const bool hist_pending = it < n_hist - 1;
    const bool check_now = it >= min_it && !hist_pending && (it % check_every == 0 || it == n_iter - 1);
    CgPoll *poll_dst = check_now ? g_poll.host_dev + (it % kPollRing) : (CgPoll *)nullptr;
    const int cur = it & 1, count = it + 1;
    if (direct) {
      CG_PASS_B(true, p, Ap, x, r, z, L, Cinv, redA, redB, pb_new, state, alpha_d + (size_t)slot * kMaxT, N, K, precond_sigma2,
                eps, stop_after, cur, 0, partA, nba, nil, 0);
      CG_PASS_C(true, zsrc, p, redB, state, beta_d + (size_t)slot * kMaxT, N, eps, cur, 0, check_now ? 1 : 0, tolerance,
                count, stagnation_window, x, x_best, poll_dst, pb_new, nbb, redB);
    } else {
      CG_PASS_B(false, p, Ap, x, r, z, L, Cinv, redA, redB, pb_new, state, alpha_d + (size_t)slot * kMaxT, N, K, precond_sigma2,
                eps, stop_after, cur, 0, nil, 0, nil, 0);
      CG_REDUCE(pb_new, redB, nbb);
      CG_PASS_C(false, zsrc, p, redB, state, beta_d + (size_t)slot * kMaxT, N, eps, cur, 0, check_now ? 1 : 0, tolerance,
                count, stagnation_window, x, x_best, poll_dst, nil, 0, (double *)nullptr);
    }
    return (int)hipGetLastError();
  };
  for (it = 0; it < n_iter; ++it) {
    const bool hist_pending = it < n_hist - 1;
    const bool check_now = it >= min_it && !hist_pending && (it % check_every == 0 || it == n_iter - 1);
    if (check_now) *reinterpret_cast<volatile int *>(&hpoll[it % kPollRing].seq) = 0;      // (stale stamp of an earlier solve)
    const int rc = enqueue_iteration(it);
    if (rc) return rc;
    if (polled_it >= 0) {                 // consume the previous poll while this iteration runs
      if (!wait_record(&hpoll[polled_it % kPollRing], polled_it + 1, kPollSpinUs)) CG_CHECK(hipEventSynchronize(g_poll.ev[polled_it % kPollRing]));
      last = hpoll[polled_it % kPollRing];
      polled_it = -1;
      if (last.done) {
        ++it;
        break;
      }
    }
    if (check_now) {
      CG_CHECK(hipEventRecord(g_poll.ev[it % kPollRing], st));
      polled_it = it;
    }
  }
#undef CG_REDUCE
#undef CG_PASS_B
#undef CG_PASS_C
  // x *= |rhs| (or the saved best iterate); the final poll record and the coefficient histories are already in pinned host
  // memory when the ONE synchronisation behind the solve returns
  // (the host needs the record and the histories, not x: what the caller enqueues next is ordered behind k_unnormalise on the
  //  stream, so the solve returns as soon as the record lands — every earlier kernel's pinned writes are complete by then)
  if (N > 0) {
    *reinterpret_cast<volatile int *>(&hpoll[kPollRing].seq) = 0;
    hipLaunchKernelGGL(k_unnormalise, dim3(nbc), dim3(256), 0, st, x, x_best, state, N, T, g_poll.host_dev + kPollRing);
    CG_CHECK(hipGetLastError());
    if (!wait_record(&hpoll[kPollRing], 1, kFinalSpinUs)) CG_CHECK(hipStreamSynchronize(st));
  } else {
    CG_CHECK(hipMemcpyAsync(&hpoll[kPollRing], &state->poll, sizeof(CgPoll), hipMemcpyDeviceToHost, st));
    CG_CHECK(hipStreamSynchronize(st));
  }
  last = hpoll[kPollRing];
  const int iters_done = last.done ? last.iters : it;
  const int nh = iters_done < hist_len ? iters_done : hist_len;
  if (nh > 0) {   // history rows are kMaxT wide, [hist_len][kMaxT] on the caller's side
    memcpy(alpha_hist_host, g_poll.hist, (size_t)nh * kMaxT * sizeof(float));
    memcpy(beta_hist_host, g_poll.hist + kHistFloats, (size_t)nh * kMaxT * sizeof(float));
  }
  if (iterations_host) *iterations_host = iters_done;
  if (last.done != 1 && last.snap_cur < last.mean_resid) last.mean_resid = last.snap_cur;   // the saved iterate was returned
  if (mean_resid_host) *mean_resid_host = last.mean_resid;
  if (last.done == 2 || last.mean_resid != last.mean_resid) return RPGP_ENUMERIC;
  return 0;
}

// ---- stochastic Lanczos quadrature from the coefficient histories (host arithmetic) -----------------------------------------
// log|A| ~ (n / p) sum_probes sum_m (Q[0][m])^2 log(lambda_m) over the Lanczos tridiagonals the CG coefficients define
// (GPyTorch's linear_cg bookkeeping):
//   T[k][k] = 1 / alpha_k + beta_{k-1} / alpha_{k-1},   T[k][k+1] = sqrt(beta_k) / alpha_k,
// a masked alpha (|alpha| <= 1e-30: converged column) counts as reciprocal 1 — the trailing block it decouples carries no
// weight.  Eigenvalues and the FIRST components of the eigenvectors by the implicit-shift QL iteration on (diag, off-diag),
// rotating only a first-row vector.  One probe's iteration is a chain of dependent divisions and square roots (~100 cycles
// per plane rotation, ~500 rotations for 20 Lanczos steps); the probes are independent, so they advance ROUND-ROBIN, one
// rotation each (QlProbe below): the out-of-order core overlaps the chains and ten 20 x 20 problems take ~30 us instead of
// the ~150 us of one after the other (or of a batched LAPACK call through torch, plus ~100 us of numpy indexing to lay the
// matrices out) — per optimiser step, on the host-bound stretch between the solve and the backward pass.
}  // extern "C"

namespace {

struct QlProbe {
  double *d, *e, *z;      // diagonal, off-diagonal (e[k] couples k and k + 1), first row of the eigenvector matrix
  int m, l, mm, i, sweeps, phase;      // phase 0: deflation test / sweep set-up, 1: inside a sweep, 2: finished, 3: failed
  double g, sn, cs, pp;
};

// one unit of work: the set-up of a sweep, or one plane rotation of it
inline void ql_step(QlProbe &q) {
  double *d = q.d, *e = q.e, *z = q.z;
  if (q.phase == 0) {
    const int m = q.m;
    for (;;) {
      if (q.l >= m) {
        q.phase = 2;
        return;
      }
      int mm = q.l;
      for (; mm < m - 1; ++mm) {
        const double dd = fabs(d[mm]) + fabs(d[mm + 1]);
        if (fabs(e[mm]) <= 2.3e-16 * dd) break;
      }
      if (mm != q.l) {
        q.mm = mm;
        break;
      }
      ++q.l;                 // d[l] is an eigenvalue
      q.sweeps = 0;
    }
    if (++q.sweeps > 80) {
      q.phase = 3;
      return;
    }
    const int l = q.l;
    double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
    const double r = sqrt(g * g + 1.0);
    q.g = d[q.mm] - d[l] + e[l] / (g + (g >= 0.0 ? r : -r));
    q.sn = 1.0;
    q.cs = 1.0;
    q.pp = 0.0;
    q.i = q.mm - 1;
    q.phase = 1;
    return;
  }
  const int i = q.i;
  double f = q.sn * e[i];
  const double b = q.cs * e[i];
  double r = sqrt(f * f + q.g * q.g);      // (entries of a CG tridiagonal: no overflow to guard, libm's hypot is 10 x slower)
  e[i + 1] = r;
  if (r == 0.0) {                          // an exact zero inside the sweep: split and start over at the same l
    d[i + 1] -= q.pp;
    e[q.mm] = 0.0;
    q.phase = 0;
    return;
  }
  const double inv_r = 1.0 / r, sn = f * inv_r, cs = q.g * inv_r;
  double g = d[i + 1] - q.pp;
  r = (d[i] - g) * sn + 2.0 * cs * b;
  const double pp = sn * r;
  d[i + 1] = g + pp;
  q.g = cs * r - b;
  q.sn = sn;
  q.cs = cs;
  q.pp = pp;
  f = z[i + 1];
  z[i + 1] = sn * z[i] + cs * f;
  z[i] = cs * z[i] - sn * f;
  if (--q.i < q.l) {
    d[q.l] -= pp;
    e[q.l] = q.g;
    e[q.mm] = 0.0;
    q.phase = 0;
  }
}

}  // namespace

extern "C" {

int rpgp_slq_logdet(const float *alpha_hist, const float *beta_hist, int iters, int ld, int num_probes, double n,
                    double *logdet_out) {
  if (!alpha_hist || !beta_hist || !logdet_out || iters <= 0 || iters > 4096 || num_probes <= 0 || num_probes > 1024 ||
      ld < num_probes)
    return RPGP_EINVAL;
  const int m = iters;
  double *buf = static_cast<double *>(malloc(sizeof(double) * 3 * (size_t)(m + 1) * num_probes));
  QlProbe *qs = static_cast<QlProbe *>(malloc(sizeof(QlProbe) * num_probes));
  if (!buf || !qs) {
    free(buf);
    free(qs);
    return RPGP_EINVAL;
  }
  for (int pr = 0; pr < num_probes; ++pr) {
    QlProbe &q = qs[pr];
    q.d = buf + (size_t)pr * 3 * (m + 1);
    q.e = q.d + (m + 1);
    q.z = q.e + (m + 1);
    double inv_prev = 1.0, beta_prev = 0.0;
    for (int k = 0; k < m; ++k) {
      const double a = alpha_hist[(size_t)k * ld + pr], b = beta_hist[(size_t)k * ld + pr];
      const double inv_a = fabs(a) > 1e-30 ? 1.0 / a : 1.0;
      q.d[k] = k == 0 ? inv_a : inv_a + beta_prev * inv_prev;
      q.e[k] = sqrt(b > 0.0 ? b : 0.0) * inv_a;          // couples k and k + 1 (the last one is unused)
      inv_prev = inv_a;
      beta_prev = b;
      q.z[k] = k == 0 ? 1.0 : 0.0;
    }
    q.e[m - 1] = 0.0;
    q.m = m;
    q.l = 0;
    q.sweeps = 0;
    q.phase = 0;
  }
  for (bool active = true; active;) {
    active = false;
    for (int pr = 0; pr < num_probes; ++pr)
      if (qs[pr].phase < 2) {
        ql_step(qs[pr]);
        active = true;
      }
  }
  double total = 0.0;
  int rc = 0;
  for (int pr = 0; pr < num_probes; ++pr) {
    const QlProbe &q = qs[pr];
    if (q.phase == 3) rc = RPGP_ENUMERIC;
    double acc = 0.0;
    for (int k = 0; k < m; ++k) acc += q.z[k] * q.z[k] * log(q.d[k] > 1e-30 ? q.d[k] : 1e-30);
    total += acc;
  }
  free(buf);
  free(qs);
  if (rc) return rc;
  *logdet_out = n * total / num_probes;
  return 0;
}

}  // extern "C"
