// rpgp_cg.hip — native mBCG executor: preconditioned batched conjugate gradients with Lanczos coefficients
//  as ONE host call.  The operator application is a direct call into the fused
// kernels of rpgp_kernels.hip; the vector recurrences are four fused kernels per iteration whose scalars (alpha, beta,
// r.z, residual norms) live in device memory, so there is no host synchronisation except the periodic convergence test.
//
// Per iteration (T <= 16 right-hand sides, row-major N x T):
//   Ap = A p                                            rpgp_mvm_sym[_prepared] / rpgp_ski_mvm
//   k_coldot : partial sums of p.Ap per column
//   k_update : alpha = rz / pAp ; x += alpha p ; r -= alpha Ap ; partial |r|^2 ; partial L^T r (preconditioner)
//   k_precond: w = Cinv (L^T r) ; z = (r - L w) / sigma^2 ; partial r.z            (identity preconditioner: z = r)
//   k_direction: beta = rz' / rz ; p = z + beta p ; alpha/beta history ; mean residual norm ; convergence decision
// The convergence flag lives on the device; the host polls it one iteration late through pinned memory, so the queue
// never drains on the round trip (the one iteration queued past convergence is a no-op on x).
// Partial sums are per-workgroup slabs reduced in a fixed order by the consuming kernel's prologue (deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#include "../../include/rpgp.h"

namespace {

constexpr int kMaxT = 16;
constexpr int kMaxK = 16;          // preconditioner rank
constexpr int kMaxBlocks = 512;     // workgroups of the streaming vector kernels (consumers reduce the T-wide partials in their prologue)
constexpr int kMaxBlocksW = 512;    // ... of k_update / k_Ltr, whose K x T wide L^T r partials are reduced once by k_wsolve
constexpr int kMaxHist = 64;        // Lanczos coefficients kept for at most this many iterations

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
};
struct CgState {
  float rz[2][kMaxT];       // r.z of the current iterate, ping-pong by iteration parity (no intra-kernel race)
  float rhs_norm[kMaxT];
  float resid[kMaxT];       // residual norms of the current iterate
  int rhs_zero[kMaxT];
  float snap_resid[2];      // mean residual of the iterate saved in x_best (ping-pong by iteration parity)
  CgPoll poll;
};

__device__ __forceinline__ float block_sum(float v, float *sh) {
  // 256 threads -> one value (all threads get it)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// dst[t] = sum_q partial[q][t] for t < T with all 256 threads: 16 lanes per column take every 16th slab, then the 16
// lane sums are added in a fixed order (deterministic).  `scratch` holds 256 floats.  Ends with a barrier.
__device__ __forceinline__ void reduce_partials(const float *__restrict__ partial, int nparts, int T,
                                                float *__restrict__ dst, float *__restrict__ scratch) {
  const int t = threadIdx.x & (kMaxT - 1), q0 = threadIdx.x / kMaxT;
  float s = 0.f;
  if (t < T)
    for (int q = q0; q < nparts; q += 256 / kMaxT) s += partial[(size_t)q * T + t];
  scratch[q0 * kMaxT + t] = s;
  __syncthreads();
  if ((int)threadIdx.x < T) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 256 / kMaxT; ++q) tot += scratch[q * kMaxT + threadIdx.x];
    dst[threadIdx.x] = tot;
  }
  __syncthreads();
}

// partial[blk][t] = sum over the block's rows of a[i][t] * b[i][t]
// (TT = exact number of columns: the per-row loads are unconditional and issued together; with a runtime T every load sat
// behind its own `t < T` branch — the same pathology the cached-K stream had, §DESIGN 3.2)
template <int TT>
__global__ __launch_bounds__(256) void k_coldot(const float *__restrict__ a, const float *__restrict__ b,
                                                float *__restrict__ partial, long long N) {
  __shared__ float sh[4];
  float acc[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) acc[t] = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
    float av[TT], bv[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) { av[t] = a[i * TT + t]; bv[t] = b[i * TT + t]; }
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = __builtin_fmaf(av[t], bv[t], acc[t]);
  }
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const float s = block_sum(acc[t], sh);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * TT + t] = s;
  }
}

// r = rhs / |rhs| (columns with |rhs| < 1e-10 are flagged zero and left unscaled), x = 0
__global__ __launch_bounds__(256) void k_normalise(const float *__restrict__ rhs, const float *__restrict__ partial,
                                                   int nparts, float *__restrict__ r, float *__restrict__ x,
                                                   CgState *__restrict__ st, long long N, int T) {
  __shared__ float ssum[kMaxT];
  __shared__ float snorm[kMaxT];
  __shared__ float scratch[256];
  reduce_partials(partial, nparts, T, ssum, scratch);
  if ((int)threadIdx.x < T) {
    float nrm = sqrtf(ssum[threadIdx.x]);
    const int zero = nrm < 1e-10f;
    if (zero) nrm = 1.0f;
    snorm[threadIdx.x] = nrm;
    if (blockIdx.x == 0) {
      st->rhs_norm[threadIdx.x] = nrm;
      st->rhs_zero[threadIdx.x] = zero;
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
  }
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256)
    for (int t = 0; t < T; ++t) {
      r[i * T + t] = rhs[i * T + t] / snorm[t];
      x[i * T + t] = 0.f;
    }
}

// One 256-row tile of w += L^T r: the tile's r (kMaxT-wide rows in sR) and L rows (sL) are in LDS; thread (kk, t)
// accumulates its entry over the tile's rows.  Callers bracket it with barriers.
__device__ __forceinline__ float ltr_tile(const float *__restrict__ sL, const float *__restrict__ sR, float acc) {
  const int kk = threadIdx.x / kMaxT, t = threadIdx.x % kMaxT;
#pragma unroll 8
  for (int rr = 0; rr < 256; ++rr) acc = __builtin_fmaf(sL[rr * kMaxK + kk], sR[rr * kMaxT + t], acc);
  return acc;
}

__device__ __forceinline__ void load_L_tile(const float *__restrict__ L, float *__restrict__ sL, long long row0,
                                            long long N, int K) {
  for (int e = threadIdx.x; e < 256 * kMaxK; e += 256) {
    const int rr = e / kMaxK, kk = e % kMaxK;
    sL[e] = (kk < K && row0 + rr < N) ? L[(row0 + rr) * K + kk] : 0.f;
  }
}

// partial_w[blk][kk][t] = sum over the block's rows of L[i][kk] r[i][t]       (L: N x K row-major); first iterate only
__global__ __launch_bounds__(256) void k_Ltr(const float *__restrict__ L, const float *__restrict__ r,
                                             float *__restrict__ partial_w, long long N, int T, int K) {
  __shared__ float sL[256 * kMaxK];
  __shared__ float sR[256 * kMaxT];
  float acc = 0.f;
  for (long long tile = blockIdx.x, nt = (N + 255) / 256; tile < nt; tile += gridDim.x) {
    const long long row0 = tile * 256;
    __syncthreads();
    load_L_tile(L, sL, row0, N, K);
    for (int e = threadIdx.x; e < 256 * kMaxT; e += 256) {
      const int rr = e / kMaxT, t = e % kMaxT;
      sR[e] = (t < T && row0 + rr < N) ? r[(row0 + rr) * T + t] : 0.f;
    }
    __syncthreads();
    acc = ltr_tile(sL, sR, acc);
  }
  const int kk = threadIdx.x / kMaxT, t = threadIdx.x % kMaxT;
  if (kk < K && t < T) partial_w[((size_t)blockIdx.x * K + kk) * T + t] = acc;
}

// z = M^-1 r with M = L L^T + sigma2 I (Woodbury, Cinv = (sigma2 I + L^T L)^-1), partial_rz = sum r.z
// tv[a][t] = sum_b Cinv[a][b] * (sum over the producer's workgroups of partial_w[.][b][t]) in float64, ONE workgroup per
// column t.  This used to be the prologue of EVERY k_precond workgroup (each re-reading all K x T slabs: 60 of that
// kernel's 70 us at N = 391k); as its own launch (grid = T) it costs ~5 us and lets the streaming kernels use as many
// workgroups as the HBM stream wants.  Thread (g = tid / 16, b = tid % 16): 16 groups take every 16th slab.
__global__ __launch_bounds__(256) void k_wsolve(const float *__restrict__ partial_w, int nparts_w,
                                                const double *__restrict__ Cinv, double *__restrict__ tv, int T, int K) {
  __shared__ double part[16][kMaxK];
  __shared__ double sw[kMaxK];
  const int t = blockIdx.x;
  const int b = threadIdx.x & 15, g = threadIdx.x >> 4;
  double s = 0.0;
  if (b < K) {
    int p = g;
    for (; p + 48 < nparts_w; p += 64) {
      const float v0 = partial_w[((size_t)p * K + b) * T + t], v1 = partial_w[((size_t)(p + 16) * K + b) * T + t];
      const float v2 = partial_w[((size_t)(p + 32) * K + b) * T + t], v3 = partial_w[((size_t)(p + 48) * K + b) * T + t];
      s += (double)v0;
      s += (double)v1;
      s += (double)v2;
      s += (double)v3;
    }
    for (; p < nparts_w; p += 16) s += (double)partial_w[((size_t)p * K + b) * T + t];
    part[g][b] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < K) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += part[q][threadIdx.x];
    sw[threadIdx.x] = tot;
  }
  __syncthreads();
  if ((int)threadIdx.x < K) {
    double acc = 0.0;
    for (int c = 0; c < K; ++c) acc += Cinv[threadIdx.x * K + c] * sw[c];
    tv[threadIdx.x * kMaxT + t] = acc;
  }
}

// z = M^-1 r = (r - L tv) / sigma2 with tv = Cinv L^T r from k_wsolve (float64: the subtraction cancels to ~sigma^2 / |K|
// of r along the range of L); partial r.z per workgroup.   K == 0: identity preconditioner (z = r)
template <int TT>
__global__ __launch_bounds__(256) void k_precond(const float *__restrict__ L, const double *__restrict__ tv,
                                                 const float *__restrict__ r, float *__restrict__ z,
                                                 float *__restrict__ partial_rz, long long N, int K, float sigma2) {
  __shared__ double stv[kMaxK * kMaxT];
  __shared__ float sh[4];
  if (K > 0) {
    for (int e = threadIdx.x; e < kMaxK * kMaxT; e += 256) stv[e] = (e / kMaxT < K && e % kMaxT < TT) ? tv[e] : 0.0;
    __syncthreads();
  }
  float acc[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) acc[t] = 0.f;
  const double inv_s = K > 0 ? 1.0 / (double)sigma2 : 1.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
    float rv[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) rv[t] = r[i * TT + t];
    float zv[TT];
    if (K > 0) {
      double corr[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) corr[t] = 0.0;
      for (int b = 0; b < K; ++b) {
        const double lb = (double)L[i * K + b];
#pragma unroll
        for (int t = 0; t < TT; ++t) corr[t] = fma(lb, stv[b * kMaxT + t], corr[t]);
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) zv[t] = (float)(((double)rv[t] - corr[t]) * inv_s);
    } else {
#pragma unroll
      for (int t = 0; t < TT; ++t) zv[t] = rv[t];
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      z[i * TT + t] = zv[t];
      acc[t] = __builtin_fmaf(rv[t], zv[t], acc[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const float s = block_sum(acc[t], sh);
    if (threadIdx.x == 0) partial_rz[(size_t)blockIdx.x * TT + t] = s;
  }
}

// first direction: p = z, rz = sum partial_rz
__global__ __launch_bounds__(256) void k_first_dir(const float *__restrict__ z, float *__restrict__ p,
                                                   const float *__restrict__ partial_rz, int nparts,
                                                   CgState *__restrict__ st, long long N, int T) {
  __shared__ float srz[kMaxT];
  __shared__ float scratch[256];
  if (blockIdx.x == 0) {
    reduce_partials(partial_rz, nparts, T, srz, scratch);
    if ((int)threadIdx.x < T) {
      st->rz[0][threadIdx.x] = srz[threadIdx.x];
      st->resid[threadIdx.x] = 1.0f;
    }
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N * T; i += (long long)gridDim.x * 256) p[i] = z[i];
}

// alpha = rz / pAp (guarded); x += alpha p; r -= alpha Ap; partial |r|^2; partial L^T r for the preconditioner.
// After convergence (st->poll.done) alpha = 0: iterations the host had already enqueued leave x and r untouched.
template <int TT>
__global__ __launch_bounds__(256) void k_update(const float *__restrict__ p, const float *__restrict__ Ap,
                                                const float *__restrict__ partial_pAp, int nparts,
                                                float *__restrict__ x, float *__restrict__ r,
                                                float *__restrict__ partial_rr, const CgState *__restrict__ st,
                                                float *__restrict__ alpha_out, const float *__restrict__ L,
                                                float *__restrict__ partial_w, long long N, int K, float eps,
                                                float stop_after, int cur) {
  constexpr int T = TT;
  __shared__ float spAp[kMaxT];
  __shared__ float salpha[kMaxT];
  __shared__ float sh[4];
  __shared__ float sR[256 * kMaxT];      // doubles as the reduce_partials scratch before the first tile
  __shared__ float sL[256 * kMaxK];
  reduce_partials(partial_pAp, nparts, T, spAp, sR);
  if ((int)threadIdx.x < T) {
    const float s = spAp[threadIdx.x];
    float a = (fabsf(s) > eps) ? st->rz[cur][threadIdx.x] / s : 0.f;
    if (st->resid[threadIdx.x] < stop_after || st->rhs_zero[threadIdx.x] || st->poll.done) a = 0.f;
    salpha[threadIdx.x] = a;
    if (blockIdx.x == 0) alpha_out[threadIdx.x] = a;
  }
  __syncthreads();
  float acc[kMaxT];
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) acc[t] = 0.f;
  float accw = 0.f;
  for (long long tile = blockIdx.x, nt = (N + 255) / 256; tile < nt; tile += gridDim.x) {
    const long long row0 = tile * 256;
    const long long i = row0 + threadIdx.x;
    if (K > 0) {
      __syncthreads();                     // previous tile's ltr_tile is done with sL / sR
      load_L_tile(L, sL, row0, N, K);
    }
    float rvv[kMaxT];
#pragma unroll
    for (int t = 0; t < kMaxT; ++t) rvv[t] = 0.f;
    if (i < N) {
      float pv[TT], apv[TT], xv[TT], rr_[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) {          // all loads of the row first, unconditional
        pv[t] = p[i * TT + t];
        apv[t] = Ap[i * TT + t];
        xv[t] = x[i * TT + t];
        rr_[t] = r[i * TT + t];
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const float a = salpha[t];
        x[i * TT + t] = __builtin_fmaf(a, pv[t], xv[t]);
        const float rv = __builtin_fmaf(-a, apv[t], rr_[t]);
        r[i * TT + t] = rv;
        rvv[t] = rv;
        acc[t] = __builtin_fmaf(rv, rv, acc[t]);
      }
    }
    if (K > 0) {
#pragma unroll
      for (int t = 0; t < kMaxT; ++t) sR[threadIdx.x * kMaxT + t] = rvv[t];
    }
    if (K > 0) {
      __syncthreads();
      accw = ltr_tile(sL, sR, accw);
    }
  }
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) {
    if (t < T) {
      const float s = block_sum(acc[t], sh);
      if (threadIdx.x == 0) partial_rr[(size_t)blockIdx.x * T + t] = s;
    }
  }
  if (K > 0) {
    const int kk = threadIdx.x / kMaxT, t = threadIdx.x % kMaxT;
    if (kk < K && t < T) partial_w[((size_t)blockIdx.x * K + kk) * T + t] = accw;
  }
}

// beta = rz' / rz; p = z + beta p; bookkeeping (block 0): rz <- rz', resid, mean residual, beta history, and — on the
// iterations the host marks with check_now — the convergence decision (st->poll.done), which freezes later iterations.
template <int TT>
__global__ __launch_bounds__(256) void k_direction(const float *__restrict__ z, float *__restrict__ p,
                                                   const float *__restrict__ partial_rz, const float *__restrict__ partial_rr,
                                                   int nparts, int nparts_rr, CgState *__restrict__ st,
                                                   float *__restrict__ beta_out,
                                                   long long N, float eps, int cur, int check_now,
                                                   float tolerance, int iter_count, int stagnation_window,
                                                   const float *__restrict__ x, float *__restrict__ x_best) {
  constexpr int T = TT;
  // Best-iterate safeguard: fp32 CG on a system with cond(Khat) * 1e-6 >~ 1 (N s / sigma^2 beyond a few million) does not
  // merely stall, its recurrence residual can GROW; the iterate with the smallest tested residual is kept in x_best (every
  // workgroup takes the same decision from the same reduced numbers) and returned when the tolerance is never reached.
  __shared__ float srzn[kMaxT];
  __shared__ float srr[kMaxT];
  __shared__ float sbeta[kMaxT];
  __shared__ float sres[kMaxT];
  __shared__ float scratch[256];
  const int was_done = st->poll.done;
  reduce_partials(partial_rz, nparts, T, srzn, scratch);
  reduce_partials(partial_rr, nparts_rr, T, srr, scratch);
  if ((int)threadIdx.x < T) {
    const float rz = st->rz[cur][threadIdx.x];
    sbeta[threadIdx.x] = (fabsf(rz) > eps) ? srzn[threadIdx.x] / rz : 0.f;
    sres[threadIdx.x] = st->rhs_zero[threadIdx.x] ? 0.f : sqrtf(srr[threadIdx.x]);
  }
  __syncthreads();
  if (was_done) return;
  float mres = 0.f;
  for (int t = 0; t < T; ++t) mres += sres[t];
  mres /= (float)T;
  const float snap_prev = st->snap_resid[cur];
  const bool improved = check_now && mres == mres && mres < snap_prev;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
    float pv[TT], zv[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) { pv[t] = p[i * TT + t]; zv[t] = z[i * TT + t]; }
#pragma unroll
    for (int t = 0; t < TT; ++t) p[i * TT + t] = __builtin_fmaf(sbeta[t], pv[t], zv[t]);
    if (improved) {
#pragma unroll
      for (int t = 0; t < TT; ++t) x_best[i * TT + t] = x[i * TT + t];
    }
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      st->snap_resid[cur ^ 1] = improved ? mres : snap_prev;
      st->poll.snap_slot = cur ^ 1;
      st->poll.snap_cur = improved ? mres : snap_prev;
    }
    if ((int)threadIdx.x < T) {
      st->rz[cur ^ 1][threadIdx.x] = srzn[threadIdx.x];
      st->resid[threadIdx.x] = sres[threadIdx.x];
      beta_out[threadIdx.x] = sbeta[threadIdx.x];
    }
    if (threadIdx.x == 0) {
      float m = 0.f;
      for (int t = 0; t < T; ++t) m += sres[t];
      m /= (float)T;
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
      }
    }
  }
}

// x *= |rhs| per column; when the tolerance was never reached and a better tested iterate was saved, return that one
// (the reported mean residual is updated by the host from the same two numbers)
__global__ __launch_bounds__(256) void k_unnormalise(float *__restrict__ x, const float *__restrict__ x_best,
                                                     const CgState *__restrict__ st, long long N, int T) {
  const float snap = st->snap_resid[st->poll.snap_slot];
  const bool use_best = st->poll.done != 1 && snap < st->poll.mean_resid;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256)
    for (int t = 0; t < T; ++t) x[i * T + t] = (use_best ? x_best[i * T + t] : x[i * T + t]) * st->rhs_norm[t];
}

inline int nblocks_for(long long N, int cap = kMaxBlocks) {
  long long b = (N + 255) / 256;      // one 256-row tile per workgroup until the partial-sum slabs are full
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// Pinned host landing zone for the lagged convergence polls (one per host thread; the executor keeps no other
// state between calls).
constexpr int kPollRing = 4;
struct PollCtx {
  CgPoll *host = nullptr;
  hipEvent_t ev[kPollRing];
  bool ok = false;
  int init() {
    if (ok) return 0;
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&host), (kPollRing + 1) * sizeof(CgPoll), hipHostMallocDefault);
    if (e != hipSuccess) return (int)e;
    for (int i = 0; i < kPollRing; ++i) {
      e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
      if (e != hipSuccess) return (int)e;
    }
    ok = true;
    return 0;
  }
};
thread_local PollCtx g_poll;

int apply_operator(const rpgp_operator *op, const float *V, float *out, int T, void *ws, size_t ws_bytes, void *stream) {
  switch (op->kind) {
    case RPGP_OP_FUSED:
      return rpgp_mvm_sym(op->Z, V, out, op->N, op->ldz, T, op->j0, op->j1, op->scale, op->noise, ws, ws_bytes, stream);
    case RPGP_OP_FUSED_PREPARED:
      return rpgp_mvm_sym_prepared(op->prep, V, out, op->N, op->J, T, op->j0, op->j1, op->scale, op->noise, ws, ws_bytes,
                                   stream);
    case RPGP_OP_SKI:
      return rpgp_ski_mvm(op->Z, op->Z, op->grid_params, V, out, op->N, op->N, op->ldz, op->ldz, op->J, op->G, T,
                          op->scale, op->noise, ws, ws_bytes, stream);
    case RPGP_OP_DENSE:
      return rpgp_dense_mvm(op->Kd, V, out, op->N, op->ldk, T, op->noise, stream);
    case RPGP_OP_SYMCACHE:
      return rpgp_symcache_mvm(op->Kd, (size_t)op->ldk, op->G, V, out, op->N, T, op->scale, op->noise, 1, 0, ws, ws_bytes,
                               stream);
    case RPGP_OP_FAMILY:
      return rpgp_family_mvm_sym(op->family, op->Z, V, out, op->N, op->ldz, T, op->scale, op->noise, ws, ws_bytes,
                                 stream);
    default:
      return RPGP_EINVAL;
  }
}

size_t operator_workspace(const rpgp_operator *op, int T) {
  switch (op->kind) {
    case RPGP_OP_FUSED:
    case RPGP_OP_FUSED_PREPARED:
      return rpgp_mvm_sym_workspace_bytes(op->N, T);
    case RPGP_OP_SKI:
      return rpgp_ski_workspace_bytes(op->J, op->G, T);
    case RPGP_OP_DENSE:
      return 256;
    case RPGP_OP_SYMCACHE:
      return rpgp_symcache_workspace_bytes(op->N, T, 1, 0);
    case RPGP_OP_FAMILY:
      return rpgp_family_mvm_workspace_bytes(op->N, op->N, T, 1);
    default:
      return 0;
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

size_t rpgp_mbcg_workspace_bytes(const rpgp_operator *op, int T, int precond_rank) {
  if (!op || T <= 0 || T > kMaxT || precond_rank < 0 || precond_rank > kMaxK) return 0;
  const size_t nt = (size_t)op->N * T * sizeof(float);
  size_t total = 5 * align256(nt);                                             // r, p, z, Ap, x_best
  total += align256(sizeof(CgState));
  total += 3 * align256((size_t)kMaxBlocks * kMaxT * sizeof(float));           // partial pAp / rr / rz
  total += align256((size_t)kMaxBlocksW * kMaxK * kMaxT * sizeof(float));      // partial L^T r
  total += align256((size_t)kMaxK * kMaxT * sizeof(double));                   // tv = Cinv L^T r
  total += 2 * align256((size_t)(kMaxHist + 1) * kMaxT * sizeof(float));       // alpha / beta history
  total += align256(operator_workspace(op, T));
  return total;
}

int rpgp_mbcg_solve(const rpgp_operator *op, const float *rhs, float *x, int T, int max_iter, int min_iter,
                    int hist_len, int check_every, int stagnation_window, float tolerance, int precond_rank,
                    const float *L,
                    const double *Cinv, float precond_sigma2, float *alpha_hist_host, float *beta_hist_host,
                    int *iterations_host, float *mean_resid_host, void *workspace, size_t workspace_bytes, void *stream) {
  if (!op || !rhs || !x || T <= 0 || T > kMaxT || max_iter < 0 || hist_len < 0 || check_every <= 0 ||
      precond_rank < 0 || precond_rank > kMaxK || (precond_rank > 0 && (!L || !Cinv)) || op->N <= 0)
    return RPGP_EINVAL;
  if (hist_len > kMaxHist || (hist_len > 0 && (!alpha_hist_host || !beta_hist_host))) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_mbcg_workspace_bytes(op, T, precond_rank)) return RPGP_EWORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long N = op->N;
  const int K = precond_rank;
  const size_t nt = (size_t)N * T * sizeof(float);

  char *w = reinterpret_cast<char *>(workspace);
  float *r = reinterpret_cast<float *>(w); w += align256(nt);
  float *p = reinterpret_cast<float *>(w); w += align256(nt);
  float *z = reinterpret_cast<float *>(w); w += align256(nt);
  float *Ap = reinterpret_cast<float *>(w); w += align256(nt);
  float *x_best = reinterpret_cast<float *>(w); w += align256(nt);
  CgState *state = reinterpret_cast<CgState *>(w); w += align256(sizeof(CgState));
  float *part_a = reinterpret_cast<float *>(w); w += align256((size_t)kMaxBlocks * kMaxT * sizeof(float));
  float *part_rr = reinterpret_cast<float *>(w); w += align256((size_t)kMaxBlocks * kMaxT * sizeof(float));
  float *part_rz = reinterpret_cast<float *>(w); w += align256((size_t)kMaxBlocks * kMaxT * sizeof(float));
  float *part_w = reinterpret_cast<float *>(w); w += align256((size_t)kMaxBlocksW * kMaxK * kMaxT * sizeof(float));
  double *tv = reinterpret_cast<double *>(w); w += align256((size_t)kMaxK * kMaxT * sizeof(double));
  float *alpha_d = reinterpret_cast<float *>(w); w += align256((size_t)(kMaxHist + 1) * kMaxT * sizeof(float));
  float *beta_d = reinterpret_cast<float *>(w); w += align256((size_t)(kMaxHist + 1) * kMaxT * sizeof(float));
  void *op_ws = w;
  const size_t op_ws_bytes = operator_workspace(op, T);

  // At N = 391k the 17 MB vectors make these kernels HBM streams: up to 1024 workgroups (4 per CU); the K x T
  // preconditioner partials are reduced once per iteration by k_wsolve instead of by every consumer workgroup.
  const int nb = nblocks_for(N);
  const int nbw = nblocks_for(N, kMaxBlocksW);
  const float eps = 1e-30f, stop_after = 1e-10f;
  {
    const int prc = g_poll.init();
    if (prc) return prc;
  }
  CgPoll *hpoll = g_poll.host;

  // normalise right-hand sides, x = 0
  CG_DISPATCH_T(T, hipLaunchKernelGGL((k_coldot<TT>), dim3(nb), dim3(256), 0, st, rhs, rhs, part_a, N));
  hipLaunchKernelGGL(k_normalise, dim3(nb), dim3(256), 0, st, rhs, part_a, nb, r, x, state, N, T);
  // z0 = M^-1 r0, p0 = z0, rz0
  if (K > 0) hipLaunchKernelGGL(k_Ltr, dim3(nbw), dim3(256), 0, st, L, r, part_w, N, T, K);
  if (K > 0) hipLaunchKernelGGL(k_wsolve, dim3(T), dim3(256), 0, st, part_w, nbw, Cinv, tv, T, K);
  CG_DISPATCH_T(T, hipLaunchKernelGGL((k_precond<TT>), dim3(nb), dim3(256), 0, st, L, tv, r, z, part_rz, N, K, precond_sigma2));
  hipLaunchKernelGGL(k_first_dir, dim3(nb), dim3(256), 0, st, z, p, part_rz, nb, state, N, T);
  CG_CHECK(hipGetLastError());

  int it = 0;
  const int n_iter = max_iter < N ? max_iter : (int)N;
  const int min_it = min_iter < n_iter - 1 ? min_iter : n_iter - 1;
  const int n_hist = hist_len < n_iter ? hist_len : n_iter;
  // The convergence decision is taken on the device (k_direction); the host reads it one iteration late from pinned
  // memory, so the next iteration is already queued while it waits and the GPU never idles on the round trip.  The
  // iteration queued past convergence is a no-op on x (alpha = 0).
  int polled_it = -1;                     // iteration whose poll is in flight (-1: none)
  CgPoll last = {1.0f, 0, 0, 0.f, 0, 0, 3.0e38f};
  for (it = 0; it < n_iter; ++it) {
    int rc = apply_operator(op, p, Ap, T, op_ws, op_ws_bytes, stream);
    if (rc) return rc;
    CG_DISPATCH_T(T, hipLaunchKernelGGL((k_coldot<TT>), dim3(nb), dim3(256), 0, st, p, Ap, part_a, N));
    const int slot = it < hist_len ? it : kMaxHist;      // history row (the last row is a scratch slot)
    CG_DISPATCH_T(T, hipLaunchKernelGGL((k_update<TT>), dim3(nbw), dim3(256), 0, st, p, Ap, part_a, nb, x, r, part_rr,
                                        state, alpha_d + (size_t)slot * kMaxT, L, part_w, N, K, eps, stop_after, it & 1));
    if (K > 0) hipLaunchKernelGGL(k_wsolve, dim3(T), dim3(256), 0, st, part_w, nbw, Cinv, tv, T, K);
    CG_DISPATCH_T(T, hipLaunchKernelGGL((k_precond<TT>), dim3(nb), dim3(256), 0, st, L, tv, r, z, part_rz, N, K,
                                        precond_sigma2));
    const bool hist_pending = it < n_hist - 1;
    const bool check_now = it >= min_it && !hist_pending && (it % check_every == 0 || it == n_iter - 1);
    // (k_direction reduces two partial arrays in every workgroup's prologue: it runs best with fewer workgroups)
    CG_DISPATCH_T(T, hipLaunchKernelGGL((k_direction<TT>), dim3(nb < 256 ? nb : 256), dim3(256), 0, st, z, p, part_rz, part_rr, nb, nbw,
                                        state, beta_d + (size_t)slot * kMaxT, N, eps, it & 1, check_now ? 1 : 0, tolerance,
                                        it + 1, stagnation_window, x, x_best));
    if (polled_it >= 0) {                 // consume the previous poll while this iteration runs
      CG_CHECK(hipEventSynchronize(g_poll.ev[polled_it % kPollRing]));
      last = hpoll[polled_it % kPollRing];
      polled_it = -1;
      if (last.done) {
        ++it;
        break;
      }
    }
    if (check_now) {
      CG_CHECK(hipMemcpyAsync(&hpoll[it % kPollRing], &state->poll, sizeof(CgPoll), hipMemcpyDeviceToHost, st));
      CG_CHECK(hipEventRecord(g_poll.ev[it % kPollRing], st));
      polled_it = it;
    }
  }
  hipLaunchKernelGGL(k_unnormalise, dim3(nb), dim3(256), 0, st, x, x_best, state, N, T);
  CG_CHECK(hipMemcpyAsync(&hpoll[kPollRing], &state->poll, sizeof(CgPoll), hipMemcpyDeviceToHost, st));
  CG_CHECK(hipStreamSynchronize(st));
  last = hpoll[kPollRing];
  const int iters_done = last.done ? last.iters : it;
  const int nh = iters_done < hist_len ? iters_done : hist_len;
  if (nh > 0) {   // history rows are kMaxT wide on the device, [hist_len][kMaxT] on the host
    CG_CHECK(hipMemcpyAsync(alpha_hist_host, alpha_d, (size_t)nh * kMaxT * sizeof(float), hipMemcpyDeviceToHost, st));
    CG_CHECK(hipMemcpyAsync(beta_hist_host, beta_d, (size_t)nh * kMaxT * sizeof(float), hipMemcpyDeviceToHost, st));
    CG_CHECK(hipStreamSynchronize(st));
  }
  if (iterations_host) *iterations_host = iters_done;
  if (last.done != 1 && last.snap_cur < last.mean_resid) last.mean_resid = last.snap_cur;   // the saved iterate was returned
  if (mean_resid_host) *mean_resid_host = last.mean_resid;
  if (last.done == 2 || last.mean_resid != last.mean_resid) return RPGP_ENUMERIC;
  return 0;
}

}  // extern "C"
