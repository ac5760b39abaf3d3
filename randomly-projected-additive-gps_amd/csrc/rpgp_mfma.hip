// rpgp_mfma.hip — matrix-core form of the factorised symmetric MVM sweep (gfx950 only).
//
// The VALU kernel (mvm_fact_kernel, rpgp_kernels.hip) spends three vector issues per pair-term:
//     t = fma(a, 2b, -b^2);  e = v_exp_f32(t);  K = fma(e, exp2(-a^2), K)
// and runs at 98 % of its VALU-issue bound.  The first of the three is a rank-2 outer product
//     t[i][c] = (2 a_i) * b_c + (-a_i^2) * 1
// which is exactly one `v_mfma_f32_32x32x2_f32` (exact fp32: bit-for-bit the k-ordered fmaf chain) per projection per
// 32 x 32 pair tile — 64 matrix-pipe cycles for 1024 pair-terms on a pipe that was idle.  What stays on the VALU per
// pair-term is  e = v_exp_f32(t)  and  K = fma(e, exp2(-b_c^2), K)  (the factor that used to multiply on the row side
// now sits in the exponent, the column factor became the multiplier because in the MFMA result layout a lane owns ONE
// column and 16 rows: one multiplier register per projection per tile).
//
// Result layout of the 32x32 tile (cdna_hip_programming.md §3): lane l holds column c = l & 31 and rows
// rho(r, h) = (r & 3) + 8 (r >> 2) + 4 h, r = 0..15, h = l >> 5.  Hence
//   * transposed product  outT[c] = sum_i K[i][c] v[i]   is an in-lane sum over the 16 registers (+ the other half);
//   * row product         outR[i] = sum_c K[i][c] v[c]   is a per-lane partial  accR[r] += K[r] v[c]  kept in registers
//     for the whole column chunk and reduced across the 32 lanes ONCE per workgroup;
// the 64 DPP rotations per subtile of the VALU kernel are gone, each unordered pair is still evaluated once.
//
// Work decomposition, slabs and the fixed-order reduce kernel are those of the VALU kernel with BR = 128 rows per
// workgroup (4 waves x one 32-row tile); results are deterministic (no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rpgp_internal.h"

namespace {

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// same numbering as rpgp_kernels.hip: workgroups row block by row block, block b owns ceil((N - b*BR) / chunk) chunks
__device__ __forceinline__ void wg_to_tile(int lin, int N, int BR, int chunk, int &rb, int &kchunk) {
  int b = 0, acc = 0;
  for (;;) {
    const int cb = (N - b * BR + chunk - 1) / chunk;
    if (lin < acc + cb) break;
    acc += cb;
    ++b;
  }
  rb = b;
  kchunk = lin - acc;
}

// LDS record stride of one column's JT values: multiples of 4 floats are padded so that (stride / 4) is odd and the
// per-lane ds_read_b128 of 16 consecutive columns hits 16 distinct 4-bank slots (MI355X_MICROARCH.md §LDS)
template <int JT> struct RecStride {
  static constexpr int v = (JT % 4 == 0) ? ((JT / 4) % 2 == 1 ? JT : JT + 4) : JT;
};

constexpr int kSC = 128;   // columns staged in LDS per sub-chunk (4 MFMA tiles per wave and stage)

template <int JT, int TT>
__global__ __launch_bounds__(256, 3) void mvm_mfma_kernel(const float2v *__restrict__ rowtab,   // {2a, -a^2}
                                                       const float2v *__restrict__ coltab,   // {b, exp2(-b^2)}
                                                       const float *__restrict__ V, float *__restrict__ slabR,
                                                       float *__restrict__ slabT, int N, int J, int ldv, int j0, int t0,
                                                       int tcnt, int chunk_cols, int accumulate, int w0, int rb_first,
                                                       int slab_row0, int slab_rows) {
  constexpr int BR = rpgp_internal::kMfmaBR;
  constexpr int SC = kSC;
  constexpr int STR = RecStride<JT>::v;
  __shared__ __attribute__((aligned(16))) float sB[(SC + 1) * STR];   // record SC: all ones (the k = 1 row of B)
  __shared__ __attribute__((aligned(16))) float sE[SC * STR];
  __shared__ __attribute__((aligned(16))) float sV[SC * TT];
  __shared__ __attribute__((aligned(16))) float sT[8 * SC * TT];      // [(wave, half)][column][t]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  int rb, kchunk;
  wg_to_tile(blockIdx.x + w0, N, BR, chunk_cols, rb, kchunk);
  const int r0 = rb * BR;
  const long long cb = (long long)r0 + (long long)kchunk * chunk_cols;
  if (cb >= N) return;
  const int c_begin = (int)cb;
  const int c_end = (c_begin + chunk_cols < N) ? c_begin + chunk_cols : N;

  // A operand of projection j: lane (i = l & 31, k = l >> 5) holds 2 a_i (k = 0) or -a_i^2 (k = 1).
  // Rows beyond N: -a^2 = -1e30 -> t = -1e30 -> e = 0.
  float A[JT];
  {
    const int arow = r0 + wave * 32 + l31;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      float2v x = {0.f, -1.0e30f};
      if (arow < N) x = rowtab[(size_t)arow * J + j0 + j];
      A[j] = half ? x.y : x.x;
    }
  }
  // v of this lane's 16 result rows, and their row-product partials
  float vrow[16][TT], accR[16][TT];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = r0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      vrow[r][t] = (row < N && t < tcnt) ? V[(size_t)row * ldv + t0 + t] : 0.f;
      accR[r][t] = 0.f;
    }
  }
  if (tid < STR) sB[SC * STR + tid] = 1.0f;

  for (int c0 = c_begin; c0 < c_end; c0 += SC) {
    __syncthreads();
    {
      // two threads per column: thread (col, part) stages projections [part * JH, ...) of column c0 + col
      constexpr int JH = (JT + 1) / 2;
      const int colx = tid >> 1, part = tid & 1;
      const int col = c0 + colx;
      const bool cv = col < c_end;
#pragma unroll
      for (int q = 0; q < JH; ++q) {
        const int j = part * JH + q;
        if (j < JT) {
          float2v x = {0.f, 0.f};                       // padded columns: multiplier 0 -> K = 0
          if (cv) x = coltab[(size_t)col * J + j0 + j];
          sB[colx * STR + j] = x.x;
          sE[colx * STR + j] = x.y;
        }
      }
      if (tid < SC) {
        const int c2 = c0 + tid;
#pragma unroll
        for (int t = 0; t < TT; ++t)
          sV[tid * TT + t] = (c2 < c_end && t < tcnt) ? V[(size_t)c2 * ldv + t0 + t] : 0.f;
      }
    }
    __syncthreads();
    const int ncol = c_end - c0;
    const int ntile = ncol >= SC ? SC / 32 : (ncol + 31) / 32;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int tile = 0; tile < ntile; ++tile) {
      const int ct = tile * 32 + l31;
      const bool doT = (c0 + tile * 32 >= r0 + BR);
      const float *pb = sB + (half ? SC : ct) * STR;    // k = 1 lanes read the ones record
      const float *pe = sE + ct * STR;
      f32x16 K;
#pragma unroll
      for (int r = 0; r < 16; ++r) K[r] = 0.f;
      // Software pipeline of depth 2 over the projections of one tile: while the VALU works through the 16 results of
      // projection j the matrix pipe computes projection j + 1.  The scheduling barriers keep hipcc from hoisting all JT
      // MFMAs (and their JT x 16 result registers) to the top of the tile.  (Carrying the pipeline across tiles makes
      // hipcc coalesce both result buffers into one register range and serialise MFMA -> exp; the one exposed MFMA
      // latency per tile is covered by the other waves of the SIMD.)
      f32x16 Dn = __builtin_amdgcn_mfma_f32_32x32x2f32(A[0], pb[0], zero, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const f32x16 D = Dn;
        if (j + 1 < JT) Dn = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j + 1], pb[j + 1], zero, 0, 0, 0);
        const float eb = pe[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) K[r] = __builtin_fmaf(fast_exp2(D[r]), eb, K[r]);
        // issue order inside the stage: the next MFMA first, then exp / fma pairs skewed by one (the fma of result r
        // after the exp of result r + 1: a transcendental's result needs a wait state before a plain VALU op may read
        // it, and an `s_nop` there would cost an issue slot per pair-term)
        if (j + 1 < JT) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
#pragma unroll
        for (int r = 0; r < 14; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      float vc[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) vc[t] = sV[ct * TT + t];
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int t = 0; t < TT; ++t) accR[r][t] = __builtin_fmaf(K[r], vc[t], accR[r][t]);
      if (doT) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          float s = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) s = __builtin_fmaf(K[r], vrow[r][t], s);
          sT[((wave * 2 + half) * SC + ct) * TT + t] = s;
        }
      }
    }
    __syncthreads();
    if (tid < SC) {
      const int col = c0 + tid;
      if (col < c_end && col >= r0 + BR) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          if (t < tcnt) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += sT[(q * SC + tid) * TT + t];
            float *dst = slabT + ((size_t)(rb - rb_first) * N + col) * ldv + t0 + t;
            *dst = accumulate ? *dst + sum : sum;
          }
        }
      }
    }
  }

  // row products: butterfly over the 32 lanes of each half (fixed order), then lane (l & 31) == r keeps row rho(r, h)
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = accR[r][t];
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) x += __shfl_xor(x, m, 64);
      if (l31 == r) mine = x;
    }
    if (l31 < 16 && t < tcnt) {
      const int row = r0 + wave * 32 + (l31 & 3) + 8 * (l31 >> 2) + 4 * half;
      if (row < N) {
        float *dst = slabR + ((size_t)kchunk * slab_rows + (row - slab_row0)) * ldv + t0 + t;
        *dst = accumulate ? *dst + mine : mine;
      }
    }
  }
}

template <int JT>
int launch_jt(int tt, const float2v *rowtab, const float2v *coltab, const float *V, float *slabR, float *slabT, int N,
              int J, int ldv, int j0, int t0, int tcnt, int chunk_cols, int accumulate, int w0, int nwg, int rb_first,
              int slab_row0, int slab_rows, hipStream_t st) {
  dim3 grid(nwg), block(256);
  switch (tt) {
    case 1:
      hipLaunchKernelGGL((mvm_mfma_kernel<JT, 1>), grid, block, 0, st, rowtab, coltab, V, slabR, slabT, N, J, ldv, j0, t0,
                         tcnt, chunk_cols, accumulate, w0, rb_first, slab_row0, slab_rows);
      break;
    default:
      return 10001;   // RPGP_EINVAL
  }
  return (int)hipGetLastError();
}

}  // namespace

namespace rpgp_internal {

bool mfma_supported(int jt, int tt) {
  if (tt != 1) return false;
  switch (jt) {
    case 20: case 10: case 8: case 5: case 4: case 3: case 2: case 1: return true;
    default: return false;
  }
}

int launch_mvm_mfma(int jt, int tt, const void *rowtab_, const void *coltab_, const float *V, float *slabR, float *slabT,
                    int N, int J, int ldv, int j0, int t0, int tcnt, int chunk_cols, int accumulate, int w0, int nwg,
                    int rb_first, int slab_row0, int slab_rows, hipStream_t st) {
  const float2v *rowtab = reinterpret_cast<const float2v *>(rowtab_);
  const float2v *coltab = reinterpret_cast<const float2v *>(coltab_);
#define RPGP_MFMA_CASE(JT_)                                                                                            \
  case JT_:                                                                                                            \
    return launch_jt<JT_>(tt, rowtab, coltab, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, chunk_cols, accumulate, w0, nwg, \
                          rb_first, slab_row0, slab_rows, st)
  switch (jt) {
    RPGP_MFMA_CASE(20);
    RPGP_MFMA_CASE(10);
    RPGP_MFMA_CASE(8);
    RPGP_MFMA_CASE(5);
    RPGP_MFMA_CASE(4);
    RPGP_MFMA_CASE(3);
    RPGP_MFMA_CASE(2);
    RPGP_MFMA_CASE(1);
    default: return 10001;
  }
#undef RPGP_MFMA_CASE
}

}  // namespace rpgp_internal
