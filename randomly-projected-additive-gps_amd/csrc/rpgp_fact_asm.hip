// rpgp_fact_asm.hip — the factorised fused symmetric MVM (JT = 20 projections, T = 1, two rows per lane: the launch of
// bench.py and of every T = 1 solve at J = 20) with a HAND-SCHEDULED gfx950 inner loop.
//
// Same algorithm, tiling, slabs and determinism as mvm_fact_kernel<20, 1, 2> (rpgp_kernels.hip; reference semantics:
// gp_models/kernels/memory_efficient_gam_kernel.py:20-30, the `_matmul` that linear_cg calls, fitting/optimizing.py:67-71):
//   exp2(-(a - b)^2) = exp2(-a^2) * exp2(2ab - b^2);  per pair-term  t = a * 2b - b^2,  e = exp2(t),  K += e * Ea.
// What differs is who schedules the 64-step rotation loop (tools/gen_fact_asm.py -> rpgp_fact_asm_loop.inc):
//   * packed lanes run over the lane's TWO ROWS (column operands broadcast with op_sel): no horizontal add;
//   * the 64 column records of a subtile are followed in LDS by a copy of the first 63, so "column (lane + s) mod 64" is a
//     running pointer plus immediate offsets: no per-step v_and_or / v_lshlrev / v_mul / v_add;
//   * the column's v rides in the record (one pointer), every ds_read of step s+1 is issued a full step ahead, K-FMAs trail
//     their exponentials by one quad (no s_nop, no dependent-issue bubble), the finish of a step rides inside the next.
// Per step and lane: 20 + 40 + 20 floor instructions + 4 (row product, two transposed FMAs, DPP rotation) = 12.93 issue
// cycles per 64 pair-terms by the measured costs against 13.3 for the compiler's loop.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rpgp.h"
#include "rpgp_internal.h"
#include "rpgp_fact_asm_loop.inc"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v32f __attribute__((ext_vector_type(32)));

constexpr int kJT = 20;
constexpr int kBR = 512;                 // rows per workgroup: 4 waves x 2 rows per lane
constexpr int kSC = 128;                 // columns staged per barrier round = 2 rotation subtiles
constexpr int kRecFloats = 44;           // 10 quads {2b_e, -b_e^2, 2b_o, -b_o^2} + v_col + 3 pad  (176 B, 176/16 odd)
constexpr int kSubRecs = 127;            // 64 records + a copy of the first 63
constexpr int kLdsFloats = (2 * kSubRecs + 1) * kRecFloats;   // + one record: the loop's look-ahead reads one past the end

__device__ __forceinline__ void wg_to_tile_sym(int lin, int N, int BR, int chunk, rpgp_internal::Taper tp, int &rb,
                                               int &kchunk, int &chunk_b) {
  int b = 0, acc = 0;
  for (;;) {
    const int cbk = rpgp_internal::taper_chunk(b, chunk, tp.tb1, tp.tb2, tp.tb3);
    const int cb = (N - b * BR + cbk - 1) / cbk;
    if (lin < acc + cb) {
      chunk_b = cbk;
      break;
    }
    acc += cb;
    ++b;
  }
  rb = b;
  kchunk = lin - acc;
}

__global__ __launch_bounds__(256) void mvm_fact_asm_kernel(const v2f *__restrict__ rowdat, const v4f *__restrict__ coldat4,
                                                           const float *__restrict__ V, float *__restrict__ slabR,
                                                           float *__restrict__ slabT, int N, int ldv, int t0,
                                                           int chunk_cols, rpgp_internal::Taper taper, int accumulate,
                                                           int w0, int rb_first, int slab_row0, int slab_rows) {
  __shared__ __attribute__((aligned(16))) float sB[kLdsFloats];
  __shared__ __attribute__((aligned(16))) float sT[4 * kSC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int rb, kchunk, chunk_b;
  wg_to_tile_sym(blockIdx.x + w0, N, kBR, chunk_cols, taper, rb, kchunk, chunk_b);
  const int r0 = rb * kBR;
  const long long cb = (long long)r0 + (long long)kchunk * chunk_b;
  if (cb >= N) return;
  const int c_begin = (int)cb;
  const int c_end = (c_begin + chunk_b < N) ? c_begin + chunk_b : N;

  // row side: A[j] = {a_row0, a_row1}, E[j] = {Ea_row0, Ea_row1}; rows past N: Ea = 0 -> K = 0 (loads from a clamped row:
  // unconditional, so they are issued together)
  v32f A0, E0;
  v8f A1, E1;
  v2f vrow;
  {
    const int row0 = r0 + wave * 128 + lane, row1 = row0 + 64;
    const int rc0 = row0 < N ? row0 : N - 1, rc1 = row1 < N ? row1 : N - 1;
    const float m0 = row0 < N ? 1.f : 0.f, m1 = row1 < N ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < kJT; ++j) {
      const v2f x0 = rowdat[(size_t)rc0 * kJT + j], x1 = rowdat[(size_t)rc1 * kJT + j];
      if (j < 16) {
        A0[2 * j] = x0.x; A0[2 * j + 1] = x1.x;
        E0[2 * j] = x0.y * m0; E0[2 * j + 1] = x1.y * m1;
      } else {
        A1[2 * (j - 16)] = x0.x; A1[2 * (j - 16) + 1] = x1.x;
        E1[2 * (j - 16)] = x0.y * m0; E1[2 * (j - 16) + 1] = x1.y * m1;
      }
    }
    vrow.x = V[(size_t)rc0 * ldv + t0] * m0;
    vrow.y = V[(size_t)rc1 * ldv + t0] * m1;
  }
  v2f accR = {0.f, 0.f};
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)sB;      // LDS byte address

  for (int c0 = c_begin; c0 < c_end; c0 += kSC) {
    __syncthreads();
    {
      // staging: thread = (column, half record); a column's 20 {2b, -b^2} pairs are 160 contiguous bytes of coldat
      const int c = tid >> 1, h = tid & 1;
      const int col = c0 + c;
      const bool cv = col < c_end;
      const int colc = cv ? col : N - 1;
      v4f q[5];
#pragma unroll
      for (int p = 0; p < 5; ++p) q[p] = coldat4[(size_t)colc * (kJT / 2) + 5 * h + p];
      const float vv = V[(size_t)colc * ldv + t0];
      if (!cv) {
#pragma unroll
        for (int p = 0; p < 5; ++p) q[p] = v4f{0.f, -1.0e30f, 0.f, -1.0e30f};       // padded column: exp2(-1e30) = 0
      }
      const int sub = c >> 6, k = c & 63;
      float *rec = sB + (sub * kSubRecs + k) * kRecFloats + 20 * h;
#pragma unroll
      for (int p = 0; p < 5; ++p) *reinterpret_cast<v4f *>(rec + 4 * p) = q[p];
      if (h) rec[20] = cv ? vv : 0.f;                                               // float 40 of the record
      if (k < 63) {
        float *dup = rec + 64 * kRecFloats;
#pragma unroll
        for (int p = 0; p < 5; ++p) *reinterpret_cast<v4f *>(dup + 4 * p) = q[p];
        if (h) dup[20] = cv ? vv : 0.f;
      }
    }
    __syncthreads();
    const int ncol = c_end - c0;
    const int nsub = ncol >= kSC ? kSC / 64 : (ncol + 63) / 64;
    for (int sub = 0; sub < nsub; ++sub) {
      unsigned ptr = lds_base + (unsigned)((sub * kSubRecs + lane) * kRecFloats * 4);
      float accT;
      int cnt;
      asm volatile(
          "s_waitcnt lgkmcnt(0)\n" RPGP_FACT_ASM_LOOP
          : "+{v[90:91]}"(accR), "+{v94}"(ptr), "={v95}"(accT), [cnt] "=s"(cnt)
          : "{v[10:41]}"(A0), "{v[42:49]}"(A1), "{v[50:81]}"(E0), "{v[82:89]}"(E1), "{v[92:93]}"(vrow)
          : RPGP_FACT_ASM_CLOBBERS, "v96", "memory");
      // (subtiles inside the row block: every pair of the block is swept from both sides, the transposed sums are dropped)
      sT[wave * kSC + sub * 64 + lane] = accT;
    }
    __syncthreads();
    {
      const int col = c0 + tid;
      if (tid < kSC && col < c_end && col >= r0 + kBR) {
        const float sum = sT[0 * kSC + tid] + sT[1 * kSC + tid] + sT[2 * kSC + tid] + sT[3 * kSC + tid];
        float *dst = slabT + ((size_t)(rb - rb_first) * N + col) * ldv + t0;
        *dst = accumulate ? *dst + sum : sum;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = r0 + wave * 128 + r * 64 + lane;
    if (row < N) {
      float *dst = slabR + ((size_t)kchunk * slab_rows + (row - slab_row0)) * ldv + t0;
      const float a = r == 0 ? accR.x : accR.y;
      *dst = accumulate ? *dst + a : a;
    }
  }
}

// ---- the same loop for J-SLICES (round 6) ------------------------------------------------------------------------------------
// north_star shards the J = 20 additive terms over the ranks: 10 / 5 / 3 - 2 projections per rank at 2 / 4 / 8 GPUs.  The
// compiler-scheduled piece kernels pay the per-pair work (column read, product with v, transposed accumulator, rotation) for a
// handful of projections with no software pipeline: 24 issue cycles per 64 pair-terms at 3 projections against 12.93 for the
// J = 20 loop.  tools/gen_fact_asm.py generates the same schedule for 2 / 3 / 4 / 5 / 8 / 10 projections
// (RPGP_FACT_ASM_LOOP_JT<n>): exact projection counts (an odd last projection is a 64-bit read and a half slot), records
// requested 8 / 4 / 4 / 2 / 2 / 1 steps ahead (a 3-projection step is ~90 issue cycles: one step of look-ahead does not cover
// the LDS latency), as many steps per loop trip.  Everything around the loop (tile mapping, slabs, determinism) is the kernel
// above with a column record of [JT pairs {2b, -b^2}][v][pad] = 12 .. 28 floats.
template <int JT> struct Thin;
#define RPGP_THIN_TRAITS(JT_)                                                            \
  template <> struct Thin<JT_> {                                                         \
    static constexpr int rec = RPGP_FACT_ASM_REC_FLOATS_JT##JT_;                         \
    static constexpr int depth = RPGP_FACT_ASM_DEPTH_JT##JT_;                            \
    typedef float rowvec __attribute__((ext_vector_type(JT_ == 10 ? 32 : 2 * JT_)));     \
  }
RPGP_THIN_TRAITS(2);
RPGP_THIN_TRAITS(3);
RPGP_THIN_TRAITS(4);
RPGP_THIN_TRAITS(5);
RPGP_THIN_TRAITS(8);
RPGP_THIN_TRAITS(10);
#undef RPGP_THIN_TRAITS

template <int JT>
__global__ __launch_bounds__(256) void mvm_fact_asm_thin_kernel(const v2f *__restrict__ rowdat, const v2f *__restrict__ coldat,
                                                                const float *__restrict__ V, float *__restrict__ slabR,
                                                                float *__restrict__ slabT, int N, int J, int j0, int ldv, int t0,
                                                                int chunk_cols, rpgp_internal::Taper taper, int accumulate,
                                                                int w0, int rb_first, int slab_row0, int slab_rows) {
  constexpr int RF = Thin<JT>::rec;
  // columns staged per barrier round.  Measured: 128, 256 (and 512, where the LDS then caps the workgroups per CU below what
  // the registers allow: slower) time the same — the barriers are not what a thin step waits for; 128 keeps the image at
  // 7 - 15 KB so that the register count alone (70 - 111: four to seven waves per SIMD) sets the occupancy.  RPGP_THIN_SC
  // (compile time) is there for the A/B.
#ifndef RPGP_THIN_SC
#define RPGP_THIN_SC 128
#endif
  constexpr int SC = RPGP_THIN_SC;
  constexpr int NSUB = SC / 64;
  // NSUB subtile images of 64 records + a copy of the first 63, + the records the look-ahead of the last steps touches
  constexpr int kLds = (NSUB * kSubRecs + Thin<JT>::depth) * RF;
  __shared__ __attribute__((aligned(16))) float sB[kLds];
  __shared__ __attribute__((aligned(16))) float sT[4 * SC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int rb, kchunk, chunk_b;
  wg_to_tile_sym(blockIdx.x + w0, N, kBR, chunk_cols, taper, rb, kchunk, chunk_b);
  const int r0 = rb * kBR;
  const long long cb = (long long)r0 + (long long)kchunk * chunk_b;
  if (cb >= N) return;
  const int c_begin = (int)cb;
  const int c_end = (c_begin + chunk_b < N) ? c_begin + chunk_b : N;

  // row side: A[j] = {a_row0, a_row1}, E[j] = {Ea_row0, Ea_row1} of the slice's projections; rows past N: Ea = 0 -> K = 0
  // (the loops of 2 - 8 projections have a compact register map — 70 - 111 registers, four to seven waves per SIMD instead of
  //  the three of the J = 20 map; A and E are vectors of exactly 2 JT registers, pinned where the generator says.  Measured:
  //  the occupancy changes nothing, 0.446 ms per 3-projection sweep at N = 50000 with either map — a thin step is bound by its
  //  issue slots, 93 modelled cycles of 112 measured, of which the per-column work that does not shrink with the slice — the
  //  rotation, the two products — is 17)
  typename Thin<JT>::rowvec A0, E0;
  constexpr int NVEC = JT == 10 ? 16 : JT;
  v2f vrow;
  {
    const int row0 = r0 + wave * 128 + lane, row1 = row0 + 64;
    const int rc0 = row0 < N ? row0 : N - 1, rc1 = row1 < N ? row1 : N - 1;
    const float m0 = row0 < N ? 1.f : 0.f, m1 = row1 < N ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < NVEC; ++j) {
      if (j < JT) {
        const v2f x0 = rowdat[(size_t)rc0 * J + j0 + j], x1 = rowdat[(size_t)rc1 * J + j0 + j];
        A0[2 * j] = x0.x; A0[2 * j + 1] = x1.x;
        E0[2 * j] = x0.y * m0; E0[2 * j + 1] = x1.y * m1;
      } else {
        A0[2 * j] = 0.f; A0[2 * j + 1] = 0.f;
        E0[2 * j] = 0.f; E0[2 * j + 1] = 0.f;
      }
    }
    vrow.x = V[(size_t)rc0 * ldv + t0] * m0;
    vrow.y = V[(size_t)rc1 * ldv + t0] * m1;
  }
  v2f accR = {0.f, 0.f};
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)sB;      // LDS byte address
  // (the look-ahead of a subtile's last steps reads up to `depth` records past its image: the next image's records, or — behind
  //  the second image — the tail, which must hold finite numbers once: what is read there is never consumed)
  for (int e = tid; e < Thin<JT>::depth * RF; e += 256) sB[NSUB * kSubRecs * RF + e] = 0.f;

  for (int c0 = c_begin; c0 < c_end; c0 += SC) {
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < SC; cc += 128) {
      // staging: thread = (column, parity of the projection); a projection's {2b, -b^2} pair is one 8-byte store
      const int c = cc + (tid >> 1), h = tid & 1;
      const int col = c0 + c;
      const bool cv = col < c_end;
      const int colc = cv ? col : N - 1;
      constexpr int NH = (JT + 1) / 2;
      v2f pq[NH];
#pragma unroll
      for (int i = 0; i < NH; ++i) {
        const int j = 2 * i + h;
        pq[i] = coldat[(size_t)colc * J + j0 + (j < JT ? j : JT - 1)];
        if (!cv) pq[i] = v2f{0.f, -1.0e30f};                                    // padded column: exp2(-1e30) = 0
      }
      const float vv = V[(size_t)colc * ldv + t0];
      const int sub = c >> 6, k = c & 63;
      float *rec = sB + (sub * kSubRecs + k) * RF;
#pragma unroll
      for (int i = 0; i < NH; ++i) {
        const int j = 2 * i + h;
        if (j < JT) {
          *reinterpret_cast<v2f *>(rec + 2 * j) = pq[i];
          if (k < 63) *reinterpret_cast<v2f *>(rec + 64 * RF + 2 * j) = pq[i];
        }
      }
      if (h == 0) {
        rec[2 * JT] = cv ? vv : 0.f;
        if (k < 63) rec[64 * RF + 2 * JT] = cv ? vv : 0.f;
      }
    }
    __syncthreads();
    const int ncol = c_end - c0;
    const int nsub = ncol >= SC ? NSUB : (ncol + 63) / 64;
    for (int sub = 0; sub < nsub; ++sub) {
      unsigned ptr = lds_base + (unsigned)((sub * kSubRecs + lane) * RF * 4);
      float accT;
      int cnt;
#define RPGP_THIN_ASM(N_)                                                                                            \
      asm volatile("s_waitcnt lgkmcnt(0)\n" RPGP_FACT_ASM_LOOP_JT##N_                                                  \
                   : RPGP_FACT_ASM_CACCR_JT##N_(accR), RPGP_FACT_ASM_CPTR_JT##N_(ptr), RPGP_FACT_ASM_CACCT_JT##N_(accT), \
                     [cnt] "=s"(cnt)                                                                                   \
                   : RPGP_FACT_ASM_CA_JT##N_(A0), RPGP_FACT_ASM_CE_JT##N_(E0), RPGP_FACT_ASM_CVROW_JT##N_(vrow)         \
                   : RPGP_FACT_ASM_CLOB_JT##N_, "memory")
      if constexpr (JT == 2) RPGP_THIN_ASM(2);
      else if constexpr (JT == 3) RPGP_THIN_ASM(3);
      else if constexpr (JT == 4) RPGP_THIN_ASM(4);
      else if constexpr (JT == 5) RPGP_THIN_ASM(5);
      else if constexpr (JT == 8) RPGP_THIN_ASM(8);
      else                                                     // 10 projections: the five-quad form of the J = 20 loop and its map
        asm volatile("s_waitcnt lgkmcnt(0)\n" RPGP_FACT_ASM_LOOP_JT10
                     : "+{v[90:91]}"(accR), "+{v94}"(ptr), "={v95}"(accT), [cnt] "=s"(cnt)
                     : "{v[10:41]}"(A0), "{v[50:81]}"(E0), "{v[92:93]}"(vrow)
                     : RPGP_FACT_ASM_CLOBBERS, "v96", "memory");
#undef RPGP_THIN_ASM
      sT[wave * SC + sub * 64 + lane] = accT;
    }
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < SC; cc += 256) {
      const int ct = cc + tid, col = c0 + ct;
      if (ct < SC && col < c_end && col >= r0 + kBR) {
        const float sum = sT[0 * SC + ct] + sT[1 * SC + ct] + sT[2 * SC + ct] + sT[3 * SC + ct];
        float *dst = slabT + ((size_t)(rb - rb_first) * N + col) * ldv + t0;
        *dst = accumulate ? *dst + sum : sum;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = r0 + wave * 128 + r * 64 + lane;
    if (row < N) {
      float *dst = slabR + ((size_t)kchunk * slab_rows + (row - slab_row0)) * ldv + t0;
      const float a = r == 0 ? accR.x : accR.y;
      *dst = accumulate ? *dst + a : a;
    }
  }
}

}  // namespace

namespace rpgp_internal {

int launch_mvm_fact_asm(const void *rowdat, const void *coldat, const float *V, float *slabR, float *slabT, int N, int ldv,
                        int t0, int chunk_cols, Taper taper, int accumulate, int w0, int nwg, int rb_first, int slab_row0,
                        int slab_rows, hipStream_t st) {
  hipLaunchKernelGGL(mvm_fact_asm_kernel, dim3((unsigned)nwg), dim3(256), 0, st, reinterpret_cast<const v2f *>(rowdat),
                     reinterpret_cast<const v4f *>(coldat), V, slabR, slabT, N, ldv, t0, chunk_cols, taper, accumulate, w0,
                     rb_first, slab_row0, slab_rows);
  return (int)hipGetLastError();
}

bool fact_asm_thin_supported(int jt) { return jt == 2 || jt == 3 || jt == 4 || jt == 5 || jt == 8 || jt == 10; }

int launch_mvm_fact_asm_thin(int jt, const void *rowdat, const void *coldat, const float *V, float *slabR, float *slabT, int N,
                             int J, int j0, int ldv, int t0, int chunk_cols, Taper taper, int accumulate, int w0, int nwg,
                             int rb_first, int slab_row0, int slab_rows, hipStream_t st) {
  if (!fact_asm_thin_supported(jt)) return RPGP_EINVAL;
#define RPGP_THIN_LAUNCH(JT_)                                                                                                \
  hipLaunchKernelGGL((mvm_fact_asm_thin_kernel<JT_>), dim3((unsigned)nwg), dim3(256), 0, st, reinterpret_cast<const v2f *>(rowdat), \
                     reinterpret_cast<const v2f *>(coldat), V, slabR, slabT, N, J, j0, ldv, t0, chunk_cols, taper, accumulate, w0,  \
                     rb_first, slab_row0, slab_rows)
  switch (jt) {
    case 2: RPGP_THIN_LAUNCH(2); break;
    case 3: RPGP_THIN_LAUNCH(3); break;
    case 4: RPGP_THIN_LAUNCH(4); break;
    case 5: RPGP_THIN_LAUNCH(5); break;
    case 8: RPGP_THIN_LAUNCH(8); break;
    default: RPGP_THIN_LAUNCH(10); break;
  }
#undef RPGP_THIN_LAUNCH
  return (int)hipGetLastError();
}

}  // namespace rpgp_internal
