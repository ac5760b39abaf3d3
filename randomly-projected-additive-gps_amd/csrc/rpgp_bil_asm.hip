// rpgp_bil_asm.hip — the symmetric bilinear-derivative sweep (JT = 20 projections, up to 12 right-hand-side slots, two rows
// per lane: the backward pass of one optimiser step in the CG regime) with a HAND-SCHEDULED gfx950 inner loop.
//
// Same algorithm, tiling, slabs and determinism as bilinear_sym_kernel<20, 12, true> (rpgp_kernels.hip; SURVEY.md A.2;
// reference semantics: GAMFunction.backward, gp_models/kernels/memory_efficient_gam_kernel.py:33-59, reached from
// loss.backward() at fitting/optimizing.py:72 through GPyTorch's `_quad_form_derivative`):
//   S(i, i') = sum_t L[i][t] R[i'][t] + R[i][t] L[i'][t];   gZ[i][j] += S e_j d_j,  gZ[i'][j] -= S e_j d_j,  gs += S sum_j e_j
// with d_j = z_ij - z_i'j (pre-scaled), e_j = exp2(-d_j^2); every unordered pair once, the transposed half through 21
// DPP-rotated travelling accumulators.  What differs is who schedules the 64-step loop (tools/gen_bil_asm.py ->
// rpgp_bil_asm_loop.inc; the generator's --selftest runs the text on a CPU interpreter against the direct formula): packed
// instructions over the lane's two rows, a three-deep software pipeline over the projections that runs across the step
// boundary, S formed one step ahead from two partial sums, every LDS read a full step ahead of its use.  The compiler's
// loop holds 252 VGPRs at two waves per SIMD and issues 57-73 % of its own instruction stream (PMC, round 3).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rpgp.h"
#include "rpgp_internal.h"
#include "rpgp_bil_asm_loop.inc"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));

constexpr int kJT = 20, kTT = 12;
constexpr int kBR = 512;
constexpr int kW = kJT + 1;                        // slab width: 20 gradient columns + the scale column
constexpr int kRec = kJT + 2 * kTT;                // 44 floats per column record {z[20], L[12], R[12]}: 176 B, 176/16 odd
constexpr int kNRec = 130;                         // 64 records + a copy of the first 66 (the loop's look-ahead)

__device__ __forceinline__ void wg_to_tile_sym(int lin, int N, int BR, int chunk, int &rb, int &kchunk) {
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

// LDS-DMA of one dword per lane (wave-instruction: 64 consecutive floats at `lds_dst`); the caller waits with s_waitcnt vmcnt
__device__ __forceinline__ void glds_dword(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// Zs: pre-scaled copy of the projected coordinates (N x 20, row stride 20); L, Rm: N x T (T <= 12)
__global__ __launch_bounds__(256, 2) void bilinear_sym_asm_kernel(const float *__restrict__ Zs, const float *__restrict__ L,
                                                                  const float *__restrict__ Rm, float *__restrict__ slabR,
                                                                  float *__restrict__ slabT, int N, int T, int chunk_cols) {
  __shared__ __attribute__((aligned(16))) float sC[kNRec * kRec];
  __shared__ __attribute__((aligned(16))) float sT[4 * 64 * kW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int rb, kchunk;
  wg_to_tile_sym(blockIdx.x, N, kBR, chunk_cols, rb, kchunk);
  const int r0 = rb * kBR;
  const long long cb = (long long)r0 + (long long)kchunk * chunk_cols;
  if (cb >= N) return;
  const int c_begin = (int)cb;
  const int c_end = (c_begin + chunk_cols < N) ? c_begin + chunk_cols : N;

  // row side, packed over the lane's two rows: A[j] = {z_r0, z_r1}, LR = {L_t pairs (12), R_t pairs (12)}; rows past N and
  // slots t >= T carry L = R = 0 (S = 0: no contribution); loads from clamped addresses, masked by a multiply
  v32f A0, G0, LR0;
  v8f A1, G1;
  v16f LR1;
  v2f accS = {0.f, 0.f};
  {
    const int row0 = r0 + wave * 128 + lane, row1 = row0 + 64;
    const int rc0 = row0 < N ? row0 : N - 1, rc1 = row1 < N ? row1 : N - 1;
    const float m0 = row0 < N ? 1.f : 0.f, m1 = row1 < N ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < kJT; ++j) {
      const float x0 = Zs[(size_t)rc0 * kJT + j], x1 = Zs[(size_t)rc1 * kJT + j];
      if (j < 16) { A0[2 * j] = x0; A0[2 * j + 1] = x1; G0[2 * j] = 0.f; G0[2 * j + 1] = 0.f; }
      else { A1[2 * (j - 16)] = x0; A1[2 * (j - 16) + 1] = x1; G1[2 * (j - 16)] = 0.f; G1[2 * (j - 16) + 1] = 0.f; }
    }
#pragma unroll
    for (int t = 0; t < kTT; ++t) {
      const int tc = t < T ? t : T - 1;
      const float mt = t < T ? 1.f : 0.f;
      const float l0 = L[(size_t)rc0 * T + tc] * (m0 * mt), l1 = L[(size_t)rc1 * T + tc] * (m1 * mt);
      const float q0 = Rm[(size_t)rc0 * T + tc] * (m0 * mt), q1 = Rm[(size_t)rc1 * T + tc] * (m1 * mt);
      // flat index: L pairs at 2 t, R pairs at 24 + 2 t
      if (2 * t < 32) { LR0[2 * t] = l0; LR0[2 * t + 1] = l1; }
      const int ri = 24 + 2 * t;
      if (ri < 32) { LR0[ri] = q0; LR0[ri + 1] = q1; }
      else { LR1[ri - 32] = q0; LR1[ri - 32 + 1] = q1; }
    }
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)sC;

  for (int c0 = c_begin; c0 < c_end; c0 += 64) {
    __syncthreads();
    {
      // the subtile image by LDS-DMA: kNRec records (the 64 columns, then the first 66 again), element e -> (record, slot)
      constexpr int total = kNRec * kRec;                      // 5720 floats
      constexpr int NST = (total + 255) / 256;                 // 23 wave-wide pieces per wave
      const unsigned uw = __builtin_amdgcn_readfirstlane((unsigned)wave);
#pragma unroll 1
      for (int it = 0; it < NST; ++it) {
        const int e = tid + 256 * it;
        if (e >= total) break;                                  // (EXEC-masked lanes of a DMA do not write)
        const int rec = e / kRec, q = e - rec * kRec;
        const int col = c0 + (rec & 63);
        const int colc = col < N ? col : N - 1;                 // clamped: every lane of the DMA reads a valid address
        const bool isz = q < kJT, isl = q < kJT + kTT;
        int t = isz ? 0 : (isl ? q - kJT : q - kJT - kTT);
        t = t < T ? t : T - 1;                                  // (slot t >= T: the row side's L / R are zero there)
        const float *src = isz ? Zs : (isl ? L : Rm);
        const size_t off = isz ? (size_t)colc * kJT + q : (size_t)colc * T + t;
        glds_dword(src + off, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(it * 256 + (int)uw * 64) * 4u));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (c0 + 64 > c_end) {                                    // ragged last subtile: columns beyond the end get L = R = 0
#pragma unroll 1
        for (int it = 0; it < NST; ++it) {
          const int e = tid + 256 * it;
          const int rec = e / kRec, q = e - rec * kRec;
          if (e < total && c0 + (rec & 63) >= c_end && q >= kJT) sC[e] = 0.f;
        }
      }
    }
    __syncthreads();
    const bool doT = (c0 >= r0 + kBR);
    unsigned ptr = lds_base + (unsigned)(lane * kRec * 4);
    v16f TG0;
    v4f TG1;
    float TG2;
    int cnt;
    asm volatile(RPGP_BIL_ASM_LOOP
                 : "+{v[98:129]}"(G0), "+{v[130:137]}"(G1), "+{v[138:139]}"(accS), "+{v161}"(ptr), "={v[140:155]}"(TG0),
                   "={v[156:159]}"(TG1), "={v160}"(TG2), [cnt] "=s"(cnt)
                 : "{v[10:41]}"(A0), "{v[42:49]}"(A1), "{v[50:81]}"(LR0), "{v[82:97]}"(LR1)
                 : RPGP_BIL_ASM_CLOBBERS, "memory");
    if (doT) {                                                  // (subtiles inside the row block: both sides are swept anyway)
      float *dstT = sT + (wave * 64 + lane) * kW;
#pragma unroll
      for (int q = 0; q < 16; ++q) dstT[q] = TG0[q];
#pragma unroll
      for (int q = 0; q < 4; ++q) dstT[16 + q] = TG1[q];
      dstT[20] = TG2;
    }
    __syncthreads();
    if (doT) {
      for (int e = tid; e < 64 * kW; e += 256) {
        const int c = e / kW, q = e % kW;
        const int col = c0 + c;
        if (col < c_end) {
          const float sum = sT[(0 * 64 + c) * kW + q] + sT[(1 * 64 + c) * kW + q] + sT[(2 * 64 + c) * kW + q] +
                            sT[(3 * 64 + c) * kW + q];
          slabT[((size_t)rb * N + col) * kW + q] = sum;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = r0 + wave * 128 + r * 64 + lane;
    if (row < N) {
      float *dst = slabR + ((size_t)kchunk * N + row) * kW;
#pragma unroll
      for (int j = 0; j < kJT; ++j) dst[j] = j < 16 ? G0[2 * j + r] : G1[2 * (j - 16) + r];
      dst[kJT] = r == 0 ? accS.x : accS.y;
    }
  }
}

}  // namespace

namespace rpgp_internal {

int launch_bilinear_sym_asm(const float *Zs, const float *L, const float *R, float *slabR, float *slabT, int N, int T,
                            int chunk_cols, int nwg, hipStream_t st) {
  hipLaunchKernelGGL(bilinear_sym_asm_kernel, dim3((unsigned)nwg), dim3(256), 0, st, Zs, L, R, slabR, slabT, N, T, chunk_cols);
  return (int)hipGetLastError();
}

}  // namespace rpgp_internal
