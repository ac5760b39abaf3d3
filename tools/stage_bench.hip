// stage_bench.hip — what does ONE pipeline stage of the matrix-core MVM cost on gfx950?
//   stage = [v_mfma_f32_32x32x2_f32 of projection j + 1]  ||  [16 x (v_exp_f32, v_fmac_f32) on the results of projection j]
// Variants: with / without the MFMA, exp/fma interleave pattern, operands from registers or LDS, waves per SIMD.
// Standalone: hipcc -O3 --offload-arch=gfx950 tools/stage_bench.hip -o tools/stage_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NJ = 20;
constexpr int TILES = 256;   // tiles per wave (each NJ stages)

// interleave patterns of the 8-result half stage (t = exp temporaries, k = accumulators, m = multiplier)
//  0: e0 e1 f0 e2 f1 ... (skew 1)      1: 8 exps then 8 fmas      2: e0 e1 e2 f0 e3 f1 ... (skew 2)      3: e f e f (no skew: hazard nops needed -> s_nop)
#define HALF_SKEW1 \
  "v_exp_f32_e32 %8, %16\n\tv_exp_f32_e32 %9, %17\n\tv_fmac_f32_e32 %0, %8, %24\n\tv_exp_f32_e32 %10, %18\n\tv_fmac_f32_e32 %1, %9, %24\n\t" \
  "v_exp_f32_e32 %11, %19\n\tv_fmac_f32_e32 %2, %10, %24\n\tv_exp_f32_e32 %12, %20\n\tv_fmac_f32_e32 %3, %11, %24\n\t" \
  "v_exp_f32_e32 %13, %21\n\tv_fmac_f32_e32 %4, %12, %24\n\tv_exp_f32_e32 %14, %22\n\tv_fmac_f32_e32 %5, %13, %24\n\t" \
  "v_exp_f32_e32 %15, %23\n\tv_fmac_f32_e32 %6, %14, %24\n\ts_nop 0\n\tv_fmac_f32_e32 %7, %15, %24"
#define HALF_BLOCK8 \
  "v_exp_f32_e32 %8, %16\n\tv_exp_f32_e32 %9, %17\n\tv_exp_f32_e32 %10, %18\n\tv_exp_f32_e32 %11, %19\n\t" \
  "v_exp_f32_e32 %12, %20\n\tv_exp_f32_e32 %13, %21\n\tv_exp_f32_e32 %14, %22\n\tv_exp_f32_e32 %15, %23\n\t" \
  "v_fmac_f32_e32 %0, %8, %24\n\tv_fmac_f32_e32 %1, %9, %24\n\tv_fmac_f32_e32 %2, %10, %24\n\tv_fmac_f32_e32 %3, %11, %24\n\t" \
  "v_fmac_f32_e32 %4, %12, %24\n\tv_fmac_f32_e32 %5, %13, %24\n\tv_fmac_f32_e32 %6, %14, %24\n\tv_fmac_f32_e32 %7, %15, %24"
#define HALF_SKEW2 \
  "v_exp_f32_e32 %8, %16\n\tv_exp_f32_e32 %9, %17\n\tv_exp_f32_e32 %10, %18\n\tv_fmac_f32_e32 %0, %8, %24\n\t" \
  "v_exp_f32_e32 %11, %19\n\tv_fmac_f32_e32 %1, %9, %24\n\tv_exp_f32_e32 %12, %20\n\tv_fmac_f32_e32 %2, %10, %24\n\t" \
  "v_exp_f32_e32 %13, %21\n\tv_fmac_f32_e32 %3, %11, %24\n\tv_exp_f32_e32 %14, %22\n\tv_fmac_f32_e32 %4, %12, %24\n\t" \
  "v_exp_f32_e32 %15, %23\n\tv_fmac_f32_e32 %5, %13, %24\n\tv_fmac_f32_e32 %6, %14, %24\n\tv_fmac_f32_e32 %7, %15, %24"

template <int PAT>
__device__ __forceinline__ void half_stage(float (&k)[16], const f32x16 &d, int o, float m) {
  float t0, t1, t2, t3, t4, t5, t6, t7;
#define OPS                                                                                                          \
  : "+v"(k[o + 0]), "+v"(k[o + 1]), "+v"(k[o + 2]), "+v"(k[o + 3]), "+v"(k[o + 4]), "+v"(k[o + 5]), "+v"(k[o + 6]),  \
    "+v"(k[o + 7]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)            \
  : "v"(d[o + 0]), "v"(d[o + 1]), "v"(d[o + 2]), "v"(d[o + 3]), "v"(d[o + 4]), "v"(d[o + 5]), "v"(d[o + 6]),          \
    "v"(d[o + 7]), "v"(m)
  if constexpr (PAT == 0) asm volatile(HALF_SKEW1 OPS);
  else if constexpr (PAT == 1) asm volatile(HALF_BLOCK8 OPS);
  else asm volatile(HALF_SKEW2 OPS);
#undef OPS
}

// MFMA: 0 none, 1 f32 32x32x2
template <int MF, int PAT, bool LDSOPS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void stage_kernel(float *out, float seed) {
  __shared__ float sB[NJ * 128], sE[NJ * 128];
  const int lane = threadIdx.x & 63;
  for (int e = threadIdx.x; e < NJ * 128; e += 64 * WAVES) {
    sB[e] = 0.01f * (float)(e % 97) - 0.5f;
    sE[e] = 1.0f / (1.0f + (float)(e % 13));
  }
  __syncthreads();
  float A[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) A[j] = seed * (float)(lane + j) * 0.01f - 0.3f;
  float k[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) k[r] = 0.f;
  float total = 0.f;
  f32x16 c0, c1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { c0[r] = -0.1f * r; c1[r] = -0.2f * r; }
  const float *pb = sB + (lane & 31), *pe = sE + (lane & 31);
#define ISSUE(dst, jj, tl)                                                                        \
  {                                                                                               \
    float bq, eq;                                                                                 \
    if constexpr (LDSOPS) { bq = pb[((jj) % NJ) * 128 + ((tl) & 3) * 32]; } else { bq = A[((jj) + 3) % NJ]; } \
    if constexpr (MF == 1)                                                                        \
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(dst) : "v"(A[(jj) % NJ]), "v"(bq)); \
    else                                                                                          \
      asm volatile("v_mov_b32 %0, %1" : "+v"(dst[0]) : "v"(bq), "v"(A[(jj) % NJ]));                \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  }
#define CONSUME(cc, jj, tl)                                                                       \
  {                                                                                               \
    float m;                                                                                      \
    if constexpr (LDSOPS) m = pe[((jj) % NJ) * 128 + ((tl) & 3) * 32]; else m = A[((jj) + 7) % NJ]; \
    half_stage<PAT>(k, cc, 0, m);                                                                 \
    half_stage<PAT>(k, cc, 8, m);                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  }
  ISSUE(c0, 0, 0)
  for (int tile = 0; tile < TILES; ++tile) {
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
      ISSUE(c1, j + 1, tile)
      CONSUME(c0, j, tile)
      ISSUE(c0, j + 2, tile + (j + 2) / NJ)
      CONSUME(c1, j + 1, tile)
    }
    // tile epilogue: 32 product FMAs as in the real kernel, K reset
#pragma unroll
    for (int r = 0; r < 16; ++r) { total = __builtin_fmaf(k[r], seed, total); k[r] = 0.f; }
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = total + c0[0];
}

template <int MF, int PAT, bool LDSOPS, int WAVES>
int run(int wgs_per_cu, int ncu, float *dout) {
  const int blocks = ncu * wgs_per_cu;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((stage_kernel<MF, PAT, LDSOPS, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, dout, 0.5f);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  const int reps = 3;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((stage_kernel<MF, PAT, LDSOPS, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, dout, 0.5f);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double stages = (double)blocks * WAVES * TILES * NJ;
  const double terms = stages * 1024.0;
  const double wps = wgs_per_cu * WAVES / 4.0;
  // ns per stage per SIMD (all SIMDs busy): time / (stages per SIMD)
  const double ns_stage = ms * 1e6 / (stages / (ncu * 4.0));
  printf("mfma=%d pattern=%d lds=%d waves/SIMD=%.1f (WG %d waves): %8.3f ms  %.3e pair-terms/s  %.1f ns per stage per SIMD  -> C4 MVM %.3f ms\n",
         MF, PAT, (int)LDSOPS, wps, WAVES, ms, terms / (ms * 1e-3), ns_stage, 1.25e9 * 20 / (terms / (ms * 1e-3)) * 1e3);
  CHK(hipEventDestroy(e0));
  CHK(hipEventDestroy(e1));
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  float *dout;
  CHK(hipMalloc(&dout, (size_t)ncu * 8 * 512 * sizeof(float)));
  for (int w : {2, 3, 4}) {
    if (run<0, 0, false, 4>(w, ncu, dout)) return 1;
    if (run<0, 1, false, 4>(w, ncu, dout)) return 1;
    if (run<0, 2, false, 4>(w, ncu, dout)) return 1;
    if (run<1, 0, false, 4>(w, ncu, dout)) return 1;
    if (run<1, 1, false, 4>(w, ncu, dout)) return 1;
    if (run<1, 2, false, 4>(w, ncu, dout)) return 1;
    if (run<1, 0, true, 4>(w, ncu, dout)) return 1;
    if (run<1, 1, true, 4>(w, ncu, dout)) return 1;
    if (run<0, 1, true, 4>(w, ncu, dout)) return 1;
    printf("\n");
  }
  return 0;
}
