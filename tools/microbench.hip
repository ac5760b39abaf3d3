// microbench.hip — VALU / transcendental issue-rate probes for gfx950, used to price the fused MVM
// kernel ("measure with a microbenchmark first").  Standalone: hipcc -O3 --offload-arch=gfx950.
// Prints, per instruction mix and waves/SIMD, the wave-instruction issue interval per SIMD in shader cycles
// (from s_memtime) and the chip-wide rate from wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float float2v __attribute__((ext_vector_type(2)));

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int CH = 8;  // independent chains per lane

enum Mix { EXP = 0, FMA, PKFMA, PKMUL, PKADD, ADD, DPPMOV, PAIR_SCALAR, PAIR_PACKED, PAIR_FACT, NMIX };
const char *mix_name[] = {"v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_add_f32",
                          "v_mov_dpp(wave_rol)", "pair: sub,mul,exp,add (scalar)", "pair: pk_add,pk_mul,2exp,pk_add",
                          "pair: fma,exp,fma (factorised)"};
// VALU wave-instructions per inner iteration, and "pair terms" (exp evaluations) per lane per iteration
const int mix_instr[] = {CH, CH, CH / 2, CH / 2, CH / 2, CH, CH, 4 * CH, 2 * CH, 3 * CH};
const int mix_terms[] = {CH, 0, 0, 0, 0, 0, 0, CH, CH, CH};

template <int MIX>
__global__ __launch_bounds__(256) void bench_kernel(float *out, unsigned long long *cyc, float seed) {
  float x[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) x[i] = seed + 0.001f * (threadIdx.x + i);
  float b = seed * 0.5f, c = seed * 0.25f;
  float acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) acc[i] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (MIX == EXP) {
#pragma unroll
      for (int i = 0; i < CH; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
    } else if constexpr (MIX == FMA) {
#pragma unroll
      for (int i = 0; i < CH; ++i) x[i] = __builtin_fmaf(x[i], b, c);
    } else if constexpr (MIX == PKFMA) {
#pragma unroll
      for (int i = 0; i < CH; i += 2) {
        float2v v = {x[i], x[i + 1]}, bb = {b, b}, cc = {c, c};
        v = __builtin_elementwise_fma(v, bb, cc);
        x[i] = v.x; x[i + 1] = v.y;
      }
    } else if constexpr (MIX == PKMUL) {
#pragma unroll
      for (int i = 0; i < CH; i += 2) {
        float2v v = {x[i], x[i + 1]}, bb = {b, c};
        v = v * bb;
        x[i] = v.x; x[i + 1] = v.y;
      }
    } else if constexpr (MIX == PKADD) {
#pragma unroll
      for (int i = 0; i < CH; i += 2) {
        float2v v = {x[i], x[i + 1]}, bb = {b, c};
        v = v + bb;
        x[i] = v.x; x[i + 1] = v.y;
      }
    } else if constexpr (MIX == ADD) {
#pragma unroll
      for (int i = 0; i < CH; ++i) x[i] = x[i] + b;
    } else if constexpr (MIX == DPPMOV) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        int xi = __builtin_bit_cast(int, x[i]);
        xi = __builtin_amdgcn_update_dpp(0, xi, 0x134, 0xf, 0xf, false);
        x[i] = __builtin_bit_cast(float, xi);
      }
    } else if constexpr (MIX == PAIR_SCALAR) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        float d = x[i] - b;
        float m = -(d * d);
        acc[i] += __builtin_amdgcn_exp2f(m);
      }
      b += 1e-6f;  // keep the loop body live and non-invariant (scalar-ish, 1 extra VALU per iteration)
    } else if constexpr (MIX == PAIR_PACKED) {
#pragma unroll
      for (int i = 0; i < CH; i += 2) {
        float2v a = {x[i], x[i + 1]}, bb = {b, c};
        float2v d = a - bb;
        float2v m = -(d * d);
        float2v e = {__builtin_amdgcn_exp2f(m.x), __builtin_amdgcn_exp2f(m.y)};
        float2v ac = {acc[i], acc[i + 1]};
        ac += e;
        acc[i] = ac.x; acc[i + 1] = ac.y;
      }
      b += 1e-6f; c += 1e-6f;
    } else if constexpr (MIX == PAIR_FACT) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        float m = __builtin_fmaf(x[i], b, c);
        acc[i] = __builtin_fmaf(__builtin_amdgcn_exp2f(m), c, acc[i]);
      }
      b += 1e-6f;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += x[i] + acc[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx4 __attribute__((ext_vector_type(4)));

// The candidate restructuring of the fused MVM: the matrix pipe produces m = -(a-b)^2 for a 32x32 tile per
// projection (one v_mfma_f32_32x32x16_bf16 on 3-way bf16 splits), VALU only does v_exp_f32 + accumulate.
// B operands come from LDS (one ds_read_b128 per projection) like the real kernel's column stream.
template <int NJ, int WAVES, bool USE_MFMA>
__global__ __launch_bounds__(64 * WAVES, 2) void mfma_mix_kernel(float *out, unsigned long long *cyc, float seed) {
  __shared__ intx4 sB[NJ * 64];
  const int lane = threadIdx.x & 63;
  for (int e = threadIdx.x; e < NJ * 64; e += 64 * WAVES) sB[e] = intx4{(int)(e * 2654435761u) & 0x3f803f80, e, e * 3, e * 7} & 0x3fff3fff;
  __syncthreads();
  intx4 A[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    intx4 t = {(lane + j) * 0x01010101 & 0x3f7f3f7f, j * 0x00110011 & 0x3f7f3f7f, lane & 0x3f7f, 0x3c003c00};
    A[j] = t;
  }
  float kacc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) kacc[r] = 0.f;
  const floatx16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // software pipeline with two statically named result tiles.  The MFMA is an opaque asm statement and
  // sched_barrier(0) fences keep "MFMA(j+1) | 16 exps + adds of j" in program order, so the matrix pipe works on
  // projection j+1 while VALU consumes projection j (hipcc otherwise sinks the MFMA next to its consumer).
#define MMASM(dst, jj, itv)                                                                         \
  {                                                                                                 \
    intx4 bq = sB[((jj) % NJ) * 64 + ((lane + (itv)) & 63)];                                        \
    if constexpr (USE_MFMA)                                                                           \
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(dst) : "v"(A[(jj) % NJ]), "v"(bq)); \
    else                                                                                              \
      asm volatile("v_mov_b32 %0, %1" : "+v"(dst[0]) : "v"(bq[0]), "v"(A[(jj) % NJ]));                 \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  }
#define EXPADD8(cc, o)                                                                                         \
  {                                                                                                            \
    float t0, t1, t2, t3, t4, t5, t6, t7;                                                                      \
    asm volatile(                                                                                              \
        "v_exp_f32_e32 %8, %16\n\tv_exp_f32_e32 %9, %17\n\tv_exp_f32_e32 %10, %18\n\tv_exp_f32_e32 %11, %19\n\t" \
        "v_exp_f32_e32 %12, %20\n\tv_exp_f32_e32 %13, %21\n\tv_exp_f32_e32 %14, %22\n\tv_exp_f32_e32 %15, %23\n\t" \
        "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"       \
        "v_add_f32 %4, %4, %12\n\tv_add_f32 %5, %5, %13\n\tv_add_f32 %6, %6, %14\n\tv_add_f32 %7, %7, %15"        \
        : "+v"(kacc[o + 0]), "+v"(kacc[o + 1]), "+v"(kacc[o + 2]), "+v"(kacc[o + 3]), "+v"(kacc[o + 4]),         \
          "+v"(kacc[o + 5]), "+v"(kacc[o + 6]), "+v"(kacc[o + 7]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),   \
          "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)                                                           \
        : "v"(cc[o + 0]), "v"(cc[o + 1]), "v"(cc[o + 2]), "v"(cc[o + 3]), "v"(cc[o + 4]), "v"(cc[o + 5]),       \
          "v"(cc[o + 6]), "v"(cc[o + 7]));                                                                     \
  }
#define CONSUME(cc)  \
  EXPADD8(cc, 0)     \
  EXPADD8(cc, 8)     \
  __builtin_amdgcn_sched_barrier(0);
  floatx16 c0 = zero, c1 = zero;
  MMASM(c0, 0, 0)
  for (int it = 0; it < ITERS / 16; ++it) {
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
      MMASM(c1, j + 1, it)
      CONSUME(c0)
      MMASM(c0, j + 2, it + (j + 2) / NJ)
      CONSUME(c1)
    }
  }
#undef MMASM
#undef CONSUME
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += kacc[r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[(size_t)blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NJ, int WAVES, bool USE_MFMA>
int run_mfma_mix(int wgs_per_cu, int ncu, float *dout, unsigned long long *dcyc) {
  const int blocks = ncu * wgs_per_cu;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((mfma_mix_kernel<NJ, WAVES, USE_MFMA>), dim3(blocks), dim3(64 * WAVES), 0, 0, dout, dcyc, 0.5f);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((mfma_mix_kernel<NJ, WAVES, USE_MFMA>), dim3(blocks), dim3(64 * WAVES), 0, 0, dout, dcyc, 0.5f);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double terms = (double)blocks * WAVES * (ITERS / 16) * NJ * 1024.0;
  std::vector<unsigned long long> h((size_t)blocks * WAVES);
  CHK(hipMemcpy(h.data(), dcyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  const double wps = wgs_per_cu * WAVES / 4.0;
  // SIMD cycles per (MFMA + 16 exp + 16 add) step, assuming the resident waves of a SIMD run concurrently
  const double cyc_per_step = med / ((ITERS / 16) * (double)NJ * wps);
  printf("%s + 16 exp + 16 add  NJ=%d waves/SIMD=%.1f  time=%8.3f ms  pair-terms/s=%.3e  SIMD-cycles/step=%.1f (=%.2f per 64 pair-terms)  clk~%.2f GHz\n",
         USE_MFMA ? "mfma32x32x16bf16" : "(no mfma)       ", NJ, wps, ms, terms / (ms * 1e-3), cyc_per_step, cyc_per_step / 16.0,
         med / (ms * 1e-3) * 1e-9);
  CHK(hipEventDestroy(e0));
  CHK(hipEventDestroy(e1));
  return 0;
}

template <int MIX>
int run_mix(int wgs_per_cu, int ncu, float *dout, unsigned long long *dcyc) {
  const int blocks = ncu * wgs_per_cu;  // 256-thread blocks: 1 wave per SIMD each
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(bench_kernel<MIX>, dim3(blocks), dim3(256), 0, 0, dout, dcyc, 0.5f);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(bench_kernel<MIX>, dim3(blocks), dim3(256), 0, 0, dout, dcyc, 0.5f);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::vector<unsigned long long> h((size_t)blocks * 4);
  CHK(hipMemcpy(h.data(), dcyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double med_cyc = (double)h[h.size() / 2];
  const double instr_per_wave = (double)ITERS * mix_instr[MIX];
  // issue interval per SIMD = wave cycles / (instructions per wave * waves per SIMD)
  const double interval = med_cyc / (instr_per_wave * wgs_per_cu);
  const double winst_per_s = (double)blocks * 4 * instr_per_wave / (ms * 1e-3);
  const double terms_per_s = (double)blocks * 256 * (double)ITERS * mix_terms[MIX] / (ms * 1e-3);
  printf("%-36s waves/SIMD=%d  cyc/wave-instr/SIMD=%6.2f  time=%8.3f ms  wave-instr/s=%.3e  pair-terms/s=%.3e  clk~%.2f GHz\n",
         mix_name[MIX], wgs_per_cu, interval, ms, winst_per_s, terms_per_s, med_cyc / (ms * 1e-3) * 1e-9);
  CHK(hipEventDestroy(e0));
  CHK(hipEventDestroy(e1));
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device: %s  arch=%s  CUs=%d  clock=%d kHz  memclock=%d kHz  buswidth=%d\n", prop.name, prop.gcnArchName, ncu,
         prop.clockRate, prop.memoryClockRate, prop.memoryBusWidth);
  float *dout;
  unsigned long long *dcyc;
  CHK(hipMalloc(&dout, (size_t)ncu * 8 * 512 * sizeof(float)));
  CHK(hipMalloc(&dcyc, (size_t)ncu * 8 * 8 * sizeof(unsigned long long)));
  for (int w : {1, 2, 3}) {
    if (run_mfma_mix<20, 4, true>(w, ncu, dout, dcyc)) return 1;
    if (run_mfma_mix<20, 4, false>(w, ncu, dout, dcyc)) return 1;
    if (run_mfma_mix<20, 8, true>(w, ncu, dout, dcyc)) return 1;
    if (run_mfma_mix<20, 8, false>(w, ncu, dout, dcyc)) return 1;
  }
  printf("\n");
  for (int w : {1, 2, 4, 8}) {
    if (run_mix<EXP>(w, ncu, dout, dcyc)) return 1;
    if (run_mix<FMA>(w, ncu, dout, dcyc)) return 1;
    if (run_mix<ADD>(w, ncu, dout, dcyc)) return 1;
    if (run_mix<PKFMA>(w, ncu, dout, dcyc)) return 1;
    if (run_mix<PKMUL>(w, ncu, dout, dcyc)) return 1;
    if (run_mix<PKADD>(w, ncu, dout, dcyc)) return 1;
    if (run_mix<DPPMOV>(w, ncu, dout, dcyc)) return 1;
    if (run_mix<PAIR_SCALAR>(w, ncu, dout, dcyc)) return 1;
    if (run_mix<PAIR_PACKED>(w, ncu, dout, dcyc)) return 1;
    if (run_mix<PAIR_FACT>(w, ncu, dout, dcyc)) return 1;
    printf("\n");
  }
  return 0;
}
