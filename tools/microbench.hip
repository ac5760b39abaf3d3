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
template <int NJ, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void mfma_mix_kernel(float *out, unsigned long long *cyc, float seed) {
  __shared__ intx4 sB[NJ * 64];
  const int lane = threadIdx.x & 63;
  for (int e = threadIdx.x; e < NJ * 64; e += 64 * WAVES) sB[e] = intx4{(int)(e * 2654435761u) & 0x3f803f80, e, e * 3, e * 7} & 0x3fff3fff;
  __syncthreads();
  bf16x8 A[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    intx4 t = {(lane + j) * 0x01010101 & 0x3f7f3f7f, j * 0x00110011 & 0x3f7f3f7f, lane & 0x3f7f, 0x3c003c00};
    A[j] = __builtin_bit_cast(bf16x8, t);
  }
  floatx16 kacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) kacc[r] = 0.f;
  const floatx16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // software pipeline with two statically named result tiles: the MFMA for projection j+1 is issued before the
  // exps of projection j are consumed (no register copies)
#define MM(jj, itv) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[(jj) % NJ], __builtin_bit_cast(bf16x8, sB[((jj) % NJ) * 64 + ((lane + (itv)) & 63)]), zero, 0, 0, 0)
  floatx16 c0 = MM(0, 0), c1;
  for (int it = 0; it < ITERS / 16; ++it) {
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
      c1 = MM(j + 1, it);
#pragma unroll
      for (int r = 0; r < 16; ++r) kacc[r] += __builtin_amdgcn_exp2f(-c0[r]);
      c0 = MM(j + 2, it + (j + 2) / NJ);
#pragma unroll
      for (int r = 0; r < 16; ++r) kacc[r] += __builtin_amdgcn_exp2f(-c1[r]);
    }
  }
#undef MM
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += kacc[r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[(size_t)blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NJ, int WAVES>
int run_mfma_mix(int wgs_per_cu, int ncu, float *dout, unsigned long long *dcyc) {
  const int blocks = ncu * wgs_per_cu;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((mfma_mix_kernel<NJ, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, dout, dcyc, 0.5f);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((mfma_mix_kernel<NJ, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, dout, dcyc, 0.5f);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double terms = (double)blocks * WAVES * (ITERS / 16) * NJ * 1024.0;
  printf("mfma32x32x16bf16 + 16 exp + 16 add  NJ=%d waves/WG=%d WG/CU=%d (waves/SIMD=%.1f)  time=%8.3f ms  pair-terms/s=%.3e\n",
         NJ, WAVES, wgs_per_cu, wgs_per_cu * WAVES / 4.0, ms, terms / (ms * 1e-3));
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
    if (run_mfma_mix<20, 4>(w, ncu, dout, dcyc)) return 1;
    if (run_mfma_mix<20, 8>(w, ncu, dout, dcyc)) return 1;
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
