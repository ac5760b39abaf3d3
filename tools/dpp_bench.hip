// dpp_bench.hip — issue cost of the DPP lane-movement forms considered for the transposed accumulators:
//   wave_rol:1 (0x134, crosses the 16-lane rows) vs row_ror:1 (0x121, inside a row) vs a plain v_mov / v_add.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096, CH = 8;
template <int CTRL>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
  float x[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) x[i] = seed + threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if constexpr (CTRL == 0) {
        x[i] = x[i] + seed;
      } else if constexpr (CTRL == -1) {          // fused: add with a DPP row_ror source
        int xi = __builtin_bit_cast(int, x[i]);
        xi = __builtin_amdgcn_update_dpp(0, xi, 0x121, 0xf, 0xf, true);
        x[i] = __builtin_bit_cast(float, xi) + x[(i + 1) % CH];
      } else {
        int xi = __builtin_bit_cast(int, x[i]);
        xi = __builtin_amdgcn_update_dpp(0, xi, CTRL, 0xf, 0xf, true);
        x[i] = __builtin_bit_cast(float, xi);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CTRL>
int run(const char *name, int wpc, int ncu, float *d) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<CTRL>, dim3(ncu * wpc), dim3(256), 0, 0, d, 0.5f);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<CTRL>, dim3(ncu * wpc), dim3(256), 0, 0, d, 0.5f);
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double winst = (double)ncu * wpc * 4 * ITERS * CH;
  printf("%-28s waves/SIMD=%d  %.4f ms  %.3e wave-instr/s  (%.2f ns per wave-instr per SIMD)\n", name, wpc, ms, winst / (ms * 1e-3),
         ms * 1e6 / (winst / (ncu * 4.0)));
  return 0;
}
int main() {
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  float *d; CHK(hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 256 * 4));
  for (int w : {2, 4}) {
    if (run<0>("v_add_f32", w, p.multiProcessorCount, d)) return 1;
    if (run<0x134>("v_mov_dpp wave_rol:1", w, p.multiProcessorCount, d)) return 1;
    if (run<0x121>("v_mov_dpp row_ror:1", w, p.multiProcessorCount, d)) return 1;
    if (run<0x111>("v_mov_dpp row_shr:1", w, p.multiProcessorCount, d)) return 1;
    if (run<-1>("v_add_dpp row_ror:1 (fused)", w, p.multiProcessorCount, d)) return 1;
  }
  return 0;
}
