// lds_atomic_bench.hip — cost of LDS atomics on gfx950: float add vs int add vs plain read-modify-write,
// conflict-free consecutive addresses vs the SKI scatter's pattern (16 random cells x 12 consecutive floats per
// workgroup-instruction, i.e. 4 cells per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE, int PATTERN>
__global__ __launch_bounds__(256) void bench(const int *cells, float *out, int iters) {
  extern __shared__ float sh[];
  int *shi = reinterpret_cast<int *>(sh);
  for (int e = threadIdx.x; e < 12288; e += 256) sh[e] = 0.f;
  __syncthreads();
  const int t = threadIdx.x % 16, p = threadIdx.x / 16;
  for (int it = 0; it < iters; ++it) {
    int addr;
    if (PATTERN == 0) addr = (threadIdx.x + 64 * (it & 7)) % 12288;            // consecutive, conflict-free
    else addr = cells[(blockIdx.x * iters + it) * 16 + p] * 12 + (t < 12 ? t : 0); // 16 random cells x 12 floats
    if (PATTERN == 1 && t >= 12) continue;
    const float v = 1.0f + it;
    if (MODE == 0) atomicAdd(&sh[addr], v);
    else if (MODE == 1) atomicAdd(&shi[addr], (int)v);
    else sh[addr] += v;
  }
  __syncthreads();
  float acc = 0.f;
  for (int e = threadIdx.x; e < 12288; e += 256) acc += sh[e];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int PATTERN>
int run(const char *name, const int *dcells, float *dout, int iters) {
  const int blocks = 768;
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL((bench<MODE, PATTERN>), dim3(blocks), dim3(256), 49152, 0, dcells, dout, iters);
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL((bench<MODE, PATTERN>), dim3(blocks), dim3(256), 49152, 0, dcells, dout, iters);
  CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms, a, b));
  const double wave_instr = (double)blocks * 4 * iters;
  printf("%-44s %8.3f ms  %8.1f ns per wave-instr per CU (3 WG/CU)\n", name, ms, ms * 1e6 / (wave_instr / 256.0));
  return 0;
}

int main() {
  const int iters = 2000, blocks = 768;
  std::vector<int> cells((size_t)blocks * iters * 16);
  unsigned s = 12345;
  for (auto &c : cells) { s = s * 1664525u + 1013904223u; c = (s >> 8) % 1021; }
  int *dcells; float *dout;
  CHK(hipMalloc(&dcells, cells.size() * sizeof(int)));
  CHK(hipMemcpy(dcells, cells.data(), cells.size() * sizeof(int), hipMemcpyHostToDevice));
  CHK(hipMalloc(&dout, (size_t)blocks * 256 * sizeof(float)));
  run<0, 0>("float atomic add, consecutive", dcells, dout, iters);
  run<1, 0>("int atomic add, consecutive", dcells, dout, iters);
  run<2, 0>("plain read-add-write, consecutive", dcells, dout, iters);
  run<0, 1>("float atomic add, 16 cells x 12 floats", dcells, dout, iters);
  run<1, 1>("int atomic add, 16 cells x 12 floats", dcells, dout, iters);
  run<2, 1>("plain read-add-write (racy), 16 cells x 12", dcells, dout, iters);
  return 0;
}
