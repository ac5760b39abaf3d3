// Lab for the cached-K product's streaming kernel: variants of dense_gemv_valu_kernel with per-wave time stamps
// (wall_clock64, 100 MHz) to see where a wave's life goes.  Not part of the product.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemv_lab.hip -o tools/gemv_lab ; run: tools/gemv_lab N T nsplit_mult
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstdint>
#include <dlfcn.h>
typedef float float4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: product form (bulk batches).  MODE 1: no FMAs beyond t = 0 (VALU removed, same loads / slabs).
// MODE 2: V taken from registers (no scalar loads in the loop).  MODE 3: no LDS reduce / slab store of t > 0.
template <int TT, int MODE>
__global__ __launch_bounds__(256) void gemv(const float *__restrict__ Kd, const float *__restrict__ V,
                                             float *__restrict__ slab, int N, long long ldk, int rows_per_split,
                                             unsigned long long *__restrict__ stamps) {
  __shared__ float4v sRed[2 * TT * 64];
  const unsigned long long t0 = wall_clock64();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cl = blockIdx.x * 256 + 4 * lane;
  float acc[4][TT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[i][t] = 0.f;
  const int rs = blockIdx.y * rows_per_split;
  const int re = (rs + rows_per_split < N) ? rs + rows_per_split : N;
  constexpr int RB = 8;
  if (cl + 3 < N) {
    int row = rs + wave * RB;
    const float *kp = Kd + (size_t)row * ldk + cl;
    const float *vp = V + (size_t)row * TT;
    for (; row + RB <= re; row += 4 * RB) {
      float4v a[RB];
#pragma unroll
      for (int q = 0; q < RB; ++q) a[q] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(kp + (size_t)q * ldk));
#pragma unroll
      for (int q = 0; q < RB; ++q) {
#pragma unroll
        for (int t = 0; t < ((MODE == 1) ? 1 : TT); ++t) {
          const float v = (MODE == 2) ? (float)(q + t) : vp[q * TT + t];
          acc[0][t] = __builtin_fmaf(a[q].x, v, acc[0][t]);
          acc[1][t] = __builtin_fmaf(a[q].y, v, acc[1][t]);
          acc[2][t] = __builtin_fmaf(a[q].z, v, acc[2][t]);
          acc[3][t] = __builtin_fmaf(a[q].w, v, acc[3][t]);
        }
      }
      kp += (size_t)4 * RB * ldk;
      vp += 4 * RB * TT;
    }
  }
  const unsigned long long t1 = wall_clock64();
  constexpr int TS = (MODE == 3) ? 1 : TT;
  {
    if (wave >= 2) {
#pragma unroll
      for (int t = 0; t < TS; ++t) sRed[((wave - 2) * TT + t) * 64 + lane] = float4v{acc[0][t], acc[1][t], acc[2][t], acc[3][t]};
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        const float4v x = sRed[(wave * TT + t) * 64 + lane];
        acc[0][t] += x.x; acc[1][t] += x.y; acc[2][t] += x.z; acc[3][t] += x.w;
      }
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
      for (int t = 0; t < TS; ++t) sRed[t * 64 + lane] = float4v{acc[0][t], acc[1][t], acc[2][t], acc[3][t]};
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        const float4v x = sRed[t * 64 + lane];
        acc[0][t] += x.x; acc[1][t] += x.y; acc[2][t] += x.z; acc[3][t] += x.w;
      }
      const size_t npad = ((size_t)N + 3) & ~(size_t)3;
      float *sl = slab + (size_t)blockIdx.y * npad * TT;
      if (cl + 3 < N) {
#pragma unroll
        for (int t = 0; t < TS; ++t)
          *reinterpret_cast<float4v *>(sl + (size_t)t * npad + cl) = float4v{acc[0][t], acc[1][t], acc[2][t], acc[3][t]};
      }
    }
  }
  if (stamps && lane == 0) {
    const unsigned long long t2 = wall_clock64();
    const size_t w = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 4 + wave;
    stamps[3 * w + 0] = t0;
    stamps[3 * w + 1] = t1;
    stamps[3 * w + 2] = t2;
  }
}

// Ring form: D row requests in flight per lane; row i's FMAs are followed at once by the request for row i + D.
template <int TT, int D>
__global__ __launch_bounds__(256) void gemv_ring(const float *__restrict__ Kd, const float *__restrict__ V,
                                                  float *__restrict__ slab, int N, long long ldk, int rows_per_split,
                                                  unsigned long long *__restrict__ stamps) {
  __shared__ float4v sRed[2 * TT * 64];
  const unsigned long long t0 = wall_clock64();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cl = blockIdx.x * 256 + 4 * lane;
  float acc[4][TT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[i][t] = 0.f;
  const int rs = blockIdx.y * rows_per_split;
  const int re = (rs + rows_per_split < N) ? rs + rows_per_split : N;
  constexpr int RB = 8;
  if (cl + 3 < N) {
    int row = rs + wave * RB;
    const float *kp = Kd + (size_t)row * ldk + cl;
    const float *vp = V + (size_t)row * TT;
    if (row + RB <= re) {
      float4v a[D];
#pragma unroll
      for (int q = 0; q < D; ++q) a[q] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(kp + (size_t)q * ldk));
#pragma nounroll
      while (row + 4 * RB + RB <= re) {
        const float *kn = kp + (size_t)4 * RB * ldk;
#pragma unroll
        for (int q = 0; q < RB; ++q) {
#pragma unroll
          for (int t = 0; t < TT; ++t) {
            const float v = vp[q * TT + t];
            acc[0][t] = __builtin_fmaf(a[q % D].x, v, acc[0][t]);
            acc[1][t] = __builtin_fmaf(a[q % D].y, v, acc[1][t]);
            acc[2][t] = __builtin_fmaf(a[q % D].z, v, acc[2][t]);
            acc[3][t] = __builtin_fmaf(a[q % D].w, v, acc[3][t]);
          }
          __builtin_amdgcn_sched_barrier(0);
          const float *src = (q + D < RB) ? kp + (size_t)(q + D) * ldk : kn + (size_t)(q + D - RB) * ldk;
          a[q % D] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(src));
          __builtin_amdgcn_sched_barrier(0);
        }
        kp = kn;
        vp += 4 * RB * TT;
        row += 4 * RB;
      }
      // last batch of this wave: no requests beyond it
#pragma unroll
      for (int q = 0; q < RB; ++q) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const float v = vp[q * TT + t];
          acc[0][t] = __builtin_fmaf(a[q % D].x, v, acc[0][t]);
          acc[1][t] = __builtin_fmaf(a[q % D].y, v, acc[1][t]);
          acc[2][t] = __builtin_fmaf(a[q % D].z, v, acc[2][t]);
          acc[3][t] = __builtin_fmaf(a[q % D].w, v, acc[3][t]);
        }
        if (q + D < RB) {
          __builtin_amdgcn_sched_barrier(0);
          a[q % D] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(kp + (size_t)(q + D) * ldk));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  {
    if (wave >= 2) {
#pragma unroll
      for (int t = 0; t < TT; ++t) sRed[((wave - 2) * TT + t) * 64 + lane] = float4v{acc[0][t], acc[1][t], acc[2][t], acc[3][t]};
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const float4v x = sRed[(wave * TT + t) * 64 + lane];
        acc[0][t] += x.x; acc[1][t] += x.y; acc[2][t] += x.z; acc[3][t] += x.w;
      }
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
      for (int t = 0; t < TT; ++t) sRed[t * 64 + lane] = float4v{acc[0][t], acc[1][t], acc[2][t], acc[3][t]};
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const float4v x = sRed[t * 64 + lane];
        acc[0][t] += x.x; acc[1][t] += x.y; acc[2][t] += x.z; acc[3][t] += x.w;
      }
      const size_t npad = ((size_t)N + 3) & ~(size_t)3;
      float *sl = slab + (size_t)blockIdx.y * npad * TT;
      if (cl + 3 < N) {
#pragma unroll
        for (int t = 0; t < TT; ++t)
          *reinterpret_cast<float4v *>(sl + (size_t)t * npad + cl) = float4v{acc[0][t], acc[1][t], acc[2][t], acc[3][t]};
      }
    }
  }
  if (stamps && lane == 0) {
    const unsigned long long t2 = wall_clock64();
    const size_t w = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 4 + wave;
    stamps[3 * w + 0] = t0;
    stamps[3 * w + 1] = t1;
    stamps[3 * w + 2] = t2;
  }
}

__global__ void fill(float *p, size_t n, float v, int rnd) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (rnd) {
      unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
      x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
      p[i] = v * ((float)(x & 0xffffff) / 16777216.0f - (rnd == 2 ? 0.5f : 0.0f));
    } else {
      p[i] = v * (float)(i % 7);
    }
  }
}

typedef void (*kern_t)(const float *, const float *, float *, int, long long, int, unsigned long long *);
template <int TT>
void run_k(kern_t kf, const char *label, const float *K, const float *V, float *slab, int N, long long ldk, int nsplit, bool stamp) {
  const unsigned ncb = (N + 255) / 256;
  int cps = (N + nsplit - 1) / nsplit;
  cps = (cps + 31) / 32 * 32;
  nsplit = (N + cps - 1) / cps;
  dim3 grid(ncb, nsplit), block(256);
  const size_t nw = (size_t)ncb * nsplit * 4;
  unsigned long long *st = nullptr;
  CK(hipMalloc((void **)&st, nw * 3 * sizeof(unsigned long long)));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kf, grid, block, 0, 0, K, V, slab, N, ldk, cps, nullptr);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kf, grid, block, 0, 0, K, V, slab, N, ldk, cps, nullptr);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s T=%2d grid %u x %d (%u WGs, %d rows/split): %.1f us/launch, %.2f TB/s", label, TT, ncb, nsplit, ncb * nsplit,
         cps, ms / reps * 1e3, 4.0 * N * (double)N / (ms / reps * 1e-3) / 1e12);
  if (stamp) {
    hipLaunchKernelGGL(kf, grid, block, 0, 0, K, V, slab, N, ldk, cps, st);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nw * 3);
    CK(hipMemcpy(h.data(), st, nw * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t w = 0; w < nw; ++w) { tmin = std::min(tmin, h[3 * w]); tmax = std::max(tmax, h[3 * w + 2]); }
    std::vector<double> s0(nw), s1(nw), s2(nw);
    for (size_t w = 0; w < nw; ++w) { s0[w] = (h[3 * w] - tmin) * 0.01; s1[w] = (h[3 * w + 1] - tmin) * 0.01; s2[w] = (h[3 * w + 2] - tmin) * 0.01; }
    auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    printf("\n    span %.1f us | start p0/p50/p90/p100 = %.1f/%.1f/%.1f/%.1f | loop end p0/p10/p50/p90/p100 = %.1f/%.1f/%.1f/%.1f/%.1f | wave end p50/p100 = %.1f/%.1f",
           (tmax - tmin) * 0.01, pct(s0, 0), pct(s0, .5), pct(s0, .9), pct(s0, 1), pct(s1, 0), pct(s1, .1), pct(s1, .5), pct(s1, .9), pct(s1, 1),
           pct(s2, .5), pct(s2, 1));
  }
  printf("\n");
  CK(hipFree(st));
}

template <int TT, int MODE>
void run(const char *label, const float *K, const float *V, float *slab, int N, long long ldk, int nsplit, bool stamp) {
  run_k<TT>(gemv<TT, MODE>, label, K, V, slab, N, ldk, nsplit, stamp);
}
template <int TT, int D>
void run_ring(const char *label, const float *K, const float *V, float *slab, int N, long long ldk, int nsplit) {
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemv_ring<TT, D>, 256, 0));
  char buf[96];
  snprintf(buf, sizeof buf, "%s D=%d occ=%d", label, D, nb);
  run_k<TT>(gemv_ring<TT, D>, buf, K, V, slab, N, ldk, nsplit, false);
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 14939;
  const int mult = argc > 2 ? atoi(argv[2]) : 1;
  const long long ldk = (N + 63) / 64 * 64;
  float *K, *V, *slab;
  CK(hipMalloc((void **)&K, (size_t)N * ldk * 4));
  CK(hipMalloc((void **)&V, (size_t)N * 16 * 4));
  CK(hipMalloc((void **)&slab, (size_t)1024 * N * 4 * 12));
  const int rnd = argc > 3 ? atoi(argv[3]) : 0;
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, K, (size_t)N * ldk, rnd ? 1.0f : 1e-3f, rnd ? 1 : 0);
  hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, V, (size_t)N * 16, rnd ? 2.0f : 1e-2f, rnd ? 2 : 0);
  printf("N = %d, data: %s\n", N, rnd ? "random K in [0,1), V in [-1,1)" : "7 repeating small values");
  CK(hipDeviceSynchronize());
  const unsigned ncb = (N + 255) / 256;
  auto ns = [&](int percu) { int n = 256 * percu / ncb * mult; return n < 1 ? 1 : n; };
  run<1, 0>("T=1 product form", K, V, slab, N, ldk, ns(8), true);
  run<4, 0>("T=4 product form", K, V, slab, N, ldk, ns(6), true);
  run<11, 0>("T=11 product form", K, V, slab, N, ldk, ns(5), true);
  run<11, 1>("T=11 FMAs only for t=0", K, V, slab, N, ldk, ns(5), true);
  run<11, 2>("T=11 V from registers", K, V, slab, N, ldk, ns(5), true);
  run<11, 3>("T=11 reduce/store t=0 only", K, V, slab, N, ldk, ns(5), true);
  run<11, 0>("T=11 product, 8 WG/CU grid", K, V, slab, N, ldk, ns(8), true);
  run<1, 0>("T=1 product, 5 WG/CU grid", K, V, slab, N, ldk, ns(5), true);
  for (int pc : {5}) {
    printf("-- ring forms, grid sized for %d WG/CU\n", pc);
    run_ring<1, 2>("T=1 ring", K, V, slab, N, ldk, ns(pc));
    run_ring<1, 4>("T=1 ring", K, V, slab, N, ldk, ns(pc));
    run_ring<1, 8>("T=1 ring", K, V, slab, N, ldk, ns(pc));
    run_ring<4, 2>("T=4 ring", K, V, slab, N, ldk, ns(pc));
    run_ring<4, 4>("T=4 ring", K, V, slab, N, ldk, ns(pc));
    run_ring<11, 2>("T=11 ring", K, V, slab, N, ldk, ns(pc));
    run_ring<11, 3>("T=11 ring", K, V, slab, N, ldk, ns(pc));
    run_ring<11, 4>("T=11 ring", K, V, slab, N, ldk, ns(pc));
    run_ring<11, 8>("T=11 ring", K, V, slab, N, ldk, ns(pc));
  }
  // the product entry point on the same buffers
  if (void *h = dlopen(argc > 4 ? argv[4] : "randomly-projected-additive-gps_amd/csrc/librpgp.so", RTLD_NOW)) {
    typedef int (*mvm_t)(const float *, const float *, float *, int64_t, int64_t, int, float, void *);
    mvm_t mvm = (mvm_t)dlsym(h, "rpgp_dense_mvm");
    float *out;
    CK(hipMalloc((void **)&out, (size_t)N * 16 * 4));
    for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      hipMemPool_t pool;
      CK(hipDeviceGetDefaultMemPool(&pool, 0));
      uint64_t thr = UINT64_MAX;
      CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
      printf("default mempool release threshold -> max\n");
    }
    for (int T : {1, 4, 11}) {
      for (int i = 0; i < 3; ++i) mvm(K, V, out, N, ldk, T, 0.1f, nullptr);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i) mvm(K, V, out, N, ldk, T, 0.1f, nullptr);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("rpgp_dense_mvm T=%d: %.1f us/call (kernel + reduce)\n", T, ms / 20 * 1e3);
    }
    }
  } else {
    printf("librpgp.so not loaded: %s\n", dlerror());
  }
  return 0;
}
