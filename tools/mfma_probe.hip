// mfma_probe.hip — can the matrix pipe compute m = -(a-b)^2 accurately enough to feed v_exp_f32?
// Compares, against fp64, (i) VALU direct fp32, (ii) v_mfma_f32_32x32x16_bf16 on a 3-way bf16 split
// (products exact, K=16 slots), (iii) v_mfma_f32_16x16x4_f32 on [-a^2, a, 1, ra] x [1, 2b, -b^2, ...].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float bf16_trunc(float x) {
  unsigned u = __builtin_bit_cast(unsigned, x) & 0xffff0000u;
  return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ __bf16 to_bf16_exact(float x) {  // x must already be representable (or we accept RNE)
  return (__bf16)x;
}
__device__ void split3(float x, float &p1, float &p2, float &p3) {
  p1 = bf16_trunc(x);
  float r = x - p1;
  p2 = bf16_trunc(r);
  p3 = r - p2;  // <= 8 significant bits left
}

// order: which slot permutation to use (0: grouped, 1: interleaved cancellation-friendly)
__global__ void probe_bf16(const float *a, const float *b, float *D, int order) {
  const int lane = threadIdx.x;
  const int m = lane & 31, half = lane >> 5;
  float av = a[m], bv = b[m];  // this lane's row value (A operand) and column value (B operand)
  float a1, a2, a3, b1, b2, b3;
  split3(av, a1, a2, a3);
  split3(bv, b1, b2, b3);
  float s = av * av, sl = __builtin_fmaf(av, av, -s);
  float t = bv * bv, tl = __builtin_fmaf(bv, bv, -t);
  float s1, s2, s3, t1, t2, t3;
  split3(s, s1, s2, s3);
  split3(t, t1, t2, t3);
  float A[16], B[16];
  if (order == 0) {
    float Aa[16] = {a1, a1, a2, a2, a1, a3, a2, a3, s1, s2, s3, sl, 1.f, 1.f, 1.f, 1.f};
    float Bb[16] = {2 * b1, 2 * b2, 2 * b1, 2 * b2, 2 * b3, 2 * b1, 2 * b3, 2 * b2, -1.f, -1.f, -1.f, -1.f, -t1, -t2, -t3, -tl};
    for (int k = 0; k < 16; ++k) { A[k] = Aa[k]; B[k] = Bb[k]; }
  } else {
    // interleave so that leading terms cancel early: 2a1b1 - s1 - t1, then the next order, ...
    float Aa[16] = {a1, s1, 1.f, a1, a2, s2, 1.f, a2, a1, a3, s3, 1.f, a2, a3, sl, 1.f};
    float Bb[16] = {2 * b1, -1.f, -t1, 2 * b2, 2 * b1, -1.f, -t2, 2 * b2, 2 * b3, 2 * b1, -1.f, -t3, 2 * b3, 2 * b2, -1.f, -tl};
    for (int k = 0; k < 16; ++k) { A[k] = Aa[k]; B[k] = Bb[k]; }
  }
  bf16x8 av8, bv8;
  for (int k = 0; k < 8; ++k) {
    av8[k] = to_bf16_exact(A[8 * half + k]);
    bv8[k] = to_bf16_exact(B[8 * half + k]);
  }
  floatx16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av8, bv8, c, 0, 0, 0);
  // D layout: n = lane%32, m = (r/4)*8 + half*4 + r%4
  for (int r = 0; r < 16; ++r) {
    int mm = (r / 4) * 8 + half * 4 + (r % 4);
    D[mm * 32 + m] = c[r];
  }
}

__global__ void probe_f32(const float *a, const float *b, float *D) {
  // 16x16x4: A lane (m=l%16,k=l/16), B lane (k=l/16,n=l%16), D lane: n=l%16, m=4*(l/16)+r
  const int lane = threadIdx.x;
  const int i = lane & 15, k = lane >> 4;
  for (int tm = 0; tm < 2; ++tm)
    for (int tn = 0; tn < 2; ++tn) {
      float av = a[tm * 16 + i], bv = b[tn * 16 + i];
      float s = av * av, sl = __builtin_fmaf(av, av, -s);
      float t = bv * bv;
      float Aa[4] = {-s, av, 1.f, -sl};
      float Bb[4] = {1.f, 2 * bv, -t, 1.f};
      floatx4 c = {0, 0, 0, 0};
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(Aa[k], Bb[k], c, 0, 0, 0);
      for (int r = 0; r < 4; ++r) D[(tm * 16 + 4 * k + r) * 32 + tn * 16 + i] = c[r];
    }
}

__global__ void probe_valu(const float *a, const float *b, float *D) {
  const int lane = threadIdx.x;
  for (int e = lane; e < 1024; e += 64) {
    float d = a[e / 32] - b[e % 32];
    D[e] = -(d * d);
  }
}

int main() {
  std::mt19937 rng(1);
  for (float scale : {1.0f, 3.0f, 10.0f, 30.0f}) {
    std::normal_distribution<float> nd(0.f, scale);
    double worst[4] = {0, 0, 0, 0}, worst_close[4] = {0, 0, 0, 0};
    for (int trial = 0; trial < 50; ++trial) {
      std::vector<float> a(32), b(32);
      for (int i = 0; i < 32; ++i) { a[i] = nd(rng); b[i] = (i % 2) ? a[i] + 0.5f * std::normal_distribution<float>(0.f, 1.f)(rng) : nd(rng); }
      float *da, *db, *dD;
      CHK(hipMalloc(&da, 128)); CHK(hipMalloc(&db, 128)); CHK(hipMalloc(&dD, 4096));
      CHK(hipMemcpy(da, a.data(), 128, hipMemcpyHostToDevice));
      CHK(hipMemcpy(db, b.data(), 128, hipMemcpyHostToDevice));
      std::vector<float> D(1024);
      for (int variant = 0; variant < 4; ++variant) {
        if (variant == 0) hipLaunchKernelGGL(probe_valu, dim3(1), dim3(64), 0, 0, da, db, dD);
        if (variant == 1) hipLaunchKernelGGL(probe_bf16, dim3(1), dim3(64), 0, 0, da, db, dD, 0);
        if (variant == 2) hipLaunchKernelGGL(probe_bf16, dim3(1), dim3(64), 0, 0, da, db, dD, 1);
        if (variant == 3) hipLaunchKernelGGL(probe_f32, dim3(1), dim3(64), 0, 0, da, db, dD);
        CHK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        for (int m = 0; m < 32; ++m)
          for (int n = 0; n < 32; ++n) {
            double d = (double)a[m] - (double)b[n];
            double ref = -d * d;
            double err = std::fabs((double)D[m * 32 + n] - ref);
            if (err > worst[variant]) worst[variant] = err;
            if (std::fabs(d) < 5.0 && err > worst_close[variant]) worst_close[variant] = err;
          }
      }
      hipFree(da); hipFree(db); hipFree(dD);
    }
    printf("scale=%5.1f  max|err(m)|  [all pairs / pairs with |a-b|<5]:  valu %.3e/%.3e   bf16x3(grouped) %.3e/%.3e   bf16x3(interleaved) %.3e/%.3e   f32 16x16x4 %.3e/%.3e\n",
           scale, worst[0], worst_close[0], worst[1], worst_close[1], worst[2], worst_close[2], worst[3], worst_close[3]);
  }
  return 0;
}
