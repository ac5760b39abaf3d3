// mfma_f64_probe.hip — operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950 (lane l, result register r -> (m, n)).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double *out) {
  const int l = threadIdx.x;
  const int m = l % 16, k = l / 16;
  const double a = (k == 0) ? (double)(m + 1) : 0.0;            // A[m][k]
  const double b = (k == 0) ? (double)((l % 16) + 1) * 100.0 : 0.0;   // B[k][n], n = l % 16
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  double *d; hipMalloc(&d, 256 * sizeof(double));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 5) {
    printf("lane %2d:", l);
    for (int r = 0; r < 4; ++r) { int v = (int)(h[l * 4 + r] / 100.0 + 0.5); printf("  r%d -> (m=%d? product %d)", r, 0, v); }
    printf("\n");
  }
  // decode: value/100 = (m+1)*(n+1)
  int ok_f32_layout = 1, ok_alt = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int v = (int)(h[l * 4 + r] / 100.0 + 0.5);
    int n = l % 16;
    if (v != (4 * (l / 16) + r + 1) * (n + 1)) ok_f32_layout = 0;
    if (v != ((l / 16) + 4 * r + 1) * (n + 1)) ok_alt = 0;
  }
  printf("layout m = 4*(l/16)+r : %d   layout m = (l/16)+4*r : %d\n", ok_f32_layout, ok_alt);
  return 0;
}
