/* ORACLE — test infrastructure only (never linked, loaded or called by the product package).
 *
 * Plain C / OpenMP float64 restatement of the additive 1-D RBF kernel sum of the reference
 * (gp_models/kernels/memory_efficient_gam_kernel.py:20-30: K[i,i'] = sum_j exp(-0.5 (z_ij - z_i'j)^2), one projection at
 * a time into an N x M accumulator) and of the products the exact-GP path takes with it:
 *   oracle_kernel_f64   K = scale * K_add(Z1, Z2)                              (the dense matrix; training_routines.py:406)
 *   oracle_mvm_f64      out = scale * K_add(Z1, Z2) V (+ noise V, Z1 == Z2)    (`_matmul` inside linear_cg, reached from
 *                                                                               fitting/optimizing.py:67-71)
 * It exists so that the parity tests can afford the FULL products at the BASELINE sizes (N = 50 000: N^2 J = 5e10
 * exponentials per MVM — minutes in single-threaded numpy, seconds here on the GPU box's host cores).  Same arithmetic as
 * oracle/dense_gp.py (`additive_rbf`, `mvm`); tests/test_oracle_pinned.py checks one against the other and both against
 * the reference-generated golden vectors.  Built by oracle/cmvm.py (gcc -O3 -fopenmp; no -ffast-math: libm's exp, so the
 * values are the correctly rounded-to-<1ulp ones numpy also uses).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define COLS 1024 /* column tile: the kernel values of one row against COLS points live in L1/L2 */

/* k[c] = sum_j exp(-0.5 (z1[j] - Z2[c0 + c][j])^2), c < nc — the loop order of the reference (projection outermost) */
static void kernel_row_tile(const double *z1, const double *Z2, long c0, long nc, int J, double *k) {
  for (long c = 0; c < nc; ++c) k[c] = 0.0;
  for (int j = 0; j < J; ++j) {
    const double a = z1[j];
    for (long c = 0; c < nc; ++c) {
      const double d = a - Z2[(c0 + c) * J + j];
      k[c] += exp(-0.5 * d * d);
    }
  }
}

void oracle_kernel_f64(const double *Z1, const double *Z2, double *K, long M, long N, int J, double scale) {
#pragma omp parallel
  {
    double *k = (double *)malloc(COLS * sizeof(double));
#pragma omp for schedule(dynamic, 8)
    for (long i = 0; i < M; ++i) {
      for (long c0 = 0; c0 < N; c0 += COLS) {
        const long nc = (N - c0 < COLS) ? N - c0 : COLS;
        kernel_row_tile(Z1 + i * J, Z2, c0, nc, J, k);
        for (long c = 0; c < nc; ++c) K[i * N + c0 + c] = scale * k[c];
      }
    }
    free(k);
  }
}

/* out (M x T) = scale * K_add(Z1, Z2) @ V (N x T) [+ noise * V when add_noise (then M == N and Z1 == Z2 row for row)] */
void oracle_mvm_f64(const double *Z1, const double *Z2, const double *V, double *out, long M, long N, int J, int T,
                    double scale, double noise, int add_noise) {
#pragma omp parallel
  {
    double *k = (double *)malloc(COLS * sizeof(double));
    double *acc = (double *)malloc((size_t)T * sizeof(double));
#pragma omp for schedule(dynamic, 8)
    for (long i = 0; i < M; ++i) {
      for (int t = 0; t < T; ++t) acc[t] = 0.0;
      for (long c0 = 0; c0 < N; c0 += COLS) {
        const long nc = (N - c0 < COLS) ? N - c0 : COLS;
        kernel_row_tile(Z1 + i * J, Z2, c0, nc, J, k);
        for (long c = 0; c < nc; ++c) {
          const double kv = k[c];
          const double *v = V + (c0 + c) * T;
          for (int t = 0; t < T; ++t) acc[t] += kv * v[t];
        }
      }
      for (int t = 0; t < T; ++t) out[i * T + t] = scale * acc[t] + (add_noise ? noise * V[i * T + t] : 0.0);
    }
    free(k);
    free(acc);
  }
}

/* gZ[i][j] = -scale * sum_i' S[i][i'] (z_ij - z_i'j) exp(-0.5 (z_ij - z_i'j)^2): d/dZ of sum(W * scale K_add(Z, Z)) with
 * S = W + W^T (the analytic x1 / x2 gradients of memory_efficient_gam_kernel.py:53-58 summed for
 * x1 == x2).  S: N x N row-major.  Same arithmetic as the numpy loop it replaces in tests/test_parity_gpu.py (projection
 * outermost, row sums), so that the C2 / C3-size derivative checks take seconds. */
void oracle_bilinear_gz_f64(const double *Z, const double *S, double *gZ, long N, int J, double scale) {
#pragma omp parallel for schedule(dynamic, 8)
  for (long i = 0; i < N; ++i) {
    const double *si = S + (size_t)i * N;
    for (int j = 0; j < J; ++j) {
      const double a = Z[i * J + j];
      double acc = 0.0;
      for (long c = 0; c < N; ++c) {
        const double d = a - Z[c * J + j];
        acc += exp(-0.5 * d * d) * si[c] * d;
      }
      gZ[i * J + j] = -scale * acc;
    }
  }
}

int oracle_cmvm_version(void) { return 2; }
