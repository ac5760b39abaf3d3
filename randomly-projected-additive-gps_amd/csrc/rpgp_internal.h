// rpgp_internal.h — declarations shared between the translation units of librpgp.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rpgp_internal {

// rows per workgroup of the matrix-core tile kernels (4 waves x one 32-row MFMA tile each)
constexpr int kMfmaBR = 128;

// Is the (column piece, right-hand-side piece) pair served by the matrix-core kernels of rpgp_mfma.hip?
bool mfma_supported(int jt, int tt);

// Launch the matrix-core form of the factorised symmetric MVM sweep (rpgp_mfma.hip) for workgroups [w0, w0 + nwg) of
// the (row block, column chunk) numbering with BR = kMfmaBR.  `rowtab` = prep table {2a, -a^2}, `coltab` = {a, exp2(-a^2)}
// (the `coldat` / `rowdat` tables of rpgp_prepare — the roles of the two tables are swapped relative to the VALU kernel).
int launch_mvm_mfma(int jt, int tt, const void *rowtab, const void *coltab, const float *V, float *slabR, float *slabT,
                    int N, int J, int ldv, int j0, int t0, int tcnt, int chunk_cols, int accumulate, int w0, int nwg,
                    int rb_first, int slab_row0, int slab_rows, hipStream_t st);

// SKI stages implemented in rpgp_kernels.hip, reused by the planned (cell-sorted) SKI product of rpgp_ski.hip
int ski_toeplitz_launch(const void *hist, int hist_is_double, const float *gp, float *H, int J, int G, int T,
                        hipStream_t st, const double *tcol);
int ski_gather_launch(const float *Z, const float *gp, const float *H, const float *V, float *out, long long M, int ldz,
                      int J, int G, int T, float scale, float noise, hipStream_t st);
size_t ski_scratch_floats(int J, int G);                  // floats of per-call scratch inside rpgp_ski_workspace_bytes
size_t ski_scratch_offset_floats(int J, int G, int T);    // ... and where it starts

// A^T B (K x T, K, T <= 64) for tall fp32 matrices with float64 accumulation on the matrix cores (rpgp_precond.hip); the
// result is written as float64 and / or float32.  `part`: gram_part_bytes(K, T) of device scratch.
size_t gram_part_bytes(int K, int T);
int gram_launch(const float *A, long long lda, const float *B, long long ldb, long long N, int K, int T, double *out64,
                float *out32, double *part, hipStream_t st);

}  // namespace rpgp_internal
