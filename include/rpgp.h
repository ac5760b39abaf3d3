/*
 * rpgp.h — C-ABI of the MI355X-native additive randomly-projected GP kernel path.
 *
 * Drop-in boundary.  The reference (idelbrid/Randomly-Projected-Additive-GPs) is pure
 * Python on GPyTorch; it has no FFI of its own.  The entry points below are what a GPyTorch
 * `Kernel.forward` -> `LazyTensor` binding for this path would call; each one names the reference
 * interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) unless a name ends in `_host`;
 *   - row-major, contiguous unless a leading dimension (`ld*`) is given;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *     allocates nothing, and is re-entrant (no global mutable state besides a one-time device probe);
 *   - return value: 0 on success, otherwise a hipError_t value or one of the RPGP_E* codes below;
 *     `rpgp_error_string` turns either into text.  Python callers raise RuntimeError / ValueError.
 *
 * Kernel definition:
 *   K[i,i'] = scale * sum_{j in [j0,j1)} exp(-0.5 * (Z1[i,j] - Z2[i',j])^2)
 *   with Z = (X / lengthscale) P  (prescale)  or  (X P) / lengthscale  (postscale);
 *   for the J20 specs scale = outputscale * (1/J).
 */
#ifndef RPGP_H
#define RPGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPGP_ABI_VERSION 4

#define RPGP_EINVAL     10001 /* bad argument (shape, range, null pointer) */
#define RPGP_EWORKSPACE 10002 /* workspace too small: call the matching *_workspace_bytes */
#define RPGP_ENODEVICE  10003 /* no gfx950 device visible */
#define RPGP_ENUMERIC   10004 /* NaNs encountered in an iterative solve */

/* ABI version of the loaded library. */
int rpgp_version(void);

/* Text for a return code of any function below (static storage). */
const char *rpgp_error_string(int code);

/* One-time probe of the current device (checks gfx9 wave64, finds the DPP rotate direction used by the
 * symmetric kernel).  Called lazily by the compute entry points; exposed so hosts can fail early. */
int rpgp_init(void);

/*
 * DPA-GP diversification of the projection directions (replaces the gradient loop of rp.space_equally, rp.py:241-266, for
 * d < J): `niter` plain gradient steps of size `lr` on sum_{a != b} cos^4(angle(P_a, P_b)), then row normalisation, in
 * ONE single-workgroup launch.  P: J x d row-major DEVICE array, updated in place (J, d <= 64); final_loss (DEVICE, 1
 * float, may be NULL) receives the energy before the normalisation, as the reference returns it.
 */
int rpgp_space_equally(float *P, int J, int d, float lr, int niter, float *final_loss, void *stream);

/*
 * Projection  Z = X @ Peff   (X: N x d, Peff: d x J, Z: N x J).
 * Replaces gp_models/kernels/scaled_projection_kernel.py:21-27 (`x1.div(lengthscale)` + `projection_module(x1)`);
 * the caller folds the ARD lengthscale into Peff (prescale: diag(1/l) P; postscale: P diag(1/l)).
 */
int rpgp_project(const float *X, const float *Peff, float *Z, int64_t N, int d, int J, void *stream);

/*
 * Backward of the projection w.r.t. Peff:  dPeff = X^T @ G  (G: N x J, dPeff: d x J).
 * Replaces autograd through scaled_projection_kernel.py:21-27 for the lengthscale / learn_proj gradients.
 */
int rpgp_project_grad(const float *X, const float *G, float *dPeff, int64_t N, int d, int J, void *stream);

/*
 * Symmetric fused MVM:  out = scale * sum_{j in [j0,j1)} K_j(Z,Z) @ V  +  noise * V      (V, out: N x T)
 * K is never stored; each unordered pair (i,i') is evaluated once.
 * Replaces the dense `K @ V` inside gpytorch.utils.linear_cg reached from fitting/optimizing.py:67-71
 * (kernel built at training_routines.py:148-159,169-171,406 and evaluated by scaled_projection_kernel.py:37).
 * `ldz` = row stride of Z in floats (>= j1).  Results are deterministic (no float atomics).
 */
size_t rpgp_mvm_sym_workspace_bytes(int64_t N, int T);
int rpgp_mvm_sym(const float *Z, const float *V, float *out, int64_t N, int ldz, int T,
                 int j0, int j1, float scale, float noise,
                 void *workspace, size_t workspace_bytes, void *stream);

/*
 * Prepared (factorised) fast path for repeated symmetric MVMs with the same Z (every CG iteration of one
 * hyper-parameter step).  exp(-(a-b)^2/2) is evaluated as exp2(-a'^2) * exp2(2a'b' - b'^2) on centred, pre-scaled
 * coordinates: 3 VALU issues per pair-term instead of 4.  `rpgp_prepare` centres each projection at the midpoint of its
 * range, writes the packed row/column tables into `prep` and records whether the exponent range is safe
 * (max a'^2 < 100); `rpgp_prepare_status` copies that verdict to the host (synchronises the stream once).  Only call
 * `rpgp_mvm_sym_prepared` when fast_ok == 1 — otherwise it fills `out` with NaN — and use rpgp_mvm_sym (exact direct
 * form, any range) instead.  Same result contract, workspace and determinism as rpgp_mvm_sym; relative error per
 * kernel entry <= ~1e-7 * (1 + a'^2).   J <= 64.
 */
size_t rpgp_prepare_bytes(int64_t N, int J);
int rpgp_prepare(const float *Z, int64_t N, int ldz, int J, void *prep, size_t prep_bytes, void *stream);
int rpgp_prepare_status(const void *prep, int *fast_ok_host, float *max_abs_host, void *stream);
int rpgp_mvm_sym_prepared(const void *prep, const float *V, float *out, int64_t N, int J, int T,
                          int j0, int j1, float scale, float noise,
                          void *workspace, size_t workspace_bytes, void *stream);

/*
 * Pair-sharding of the symmetric MVM across ranks (multi-GPU).  The (row block, column chunk) workgroups of the tile
 * decomposition are numbered row block by row block; rank r of `world` executes the contiguous range
 * [total*r/world, total*(r+1)/world) (the chunk size is chosen for the per-rank share, so every rank still launches a
 * few thousand workgroups of equal size).  Each rank therefore evaluates ALL J projections on 1/world of the (i,i')
 * pairs at the full per-term efficiency and produces a partial N x T output (row AND transposed contributions);
 * the partials are summed with one all-reduce — the same message as J-sharding (`MultiDeviceKernel` counterpart,
 * training_routines.py:407-408) without the per-pair overhead of thin J-slices.  Pass the noise on exactly ONE rank
 * (its slab reduce adds noise*V for every row) and 0 on the others.  world = 1, rank = 0 is the plain call.  Workspace: rpgp_mvm_sym_range_workspace_bytes.
 */
size_t rpgp_mvm_sym_range_workspace_bytes(int64_t N, int T, int world, int rank);
int rpgp_mvm_sym_range(const float *Z, const float *V, float *out, int64_t N, int ldz, int T, int j0, int j1,
                       int world, int rank, float scale, float noise,
                       void *workspace, size_t workspace_bytes, void *stream);
int rpgp_mvm_sym_prepared_range(const void *prep, const float *V, float *out, int64_t N, int J, int T,
                                int j0, int j1, int world, int rank, float scale, float noise,
                                void *workspace, size_t workspace_bytes, void *stream);

/*
 * Rectangular fused MVM:  out = scale * sum_j K_j(Z1,Z2) @ V      (Z1: M x ., Z2: N x ., V: N x T, out: M x T)
 * Replaces K(X*,X) @ alpha and K(X,X*) blocks of the prediction strategy driven from training_routines.py:551-575.
 */
size_t rpgp_mvm_rect_workspace_bytes(int64_t M, int64_t N, int T);
int rpgp_mvm_rect(const float *Z1, const float *Z2, const float *V, float *out,
                  int64_t M, int64_t N, int ldz1, int ldz2, int T,
                  int j0, int j1, float scale,
                  void *workspace, size_t workspace_bytes, void *stream);

/*
 * Dense block  out[m, n] = scale * sum_j exp(-0.5 (Z1[m,j]-Z2[n,j])^2)   (out: M x N, row stride ldo).
 * Used for pivoted-Cholesky rows, `evaluate()`/`to_dense()` at small N and the cached-K mode.
 * Replaces LazyEvaluatedKernelTensor.evaluate_kernel()/_getitem on the kernel of training_routines.py:169-171.
 */
int rpgp_dense(const float *Z1, const float *Z2, float *out, int64_t M, int64_t N,
               int ldz1, int ldz2, int64_t ldo, int j0, int j1, float scale, void *stream);

/*
 * Bilinear derivative for backward, symmetric operator:
 *   S = L R^T + R L^T            (L, R: N x T)
 *   gZ[i,j]  = -scale * sum_i' S[i,i'] e_j(i,i') (Z[i,j]-Z[i',j])     for j in [j0,j1)   (gZ: N x ldg)
 *   gscale   = 0.5 * sum_{i,i'} S[i,i'] sum_j e_j(i,i')               (1 float, device)
 * i.e. d/dZ and d/dscale of  sum_{ii'} (L R^T)[i,i'] K[i,i'].
 * Replaces LazyTensor._quad_form_derivative triggered by loss.backward() at fitting/optimizing.py:72.
 */
size_t rpgp_bilinear_grad_workspace_bytes(int64_t N, int J);
int rpgp_bilinear_grad(const float *Z, const float *L, const float *R, float *gZ, float *gscale,
                       int64_t N, int ldz, int ldg, int T, int j0, int j1, float scale,
                       void *workspace, size_t workspace_bytes, void *stream);

/*
 * Same derivative with an explicit symmetric weight matrix S (N x N, row stride lds):
 *   gZ[i,j] = -scale * sum_i' S[i,i'] e_j(i,i') (Z[i,j]-Z[i',j]) ;  gscale = 0.5 * sum_{ii'} S[i,i'] sum_j e_j(i,i')
 * i.e. d/dZ, d/dscale of 0.5 * sum(S * K).  Used by the dense Cholesky regime (N <= max_cholesky_size, or
 * `--use_chol`, gp_experiment_runner.py:236,326) where S = Khat^-1 - alpha alpha^T is formed exactly.
 * Workspace: rpgp_bilinear_grad_workspace_bytes.
 */
int rpgp_bilinear_grad_dense(const float *Z, const float *S, float *gZ, float *gscale,
                             int64_t N, int ldz, int ldg, int64_t lds, int j0, int j1, float scale,
                             void *workspace, size_t workspace_bytes, void *stream);

/*
 * Rank-`rank` pivoted Cholesky of K(Z,Z) (greedy max-diagonal pivots): L (N x rank, row-major) with K ~= L L^T, for the
 * Woodbury preconditioner M = L L^T + sigma^2 I (GPyTorch `pivoted_cholesky`, max_preconditioner_size = 15,
 * SURVEY.md Appendix B.3).  One single-workgroup launch for N <= 2048, otherwise rank + 1 chip-wide launches (one per
 * greedy step; pivots identical: ties go to the smallest index).  `diag_work` is N + RPGP_PIVCHOL_SCRATCH floats of
 * device scratch.  J, rank <= 64.  The family variant takes weight_sum = sum_c weights[c] (the kernel's diagonal is
 * scale * weight_sum).
 */
#define RPGP_PIVCHOL_SCRATCH 8192
int rpgp_pivoted_cholesky(const float *Z, float *L, float *diag_work, int64_t N, int ldz, int J, int rank,
                          float scale, void *stream);

/*
 * Dense-matrix MVM for the cached-K mode:  out = Kd @ V + noise * V   (Kd: N x N fp32 in HBM, row stride ldk).
 * HBM-bound stream of Kd with the thin GEMM on the matrix cores (v_mfma_f32_16x16x4_f32) — SURVEY.md §8(f) rank 2.
 */
int rpgp_dense_mvm(const float *Kd, const float *V, float *out, int64_t N, int64_t ldk, int T,
                   float noise, void *stream);

/*
 * Packed symmetric cache of the additive kernel (cached-K mode without the lower triangle): every unordered pair of
 * points is evaluated once (`rpgp_symcache_build`, the fused symmetric sweep with a store in place of the products) and
 * kept in the order that sweep consumes it; `rpgp_symcache_mvm` is the same sweep with a 16-byte load in place of the J
 * exponentials: out = scale * (sum_j K_j) V + noise * V from HALF the bytes of the dense N x N matrix.  Same role as
 * rpgp_dense + rpgp_dense_mvm (the matrix GPyTorch's lazily evaluated kernel materialises for `_matmul`,
 * training_routines.py:406 with fitting/optimizing.py:69-71) for blocks of up to 12 right-hand sides per pass; wider
 * blocks take ceil(T / 12) passes over the cache — use the dense matrix for those.
 * The cached values are the unscaled sums over j in [j0, j1) in fp32, so an outputscale change needs no rebuild.
 * (world, rank): pair-sharding as in rpgp_mvm_sym_range — the cache holds this rank's share of the pairs only and the
 * product is a partial result (noise on ONE rank); world = 1, rank = 0 is the whole matrix.  The layout depends on
 * (N, world, rank) only; a cache is valid for exactly the arguments it was built with.
 * `layout` (the same value for the build and every product on that cache; size and workspace do not depend on it):
 *   RPGP_SYMCACHE_THIN — rotation order of the fused sweep; products of 1..4 right-hand sides are HBM-bound on the
 *                        N^2/2 stored values (VALU passes of up to 12 columns for wider blocks);
 *   RPGP_SYMCACHE_WIDE — 16 x 16 tiles in v_mfma_f32_16x16x4_f32 operand order; blocks of up to 16 right-hand sides per
 *                        pass on the matrix cores (the training block of 10 probes + the residual).
 */
#define RPGP_SYMCACHE_THIN 0
#define RPGP_SYMCACHE_WIDE 1
size_t rpgp_symcache_bytes(int64_t N, int world, int rank);
size_t rpgp_symcache_workspace_bytes(int64_t N, int T, int world, int rank);
int rpgp_symcache_build(const float *Z, void *cache, size_t cache_bytes, int64_t N, int ldz, int j0, int j1, int layout,
                        int world, int rank, void *stream);
int rpgp_symcache_mvm(const void *cache, size_t cache_bytes, int layout, const float *V, float *out, int64_t N, int T,
                      float scale, float noise, int world, int rank, void *workspace, size_t workspace_bytes,
                      void *stream);

/*
 * SKI path (1-D grid interpolation per projection; replaces the `GridInterpolationKernel` wrap of
 * training_routines.py:157-158 used by model_specs/additive_spread_prescale_Jd_ski.json):
 *   K ~= scale * sum_j W_j Tm W_j^T,   W_j = cubic-convolution (Keys) interpolation weights of projection j onto ONE
 *   shared regular grid of G points,  Tm[m,m'] = exp(-0.5 ((m-m') h)^2)  (symmetric Toeplitz).
 * rpgp_ski_grid       : grid_params (device: g0, h, 1/h, has_weights = 0) from the min/max of Z1 (and Z2 if given) so that
 *                       all points lie in [g_2, g_{G-3}]  (h = range / (G-5)).  The block may be 4 + J floats long: a caller
 *                       that sets has_weights = 1 and appends w_0 .. w_{J-1} gets K ~= scale * sum_j w_j W_j Tm W_j^T
 *                       from every SKI entry point (the `weighted` components of polynomial_projection_kernels.py:88-98).
 * rpgp_ski_mvm        : out = scale * sum_j W1_j Tm W2_j^T V (+ noise V when Z1 == Z2)    (out: M x T, V: N x T).
 *                       scatter (per-(chunk, projection) fixed-point LDS histograms + slabs for T <= 12: bitwise
 *                       reproducible; float atomics for wider blocks), Toeplitz product (MFMA), gather.
 *                       HBM traffic ~ 4 (N (J + T) + M (J + T)) bytes.
 * rpgp_ski_diag       : diag[i] = scale * sum_j w_i^T Tm[4x4] w_i.
 * rpgp_ski_bilinear_grad : d/dZ and d/dscale of sum((L R^T) * K) for the square operator (T <= 12); `row_scratch`
 *                       is N floats of device scratch.
 * All need `rpgp_ski_workspace_bytes(J, G, T)` bytes of workspace; G*13*4 <= 64 KB (G <= 1260).
 */
size_t rpgp_ski_workspace_bytes(int J, int G, int T);
int rpgp_ski_grid(const float *Z1, int64_t N1, int ld1, const float *Z2, int64_t N2, int ld2, int J, int G,
                  float *grid_params, void *workspace, size_t workspace_bytes, void *stream);
/* The reference's grid rule (polynomial_projection_kernels.py:54-63; the `rp_poly` / `strictly_additive` / `additive` kinds
 * with `ski: true`): every projection gets ITS OWN grid — spacing_j = (max_j - min_j) / (G - 4), bounds
 * [min_j - 2.01 spacing_j, max_j + 2.01 spacing_j], G points spanning the bounds.  grid_params: 4 + 4 J floats,
 * [., ., ., flags = 2, w_0 .. w_{J-1} = 1, (g0_j, h_j, 1/h_j) x J]; set flags = 3 and fill the w_j for per-projection output
 * scales.  Every SKI entry point reads either block form.  (With fixed projections the reference's static bounds,
 * computed once from X, scale with 1/lengthscale_j exactly like the current projected coordinates do, so recomputing the
 * bounds from the current Z reproduces them.) */
int rpgp_ski_grid_per_projection(const float *Z1, int64_t N1, int ld1, const float *Z2, int64_t N2, int ld2, int J, int G,
                                 float *grid_params, void *workspace, size_t workspace_bytes, void *stream);
int rpgp_ski_mvm(const float *Z1, const float *Z2, const float *grid_params, const float *V, float *out,
                 int64_t M, int64_t N, int ldz1, int ldz2, int J, int G, int T, float scale, float noise,
                 void *workspace, size_t workspace_bytes, void *stream);
/*
 * The three stages of rpgp_ski_mvm as separate calls (T <= 12), for the row-sharded multi-GPU SKI operator that replaces
 * `MultiDeviceKernel` around the SKI kernel (training_routines.py:407-408 with :157-158): every rank scatters ITS rows,
 * the J x G x T float64 histogram is all-reduced (RCCL), the Toeplitz product is replicated and every rank gathers ITS
 * rows.  grid_params must be built from the GLOBAL coordinate range (same block on every rank).
 *   rpgp_ski_scatter      : hist[j][g][t] (float64, J*G*T) = sum_i w(z_ij)[g] V[i][t] over the N local rows
 *   rpgp_ski_grid_product : H[j][m][t] (float32) = w_j * sum_m' Tm[m,m'] hist[j][m'][t]
 *   rpgp_ski_gather       : out[i][t] = scale * sum_j sum_k w_k(z_ij) H[j][idx0+k][t] + noise * V[i][t]  (V may be NULL if noise == 0)
 */
int rpgp_ski_scatter(const float *Z, const float *grid_params, const float *V, double *hist, int64_t N, int ldz, int J,
                     int G, int T, void *workspace, size_t workspace_bytes, void *stream);
int rpgp_ski_grid_product(const double *hist, const float *grid_params, float *H, int J, int G, int T, void *stream);
int rpgp_ski_gather(const float *Z, const float *grid_params, const float *H, const float *V, float *out, int64_t M,
                    int ldz, int J, int G, int T, float scale, float noise, void *stream);
/*
 * Planned SKI product for the square operator (Z fixed over a whole CG solve = one hyper-parameter step): rpgp_ski_plan
 * sorts every projection's points by the grid cell of their first tap ONCE (stable radix sort: deterministic order) and
 * keeps the tap fractions in that order; the scatter of every later product is then a segmented reduction over the sorted
 * points — no atomics — and the Toeplitz stage reads its first column from the plan.
 *   rpgp_ski_mvm_planned     : same result contract as rpgp_ski_mvm(Z, Z, ...) for T <= 12 (bitwise reproducible)
 *   rpgp_ski_scatter_planned : stage 1 only (the row-sharded operator: plan built on the LOCAL rows), = rpgp_ski_scatter
 *   rpgp_ski_gather_fast     : stage 3; with J * G * T floats fitting in LDS (and >= 32 768 rows) the table H is LDS-resident,
 *                              else = rpgp_ski_gather (`plan` may be NULL: the gather reads the coordinates themselves)
 * Workspace of the products: rpgp_ski_workspace_bytes(J, G, T) (RPGP_EWORKSPACE when the per-cell tap records of J * G cells
 * exceed its scratch — more than ~340 projections: fall back to rpgp_ski_mvm).  N * J < 2^31.
 */
size_t rpgp_ski_plan_bytes(int64_t N, int J, int G);
size_t rpgp_ski_plan_workspace_bytes(int64_t N, int J, int G);
int rpgp_ski_plan(const float *Z, const float *grid_params, int64_t N, int ldz, int J, int G, void *plan, size_t plan_bytes,
                  void *workspace, size_t workspace_bytes, void *stream);
int rpgp_ski_mvm_planned(const void *plan, const float *Z, const float *grid_params, const float *V, float *out, int64_t N,
                         int ldz, int J, int G, int T, float scale, float noise, void *workspace, size_t workspace_bytes,
                         void *stream);
int rpgp_ski_scatter_planned(const void *plan, const float *V, double *hist, int64_t N, int J, int G, int T, void *workspace,
                             size_t workspace_bytes, void *stream);
int rpgp_ski_gather_fast(const void *plan, const float *Z, const float *grid_params, const float *H, const float *V, float *out,
                         int64_t M, int ldz, int J, int G, int T, float scale, float noise, void *stream);
/* Pivoted Cholesky of the SKI operator (same contract as rpgp_pivoted_cholesky).  diag_work: DEVICE scratch of
 * rpgp_ski_pivoted_cholesky_work_floats(N, rank) floats — the residual diagonal, the argmax partials and, from N = 65 536 on,
 * the factor in column-major order while it is built (step m then reads m coalesced vectors instead of every line of the
 * N x rank array); RPGP_EWORKSPACE if smaller. */
size_t rpgp_ski_pivoted_cholesky_work_floats(int64_t N, int rank);
int rpgp_ski_pivoted_cholesky(const float *Z, const float *grid_params, float *L, float *diag_work, size_t diag_work_floats,
                              int64_t N, int ldz,
                              int J, int G, int rank, float scale, void *stream);
int rpgp_ski_diag(const float *Z, const float *grid_params, float *diag, int64_t N, int ldz, int J, int G,
                  float scale, void *stream);
/* Dense block K_ski(Z1, Z2) (M x N, row stride ldo) straight from the interpolation weights and the Toeplitz lags (J <= 64). */
int rpgp_ski_dense(const float *Z1, const float *Z2, const float *grid_params, float *out, int64_t M, int64_t N,
                   int ldz1, int ldz2, int64_t ldo, int J, int G, float scale, void *stream);
int rpgp_ski_bilinear_grad(const float *Z, const float *grid_params, const float *L, const float *R, float *gZ,
                           float *gscale, int64_t N, int ldz, int ldg, int J, int G, int T, float scale,
                           void *workspace, size_t workspace_bytes, float *row_scratch, void *stream);
/* Same with per-projection output scales in the grid parameter block (see rpgp_ski_grid): additionally gcomp[j] (DEVICE,
 * J floats) = the part of gscale contributed by projection j (it carries w_j; divide by w_j for the unweighted component
 * sum); row_scratch: N * (J + 1) floats. */
int rpgp_ski_bilinear_grad_comp(const float *Z, const float *grid_params, const float *L, const float *R, float *gZ,
                                float *gscale, float *gcomp, int64_t N, int ldz, int ldg, int J, int G, int T,
                                float scale, void *workspace, size_t workspace_bytes, float *row_scratch,
                                void *stream);

/*
 * float64 parity kernels of the SKI operator (csrc/rpgp_ski_f64.hip): `--double` (/root/reference/training_routines.py:481) for
 * the `ski: true` specifications (/root/reference/training_routines.py:157-158).  Same operator and grid-parameter layout as the
 * float32 entry points above, every array float64 (grid_params included: [g0, h, 1/h, flags, w_j .., (g0_j, h_j, 1/h_j) ..]);
 * a thread per output element, the scatter with float64 hardware atomics (sums differ in the last bits between runs), the
 * Toeplitz stage a plain O(G^2) sum — written for clarity, not speed.  Any T (the host passes <= 64 columns per call).
 *   rpgp_ski_f64_mvm           out (M x T) = scale sum_j w_j W1_j Tm_j W2_j^T V (+ noise V: square operator only)
 *   rpgp_ski_f64_diag / _dense the diagonal / a dense block K_ski(Z1, Z2)
 *   rpgp_ski_f64_bilinear_grad d/dZ, d/dscale (device scalar) and (gcomp != NULL, J doubles) the per-projection parts of d/dscale
 *                              (each carries its w_j) of sum((L R^T) * K_ski(Z, Z))
 */
size_t rpgp_ski_f64_workspace_bytes(int J, int G, int T);
int rpgp_ski_f64_mvm(const double *Z1, const double *Z2, const double *grid_params, const double *V, double *out, int64_t M,
                     int64_t N, int ldz1, int ldz2, int J, int G, int T, double scale, double noise, void *workspace,
                     size_t workspace_bytes, void *stream);
int rpgp_ski_f64_diag(const double *Z, const double *grid_params, double *diag, int64_t N, int ldz, int J, int G, double scale,
                      void *stream);
int rpgp_ski_f64_dense(const double *Z1, const double *Z2, const double *grid_params, double *out, int64_t M, int64_t N, int ldz1,
                       int ldz2, int64_t ldo, int J, int G, double scale, void *stream);
int rpgp_ski_f64_bilinear_grad(const double *Z, const double *grid_params, const double *L, const double *R, double *gZ,
                               double *gscale, double *gcomp, int64_t N, int ldz, int ldg, int J, int G, int T, double scale,
                               void *workspace, size_t workspace_bytes, void *stream);
/* The stages of the float64 product and derivative as separate calls — `--double` for the row-sharded operator (one process per
 * GPU: MultiDeviceKernel around the grid-interpolation kernel, /root/reference/training_routines.py:407-408 with :157-158, under
 * :481): scatter the LOCAL rows into columns [hoff, hoff + T) of a J x G x HT float64 histogram (zero_first: clear it), all-reduce
 * it (caller), replicated grid product, gather the local rows; the derivative from the all-reduced J x G x 2T block
 * [W^T L | W^T R].  rpgp_ski_f64_bilinear_finish workspace: J * G * 2T + J doubles (+ 512 B). */
int rpgp_ski_f64_scatter(const double *Z, const double *grid_params, const double *V, double *hist, int64_t N, int ldz, int J,
                         int G, int T, int HT, int hoff, int zero_first, void *stream);
int rpgp_ski_f64_grid_product(const double *hist, const double *grid_params, double *H, int J, int G, int T, int weighted,
                              void *stream);
int rpgp_ski_f64_gather(const double *Z, const double *grid_params, const double *H, const double *V, double *out, int64_t M,
                        int ldz, int J, int G, int T, double scale, double noise, void *stream);
int rpgp_ski_f64_bilinear_finish(const double *Z, const double *grid_params, const double *hist2, const double *L, const double *R,
                                 double *gZ, double *gscale, double *gcomp, int64_t N, int ldz, int ldg, int J, int G, int T,
                                 double scale, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The derivative in two stages for the row-sharded SKI operator (MLL backward of a model whose rows are split over ranks):
 *   rpgp_ski_bilinear_scatter : hist2[j][g][0..T) = W_j^T L, hist2[j][g][T..2T) = W_j^T R over the N LOCAL rows (float64,
 *                               J * G * 2T) — all-reduce it over the ranks —
 *   rpgp_ski_bilinear_finish  : Toeplitz products of the 2T columns, then gZ (local rows), gscale and (gcomp != NULL:
 *                               row_scratch N * (J + 1) floats) gcomp as PARTIAL sums over the local rows (all-reduce them).
 * T <= 12; rpgp_ski_bilinear_grad[_comp] is exactly scatter + finish.
 */
int rpgp_ski_bilinear_scatter(const float *Z, const float *grid_params, const float *L, const float *R, double *hist2,
                              int64_t N, int ldz, int J, int G, int T, void *workspace, size_t workspace_bytes,
                              void *stream);
/* stage 1 with the points of Z taken from its plan (rpgp_ski_plan): two cell-sorted scatters, no atomics */
int rpgp_ski_bilinear_scatter_planned(const void *plan, const float *L, const float *R, double *hist2, int64_t N, int J, int G,
                                      int T, void *workspace, size_t workspace_bytes, void *stream);
int rpgp_ski_bilinear_finish(const float *Z, const float *grid_params, const double *hist2, const float *L, const float *R,
                             float *gZ, float *gscale, float *gcomp, int64_t N, int ldz, int ldg, int J, int G, int T,
                             float scale, void *workspace, size_t workspace_bytes, float *row_scratch, void *stream);

/*
 * Generalised additive family — the other members behind the same operator:
 *     K[i,i'] = scale * sum_{c < ncomp} weights[c] * phi_kind( columns [c*group, (c+1)*group) of Z )
 *   kind RBF      exp(-r^2/2), r^2 summed over the group's columns   (k > 1 sub-kernels of training_routines.py:172-174,
 *                 the product groups of polynomial_projection_kernels.py:70-86; group in {1,2,3,4,5,8,10,20})
 *   kind MATERN15 (1 + sqrt3 r) exp(-sqrt3 r)     kind IMQ  (1 + r^2)^(-1/2)     kind COSINE  cos(pi r)
 *                 (`kernel_type` of training_routines.py:47-88, imq_kernel.py:8-9; group must be 1)
 * `weights` (DEVICE, ncomp floats, required) are the per-component output scales (the `weighted` ScaleKernels of
 * polynomial_projection_kernels.py:88-98; all 1/J for additive_rp).  Z holds ncomp*group columns, already divided by
 * the lengthscales.  Same tile / dense kernels as the hot path with a different kernel-function policy; no factorised
 * fast path; float64 and the combinations not instantiated here: rpgp_family_generic_* below.
 * rpgp_family_bilinear_grad*: gZ as rpgp_bilinear_grad; gcomp[c] (DEVICE, ncomp) = 0.5 sum_ii' S_ii' phi_c(i,i'), the
 * unweighted per-component sums (d/d weights[c] = scale * gcomp[c]; d/d scale = sum_c weights[c] gcomp[c]).
 */
#define RPGP_KIND_RBF 0
#define RPGP_KIND_MATERN15 1
#define RPGP_KIND_IMQ 2
#define RPGP_KIND_COSINE 3
/* OR-ed into the `kind` of the rpgp_family_generic_* calls only: a group is the PRODUCT of `group` 1-D sub-kernels
 * (the ProductKernel groups of polynomial_projection_kernels.py:70-86) instead of one radial group-dimensional sub-kernel
 * (training_routines.py:172-174).  The two are the same function for the RBF. */
#define RPGP_KIND_PRODUCT 16
typedef struct rpgp_family {
  int kind, group, ncomp;
  const float *weights;
} rpgp_family;
size_t rpgp_family_mvm_workspace_bytes(int64_t M, int64_t N, int T, int sym);
int rpgp_family_mvm_sym(const rpgp_family *fam, const float *Z, const float *V, float *out, int64_t N, int ldz, int T,
                        float scale, float noise, void *workspace, size_t workspace_bytes, void *stream);
int rpgp_family_mvm_rect(const rpgp_family *fam, const float *Z1, const float *Z2, const float *V, float *out,
                         int64_t M, int64_t N, int ldz1, int ldz2, int T, float scale, void *workspace,
                         size_t workspace_bytes, void *stream);
int rpgp_family_dense(const rpgp_family *fam, const float *Z1, const float *Z2, float *out, int64_t M, int64_t N,
                      int ldz1, int ldz2, int64_t ldo, float scale, void *stream);
size_t rpgp_family_bilinear_grad_workspace_bytes(int64_t N, int ncols, int ncomp);
int rpgp_family_bilinear_grad(const rpgp_family *fam, const float *Z, const float *L, const float *R, float *gZ,
                              float *gcomp, int64_t N, int ldz, int ldg, int T, float scale, void *workspace,
                              size_t workspace_bytes, void *stream);
int rpgp_family_bilinear_grad_dense(const rpgp_family *fam, const float *Z, const float *S, float *gZ, float *gcomp,
                                    int64_t N, int ldz, int ldg, int64_t lds, float scale, void *workspace,
                                    size_t workspace_bytes, void *stream);
int rpgp_family_pivoted_cholesky(const rpgp_family *fam, const float *Z, float *L, float *diag_work, int64_t N,
                                 int ldz, int rank, float scale, float weight_sum, void *stream);

/* The same family with RUNTIME (kind, group) in float32 or float64 (csrc/rpgp_family_generic.hip) — what the templated
 * kernels above do not instantiate: `--double` (training_routines.py:481) for every member, k > 1 sub-kernels of the
 * non-RBF types in the RADIAL form `additive_rp` builds (training_routines.py:172-174: kernel(active_dims = a group of k
 * columns); imq_kernel.py:8-9) or, with `kind | RPGP_KIND_PRODUCT`, in the PRODUCT-of-1-D form of the rp_poly kinds
 * (polynomial_projection_kernels.py:70-86), any group size <= 32 (ncomp * group <= 64).  `dtype`: RPGP_F32 / RPGP_F64 selects the element
 * type of EVERY pointer argument (weights, Z, V, L, R, S, outputs).  Parity path: lane-owns-row, library transcendental
 * functions, one writer per output (bitwise reproducible).
 *   mvm:      out (M x T) = scale * K(Z1, Z2) V (+ noise V when Z2 == NULL: the symmetric operator on Z1, M == N); T <= 16
 *   dense:    out (M x N, leading dimension ldo) = scale * K(Z1, Z2)
 *   bilinear: gZ / gcomp exactly as rpgp_family_bilinear_grad (S == NULL: weights S = L R^T + R L^T from the N x T factors)
 *             or rpgp_family_bilinear_grad_dense (S: explicit symmetric N x N matrix, leading dimension lds).
 */
int rpgp_family_generic_mvm(int dtype, int kind, int group, int ncomp, const void *weights, const void *Z1, const void *Z2,
                            const void *V, void *out, int64_t M, int64_t N, int ldz1, int ldz2, int T, double scale,
                            double noise, void *stream);
int rpgp_family_generic_dense(int dtype, int kind, int group, int ncomp, const void *weights, const void *Z1, const void *Z2,
                              void *out, int64_t M, int64_t N, int ldz1, int ldz2, int64_t ldo, double scale, void *stream);
size_t rpgp_family_generic_bilinear_workspace_bytes(int dtype, int64_t N, int ncomp);
int rpgp_family_generic_bilinear(int dtype, int kind, int group, int ncomp, const void *weights, const void *Z, const void *L,
                                 const void *R, const void *S, void *gZ, void *gcomp, int64_t N, int ldz, int ldg, int T,
                                 int64_t lds, double scale, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Native mBCG executor (replaces gpytorch.utils.linear_cg as configured at gp_experiment_runner.py:324-329; algorithm in
 * SURVEY.md Appendix B.2).  Solves (A) X = rhs for T <= 16 right-hand sides with A described by `rpgp_operator`
 * (noise included), optional Woodbury preconditioner M = L L^T + sigma2 I given as L (N x k, k <= 16) and
 * Cinv = (sigma2 I + L^T L)^-1 (k x k, FLOAT64: the capacitance system is ill-conditioned).  Right-hand-side columns are normalised internally; the loop stops when the mean
 * column residual norm < tolerance after >= min_iter iterations (tested every `check_every` iterations — the only host
 * synchronisations) or at max_iter.  The first `hist_len` (<= 64) iterations' alpha / beta coefficients are returned in
 * HOST arrays laid out [hist_len][16] for the Lanczos tridiagonals.  Returns RPGP_ENUMERIC on NaNs.
 * stagnation_window > 0: also stop (without convergence: *mean_resid_host stays >= tolerance) when the best mean
 * residual has not improved by 1 % over that many consecutive tests — the fp32 floor of a badly conditioned system.
 */
#define RPGP_OP_FUSED 0           /* rpgp_mvm_sym on Z */
#define RPGP_OP_FUSED_PREPARED 1  /* rpgp_mvm_sym_prepared on prep */
#define RPGP_OP_SKI 2             /* rpgp_ski_mvm on Z + grid_params; prep != NULL: an rpgp_ski_plan of Z (planned product) */
#define RPGP_OP_DENSE 3           /* cached-K: symmetric Kd (N x N, row stride ldk) in HBM, applied by rpgp_dense_mvm */
#define RPGP_OP_FAMILY 4          /* rpgp_family_mvm_sym on Z + family */
#define RPGP_OP_SYMCACHE 5        /* packed symmetric cache: Kd = the cache, ldk = its size in bytes, G = its layout; scale, noise */
#define RPGP_OP_SUM 6             /* sum of G unsharded operators on the same N rows: prep -> rpgp_operator[G] in HOST memory (each
                                     part with its own scale / noise; no nesting) — the additive kernels whose multiplicative
                                     groups differ in size (general_rp_poly, training_routines.py:192-207) */
typedef struct rpgp_operator {
  int kind;
  int64_t N;
  int J, ldz, j0, j1, G;
  float scale, noise;
  const float *Z;
  const void *prep;
  const float *grid_params;
  const float *Kd;
  int64_t ldk;
  const rpgp_family *family;
  int world, rank;              /* pair-shard of RPGP_OP_FUSED / FUSED_PREPARED / SYMCACHE (0 or 1, 0: the whole operator) */
} rpgp_operator;

/*
 * Sharded solves (one process per GPU).  `fn(ctx, buf, count, dtype, stream)` is the in-place SUM all-reduce described
 * under "Multi-GPU pieces" below; the executor calls it between its enqueue-only phases on the launch stream, so a
 * sharded solve still has no host synchronisation per iteration (the convergence flag stays on the device).
 *   RPGP_SHARD_PARTIAL  vectors are replicated; the operator (RPGP_OP_FUSED / FUSED_PREPARED with (world, rank) or a
 *                       [j0, j1) slice, RPGP_OP_SYMCACHE with (world, rank)) yields this rank's partial product, the noise
 *                       term is added on rank 0 only, ONE all-reduce of the N x T block per iteration;
 *   RPGP_SHARD_ROWS     RPGP_OP_SKI on this rank's rows (op->N local rows, T <= 12; rhs / x / L are the local rows; op->N
 *                       may be 0): the J x G x T float64 histogram and two 288-double reduction vectors per iteration are
 *                       all-reduced; global_N = total number of rows (iteration cap).  grid_params must come from the
 *                       GLOBAL coordinate range and Cinv from the all-reduced capacitance matrix.
 * reducer == NULL, mode RPGP_SHARD_NONE or fn == NULL: the single-GPU solve.
 */
#define RPGP_SHARD_NONE 0
#define RPGP_SHARD_PARTIAL 1
#define RPGP_SHARD_ROWS 2
typedef int (*rpgp_allreduce_fn)(void *ctx, void *buf, size_t count, int dtype, void *stream);
typedef struct rpgp_reducer {
  int mode, world, rank;
  int64_t global_N;
  rpgp_allreduce_fn fn;
  void *ctx;
} rpgp_reducer;
/* (The hipGraph form of the iteration loop — SURVEY.md §8(f) rank 3 — was built in round 5, measured 6 - 17 % slower than the
 *  queue-ahead executor and is parked as tools/experiments/r6_removed_forms.patch; ABI 3 no longer exports its switch.) */
size_t rpgp_mbcg_workspace_bytes(const rpgp_operator *op, int T, int precond_rank);
int rpgp_mbcg_solve(const rpgp_operator *op, const float *rhs, float *x, int T, int max_iter, int min_iter,
                    int hist_len, int check_every, int stagnation_window, float tolerance, int precond_rank,
                    const float *L,
                    const double *Cinv, float precond_sigma2, const rpgp_reducer *reducer, float *alpha_hist_host,
                    float *beta_hist_host, int *iterations_host, float *mean_resid_host, void *workspace,
                    size_t workspace_bytes, void *stream);

/*
 * Stochastic Lanczos quadrature estimate of log|A| from the CG coefficient histories rpgp_mbcg_solve returns (HOST
 * arithmetic, no device work): the first `num_probes` columns of alpha_hist / beta_hist ([iters][ld] floats) are unit-norm
 * probe columns;  log|A| ~ (n / num_probes) sum_p sum_m Q_p[0][m]^2 log lambda_pm  over the Lanczos tridiagonals
 * T[k][k] = 1/alpha_k + beta_{k-1}/alpha_{k-1}, T[k][k+1] = sqrt(beta_k)/alpha_k.  Replaces what GPyTorch's
 * `inv_quad_logdet` does with the `t_mat` of `linear_cg` (lanczos_tridiag_to_diag + the log-det sum), reached from
 * `-mll(output, train_y)` at /root/reference/fitting/optimizing.py:67-72.  RPGP_ENUMERIC: the QL iteration did not converge.
 */
int rpgp_slq_logdet(const float *alpha_hist, const float *beta_hist, int iters, int ld, int num_probes, double n,
                    double *logdet_out);

/*
 * The small kernels AROUND the solve of one optimiser step of the flagship model (csrc/rpgp_step.hip): each call is one launch
 * for a stretch the reference runs as a chain of element-wise torch launches on d + 3 scalars and a few
 * N x 11 blocks — `-mll(model(train_x), train_y)` / `loss.backward()` at /root/reference/fitting/optimizing.py:67-72 on the model
 * of /root/reference/training_routines.py:131-189, 325-410.  All enqueue on `stream`; only rpgp_step_hyper waits (for three floats).
 *
 * rpgp_step_hyper: raw parameters -> outputscale = softplus(raw_os), noise = softplus(raw_noise) + min_noise, ls = softplus(raw_ls)
 *   (n_ls = 1, or d with `prescale`, or J), Peff[d x J] = P / ls with P = W^T (W: the J x d weight of the frozen Linear
 *   projection).  dev_out (device, 8 + 2 n_ls floats): [0] outputscale [1] noise [2] mean [3] sigmoid(raw_os) [4] sigmoid(raw_noise)
 *   [8..) ls, then sigmoid(raw_ls).  The three scalars the other entry points take BY VALUE come back through pinned memory
 *   (the call spins until the kernel has published them): *outputscale_host, *noise_host, *mean_host.
 * rpgp_step_probes: z = L e1 + sqrt_noise e2 (L: N x k, e1: k x p, e2: N x p standard normal draws — GPyTorch's preconditioner-
 *   distributed probe vectors), full_rhs[N x (p + 1)] = [z | y - *mean_dev].  p <= 16, k <= 64.  The probes are not normalised:
 *   rpgp_mbcg_solve normalises every column itself and returns Khat^-1 of the columns as given.
 * rpgp_step_value: inv_quad = sum_i full_rhs[i][col] solves[i][col] (N x T blocks);  out2[0] = (inv_quad + logdet) c1 + c2,
 *   out2[1] = inv_quad.  The workspace's first 4 bytes are an arrival counter: ZERO on entry (zero the buffer once), zero
 *   again on exit; it must not be shared between streams.  `ticket_out` (optional): the kernel also POSTS out2[0] to pinned
 *   host memory and *ticket_out names the slot (-1: none); rpgp_step_value_wait(ticket, &v) returns that value on the calling
 *   thread as soon as the kernel has run — the training loop's per-step `loss.item()` (fitting/optimizing.py:76) without
 *   waiting for the derivative and the optimiser update queued behind it.  RPGP_EINVAL: the ticket is older than 16 posts, from
 *   another thread, or the value has not arrived within 2 s (read out2[0] the ordinary way then).
 * rpgp_step_lr: the two sides of the bilinear derivative from the solve of [probes | r] (solves: N x (p + 1)):
 *   left = [solves_c gq / p | -gq alpha], right = [pre_probes (row stride ldp) | alpha], gq = g[0] * gscale (g: device scalar, the
 *   incoming gradient); partials[2 b], partials[2 b + 1] = workgroup b's sum(left * right), sum(alpha); *nparts_out workgroups.
 * rpgp_step_hyper_backward: dPeff (d x J, gradient w.r.t. Peff, to be scaled by zfac) -> gradients of the raw parameters:
 *   g_raw_ls = -sum(dPeff P) / ls^2 sigmoid(raw_ls) (per row / column / overall), g_raw_os = gs_scale gs[0] sigmoid(raw_os),
 *   g_raw_noise = (sum partials[2 b] + g[0] dlp_over_n) sigmoid(raw_noise), g_mean = -2 gq sum partials[2 b + 1].
 */
int rpgp_step_hyper(const float *raw_ls, int n_ls, const float *raw_os, const float *raw_noise, const float *mean,
                    const float *W, int d, int J, int prescale, float min_noise, float *Peff, float *dev_out,
                    float *outputscale_host, float *noise_host, float *mean_host, void *stream);
int rpgp_step_probes(const float *L, int k, const float *e1, const float *e2, float sqrt_noise, const float *y,
                     const float *mean_dev, int64_t N, int p, float *full_rhs, void *stream);
size_t rpgp_step_value_workspace_bytes(void);
int rpgp_step_value(const float *full_rhs, const float *solves, int64_t N, int T, int col, double logdet, double c1, double c2,
                    float *out2, void *workspace, size_t workspace_bytes, int *ticket_out, void *stream);
int rpgp_step_value_wait(int ticket, float *value_host);
size_t rpgp_step_lr_workspace_bytes(void);
int rpgp_step_lr(const float *solves, const float *pre_probes, int64_t ldp, const float *g, float gscale, int64_t N, int p,
                 float *left, float *right, float *partials, int *nparts_out, void *stream);
int rpgp_step_hyper_backward(const float *dPeff, const float *W, int d, int J, int n_ls, int prescale, float zfac,
                             const float *hyper_dev, const float *gs, const float *partials, int nparts, const float *g,
                             float gscale, float dlp_over_n, float gs_scale, float *g_raw_ls, float *g_raw_os,
                             float *g_raw_noise, float *g_mean, void *stream);

/*
 * Multi-GPU pieces (one process per GPU; replaces `MultiDeviceKernel(kernel, devices, devices[0])`,
 * training_routines.py:407-408).
 *
 * rpgp_allreduce_fn: in-place SUM all-reduce of `count` elements (dtype RPGP_F32 / RPGP_F64) of the DEVICE buffer
 * `buf` over the ranks, ENQUEUED on `stream` (no host synchronisation); every rank must end with bit-identical
 * values.  The host binds it to its communicator: `rpgp_comm_allreduce` below (ctx = the rpgp_comm), or a callback
 * that issues the RCCL collective on the same stream (torch.distributed.all_reduce in the Python host).
 *
 * rpgp_comm: SUM all-reduce over IPC-mapped peer buffers for the small, latency-bound messages of the sharded solve
 * (N x T partial products, J x G x T grid histograms, inner products).  One kernel launch per call: every rank publishes
 * its data in a staging buffer that all peers have mapped (hipIpcGetMemHandle / hipIpcOpenMemHandle), raises a flag in
 * each peer's memory, waits for the peers' flags and adds the W contributions in rank order (one-shot), or reduces its
 * own 1/W chunk and writes it to all peers (two-shot, messages > 512 KB).  Bootstrap: every rank calls
 * rpgp_comm_create, the RPGP_COMM_HANDLE_BYTES handles are exchanged out of band in rank order (the Python host uses
 * torch.distributed.all_gather_object), then rpgp_comm_connect.  A message may hold at most `max_bytes` bytes.
 * Waits are bounded (20 s): a lost peer raises the comm's error word (rpgp_comm_error) instead of hanging the GPU.
 */
#define RPGP_F32 0
#define RPGP_F64 1
#define RPGP_COMM_HANDLE_BYTES 64
typedef struct rpgp_comm rpgp_comm;
int rpgp_comm_create(int world, int rank, size_t max_bytes, rpgp_comm **out, void *handle_out);
int rpgp_comm_connect(rpgp_comm *comm, const void *all_handles);
size_t rpgp_comm_capacity(const rpgp_comm *comm);
int rpgp_comm_allreduce(void *comm, void *buf, size_t count, int dtype, void *stream);
int rpgp_comm_error(rpgp_comm *comm, int *error_host);
int rpgp_comm_destroy(rpgp_comm *comm);

/*
 * Woodbury preconditioner M = L L^T + sigma^2 I outside the mBCG executor (GPyTorch's `_preconditioner` closure and its
 * `precond_lt` / probe-vector solves, reached from `mll()` at fitting/optimizing.py:69-72): the float64 Gram products and
 * the cancelling update.
 *   rpgp_gram_f64: out[K x T] (row-major float64) = A^T B for fp32 A (N x K, leading dimension lda) and B (N x T, ldb);
 *                  exact products, float64 sums in a fixed order.  K, T <= 64.  Workspace: rpgp_gram_f64_workspace_bytes.
 *   rpgp_woodbury_apply: out[N x T] = (float)(((double)R - L Tm) / noise), Tm: K x T float64 (row-major).
 */
size_t rpgp_gram_f64_workspace_bytes(int K, int T);
int rpgp_gram_f64(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t N, int K, int T, double *out,
                  void *workspace, size_t workspace_bytes, void *stream);
int rpgp_woodbury_apply(const float *L, int64_t ldl, const float *R, int64_t ldr, const double *Tm, double noise,
                        float *out, int64_t ldo, int64_t N, int K, int T, void *stream);
/* The same with t = Cinv gram_LR formed inside the kernel (gram_LR = L^T R from rpgp_gram_f64, Cinv from rpgp_woodbury_setup):
 * M^-1 R in two launches (Gram product, this) instead of four. */
int rpgp_woodbury_apply_cinv(const float *L, int64_t ldl, const float *R, int64_t ldr, const double *gram_LR,
                             const double *Cinv, double noise, float *out, int64_t ldo, int64_t N, int K, int T,
                             void *stream);
/* One launch for the K x K capacitance matrix C = gram + noise I (gram = L^T L, row-major float64, K <= 64): chol = its lower
 * Cholesky factor, cinv = C^-1, logdet[0] = log|C| (all float64, device).  A non-positive pivot fills the outputs with NaN. */
int rpgp_woodbury_setup(const double *gram, double noise, int K, double *chol, double *cinv, double *logdet, void *stream);
/* The same; *logdet_pinned_host (may be NULL) is PINNED host memory (hipHostMalloc) that the kernel writes log|C| to as well:
 * readable by the host once the stream has passed the launch — no device-to-host copy. */
int rpgp_woodbury_setup_pinned(const double *gram, double noise, int K, double *chol, double *cinv, double *logdet,
                               double *logdet_pinned_host, void *stream);

/*
 * Float64 variants for `--double` (training_routines.py:481).  Same contracts as the fp32 entry points of the same
 * name; parity path (software exp, no symmetry exploitation, no workspace).  rpgp_mvm_f64 covers both the square
 * (Z1 == Z2, optional noise) and the rectangular product; `row_scratch` is N doubles of device scratch.
 */
int rpgp_project_f64(const double *X, const double *Peff, double *Z, int64_t N, int d, int J, void *stream);
int rpgp_project_grad_f64(const double *X, const double *G, double *dPeff, int64_t N, int d, int J, void *stream);
int rpgp_mvm_f64(const double *Z1, const double *Z2, const double *V, double *out, int64_t M, int64_t N,
                 int ldz1, int ldz2, int T, int j0, int j1, double scale, double noise, void *stream);
int rpgp_dense_f64(const double *Z1, const double *Z2, double *out, int64_t M, int64_t N, int ldz1, int ldz2,
                   int64_t ldo, int j0, int j1, double scale, void *stream);
int rpgp_bilinear_grad_f64(const double *Z, const double *L, const double *R, double *gZ, double *gscale,
                           int64_t N, int ldz, int ldg, int T, int j0, int j1, double scale,
                           double *row_scratch, void *stream);
int rpgp_bilinear_grad_dense_f64(const double *Z, const double *S, double *gZ, double *gscale, int64_t N,
                                 int ldz, int ldg, int64_t lds, int j0, int j1, double scale,
                                 double *row_scratch, void *stream);

/*
 * Measurement hook used by bench.py (roofline.achieved): between begin/end every rpgp_mvm_sym / rpgp_mvm_rect call
 * records a HIP-event pair on its stream around the dominant fused tile kernel launch(es) (the small slab-reduce
 * launch is outside the pair).  `rpgp_profile_end` synchronises those events and returns the mean duration (ms)
 * and the number of calls in HOST memory.  Not thread-safe; off by default.
 */
int rpgp_profile_begin(void);
int rpgp_profile_end(float *avg_ms_host, int *count_host);

/* Which kernel `rpgp_mvm_sym_prepared` launches for a single-GPU N x N operator with J projections and T right-hand sides on
 * THIS process' settings (benchmark / profile labelling only; no reference counterpart — GPyTorch has one `_matmul`):
 *   0 = mvm_fact_kernel (compiler-scheduled), 1 = mvm_fact_asm_kernel (hand-scheduled loop, rpgp_fact_asm.hip),
 *   2 = mvm_fact_asm_thin_kernel<J> (the same schedule generated for a sweep of exactly J = 2 / 3 / 4 / 5 / 8 / 10 projections:
 *   a rank's slice under the J-split, or a model with that many projections).  Negative: RPGP_EINVAL. */
int rpgp_prepared_kernel_id(int64_t N, int J, int T);

/*
 * Phase markers (SURVEY.md §5 "Tracing / profiling"; the loop they annotate is fitting/optimizing.py:65-76): roctx ranges
 * pushed / popped on the calling thread, recorded by `rocprofv3 --marker-trace`.  The library marks its own phases (one mBCG
 * solve, every 8th CG iteration, all-reduce hook calls); the host stack marks projection + tables, the preconditioner, the
 * derivative and the optimiser update through these two calls.  ROCm's librocprofiler-sdk-roctx is bound at first use ONLY if
 * it is already loaded in the process (rocprofv3 --marker-trace preloads it): otherwise rpgp_range_available() == 0 and the
 * calls do nothing.
 */
int rpgp_range_push(const char *name);
int rpgp_range_pop(void);
int rpgp_range_available(void);

#ifdef __cplusplus
}
#endif
#endif /* RPGP_H */
