// rpgp_cg.hip — native mBCG executor: preconditioned batched conjugate gradients with Lanczos coefficients
//  as ONE host call.  The operator application is a direct call into the fused
// kernels of rpgp_kernels.hip; the vector recurrences are four fused kernels per iteration whose scalars (alpha, beta,
// r.z, residual norms) live in device memory, so there is no host synchronisation except the periodic convergence test.
//
// Per iteration (T <= 16 right-hand sides, row-major N x T):
//   Ap = A p                                            rpgp_mvm_sym[_prepared] / rpgp_ski_mvm
//   k_pAp    : partial sums of p.Ap per column
//   k_update : alpha = rz / pAp ; x += alpha p ; r -= alpha Ap ; partial |r|^2 ; partial L^T r (preconditioner)
//   k_precond: w = Cinv (L^T r) ; z = (r - L w) / sigma^2 ; partial r.z            (identity preconditioner: z = r)
//   k_direct : beta = rz' / rz ; p = z + beta p ; records alpha/beta history ; mean residual norm
// Partial sums are per-workgroup slabs reduced in a fixed order by the consuming kernel's prologue (deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#include "../../include/rpgp.h"

namespace {

constexpr int kMaxT = 16;
constexpr int kMaxK = 16;          // preconditioner rank
constexpr int kMaxBlocks = 256;
constexpr int kMaxHist = 64;        // Lanczos coefficients kept for at most this many iterations

#define CG_CHECK(expr)                            \
  do {                                            \
    hipError_t _e = (expr);                       \
    if (_e != hipSuccess) return (int)_e;         \
  } while (0)

// device-resident scalar state
struct CgState {
  float rz[2][kMaxT];       // r.z of the current iterate, ping-pong by iteration parity (no intra-kernel race)
  float rhs_norm[kMaxT];
  float resid[kMaxT];       // residual norms of the current iterate
  float mean_resid;
  int rhs_zero[kMaxT];
};

__device__ __forceinline__ float block_sum(float v, float *sh) {
  // 256 threads -> one value (all threads get it)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// partial[blk][t] = sum over the block's rows of a[i][t] * b[i][t]
__global__ __launch_bounds__(256) void k_coldot(const float *__restrict__ a, const float *__restrict__ b,
                                                float *__restrict__ partial, long long N, int T) {
  __shared__ float sh[4];
  float acc[kMaxT];
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) acc[t] = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
#pragma unroll
    for (int t = 0; t < kMaxT; ++t)
      if (t < T) acc[t] = __builtin_fmaf(a[i * T + t], b[i * T + t], acc[t]);
  }
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) {
    if (t < T) {
      const float s = block_sum(acc[t], sh);
      if (threadIdx.x == 0) partial[(size_t)blockIdx.x * T + t] = s;
    }
  }
}

// r = rhs / |rhs| (columns with |rhs| < 1e-10 are flagged zero and left unscaled), x = 0
__global__ __launch_bounds__(256) void k_normalise(const float *__restrict__ rhs, const float *__restrict__ partial,
                                                   int nparts, float *__restrict__ r, float *__restrict__ x,
                                                   CgState *__restrict__ st, long long N, int T) {
  __shared__ float snorm[kMaxT];
  if ((int)threadIdx.x < T) {
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * T + threadIdx.x];
    float nrm = sqrtf(s);
    const int zero = nrm < 1e-10f;
    if (zero) nrm = 1.0f;
    snorm[threadIdx.x] = nrm;
    if (blockIdx.x == 0) {
      st->rhs_norm[threadIdx.x] = nrm;
      st->rhs_zero[threadIdx.x] = zero;
    }
  }
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256)
    for (int t = 0; t < T; ++t) {
      r[i * T + t] = rhs[i * T + t] / snorm[t];
      x[i * T + t] = 0.f;
    }
}

// partial_w[blk][kk][t] = sum_i L[i][kk] r[i][t]       (L: N x K row-major)
__global__ __launch_bounds__(256) void k_Ltr(const float *__restrict__ L, const float *__restrict__ r,
                                             float *__restrict__ partial_w, long long N, int T, int K,
                                             long long rows_per_block) {
  __shared__ float sL[64 * kMaxK];
  __shared__ float sR[64 * kMaxT];
  const long long n0 = (long long)blockIdx.x * rows_per_block;
  const long long n1 = (n0 + rows_per_block < N) ? n0 + rows_per_block : N;
  const int kk = threadIdx.x / kMaxT, t = threadIdx.x % kMaxT;   // 16 x 16 threads
  float acc = 0.f;
  for (long long c0 = n0; c0 < n1; c0 += 64) {
    const int nr = (int)((n1 - c0 < 64) ? n1 - c0 : 64);
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * K; e += 256) {
      const int rr = e / K;
      sL[rr * kMaxK + e % K] = rr < nr ? L[(c0 + rr) * K + e % K] : 0.f;
    }
    for (int e = threadIdx.x; e < 64 * T; e += 256) {
      const int rr = e / T;
      sR[rr * kMaxT + e % T] = rr < nr ? r[(c0 + rr) * T + e % T] : 0.f;
    }
    __syncthreads();
    if (kk < K && t < T)
      for (int rr = 0; rr < nr; ++rr) acc = __builtin_fmaf(sL[rr * kMaxK + kk], sR[rr * kMaxT + t], acc);
  }
  if (kk < K && t < T) partial_w[((size_t)blockIdx.x * K + kk) * T + t] = acc;
}

// z = M^-1 r with M = L L^T + sigma2 I (Woodbury, Cinv = (sigma2 I + L^T L)^-1), partial_rz = sum r.z
// K == 0: identity preconditioner (z = r)
__global__ __launch_bounds__(256) void k_precond(const float *__restrict__ L, const double *__restrict__ Cinv,
                                                 const float *__restrict__ partial_w, int nparts_w,
                                                 const float *__restrict__ r, float *__restrict__ z,
                                                 float *__restrict__ partial_rz, long long N, int T, int K,
                                                 float sigma2) {
  // the capacitance system (sigma2 I + L^T L) has a condition number ~ |K| / sigma2 (1e5 at N = 391k): its k x k
  // solve is carried in float64 (as the torch path does), only the N-sized vector work is float32
  __shared__ double sw[kMaxK * kMaxT];    // w = L^T r
  __shared__ float stv[kMaxK * kMaxT];    // Cinv w
  __shared__ float sh[4];
  if (K > 0) {
    for (int e = threadIdx.x; e < K * T; e += 256) {
      double s = 0.0;
      for (int p = 0; p < nparts_w; ++p) s += (double)partial_w[(size_t)p * K * T + e];
      sw[(e / T) * kMaxT + e % T] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < K * T; e += 256) {
      const int a = e / T, t = e % T;
      double s = 0.0;
      for (int b = 0; b < K; ++b) s += Cinv[a * K + b] * sw[b * kMaxT + t];
      stv[a * kMaxT + t] = (float)s;
    }
    __syncthreads();
  }
  float acc[kMaxT];
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) acc[t] = 0.f;
  const float inv_s = K > 0 ? 1.0f / sigma2 : 1.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
    float lrow[kMaxK];
#pragma unroll
    for (int b = 0; b < kMaxK; ++b) lrow[b] = b < K ? L[i * K + b] : 0.f;
#pragma unroll
    for (int t = 0; t < kMaxT; ++t) {
      if (t < T) {
        const float rv = r[i * T + t];
        float corr = 0.f;
#pragma unroll
        for (int b = 0; b < kMaxK; ++b)
          if (b < K) corr = __builtin_fmaf(lrow[b], stv[b * kMaxT + t], corr);
        const float zv = (rv - corr) * inv_s;
        z[i * T + t] = zv;
        acc[t] = __builtin_fmaf(rv, zv, acc[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) {
    if (t < T) {
      const float s = block_sum(acc[t], sh);
      if (threadIdx.x == 0) partial_rz[(size_t)blockIdx.x * T + t] = s;
    }
  }
}

// first direction: p = z, rz = sum partial_rz
__global__ __launch_bounds__(256) void k_first_dir(const float *__restrict__ z, float *__restrict__ p,
                                                   const float *__restrict__ partial_rz, int nparts,
                                                   CgState *__restrict__ st, long long N, int T) {
  if (blockIdx.x == 0 && (int)threadIdx.x < T) {
    float s = 0.f;
    for (int q = 0; q < nparts; ++q) s += partial_rz[(size_t)q * T + threadIdx.x];
    st->rz[0][threadIdx.x] = s;
    st->resid[threadIdx.x] = 1.0f;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N * T; i += (long long)gridDim.x * 256) p[i] = z[i];
}

// alpha = rz / pAp (guarded); x += alpha p; r -= alpha Ap; partial |r|^2
__global__ __launch_bounds__(256) void k_update(const float *__restrict__ p, const float *__restrict__ Ap,
                                                const float *__restrict__ partial_pAp, int nparts,
                                                float *__restrict__ x, float *__restrict__ r,
                                                float *__restrict__ partial_rr, const CgState *__restrict__ st,
                                                float *__restrict__ alpha_out, long long N, int T, float eps,
                                                float stop_after, int cur) {
  __shared__ float salpha[kMaxT];
  __shared__ float sh[4];
  if ((int)threadIdx.x < T) {
    float s = 0.f;
    for (int q = 0; q < nparts; ++q) s += partial_pAp[(size_t)q * T + threadIdx.x];
    float a = (fabsf(s) > eps) ? st->rz[cur][threadIdx.x] / s : 0.f;
    if (st->resid[threadIdx.x] < stop_after || st->rhs_zero[threadIdx.x]) a = 0.f;
    salpha[threadIdx.x] = a;
    if (blockIdx.x == 0) alpha_out[threadIdx.x] = a;
  }
  __syncthreads();
  float acc[kMaxT];
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) acc[t] = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
#pragma unroll
    for (int t = 0; t < kMaxT; ++t) {
      if (t < T) {
        const float a = salpha[t];
        x[i * T + t] = __builtin_fmaf(a, p[i * T + t], x[i * T + t]);
        const float rv = __builtin_fmaf(-a, Ap[i * T + t], r[i * T + t]);
        r[i * T + t] = rv;
        acc[t] = __builtin_fmaf(rv, rv, acc[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) {
    if (t < T) {
      const float s = block_sum(acc[t], sh);
      if (threadIdx.x == 0) partial_rr[(size_t)blockIdx.x * T + t] = s;
    }
  }
}

// beta = rz' / rz; p = z + beta p; bookkeeping (block 0): rz <- rz', resid, mean residual, beta history
__global__ __launch_bounds__(256) void k_direction(const float *__restrict__ z, float *__restrict__ p,
                                                   const float *__restrict__ partial_rz, const float *__restrict__ partial_rr,
                                                   int nparts, CgState *__restrict__ st, float *__restrict__ beta_out,
                                                   long long N, int T, float eps, int cur) {
  __shared__ float sbeta[kMaxT];
  __shared__ float sres[kMaxT];
  if ((int)threadIdx.x < T) {
    float rzn = 0.f, rr = 0.f;
    for (int q = 0; q < nparts; ++q) {
      rzn += partial_rz[(size_t)q * T + threadIdx.x];
      rr += partial_rr[(size_t)q * T + threadIdx.x];
    }
    const float rz = st->rz[cur][threadIdx.x];
    sbeta[threadIdx.x] = (fabsf(rz) > eps) ? rzn / rz : 0.f;
    sres[threadIdx.x] = st->rhs_zero[threadIdx.x] ? 0.f : sqrtf(rr);
  }
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256)
    for (int t = 0; t < T; ++t) p[i * T + t] = __builtin_fmaf(sbeta[t], p[i * T + t], z[i * T + t]);
  if (blockIdx.x == 0) {
    __syncthreads();
    if ((int)threadIdx.x < T) {
      float rzn = 0.f;
      for (int q = 0; q < nparts; ++q) rzn += partial_rz[(size_t)q * T + threadIdx.x];
      st->rz[cur ^ 1][threadIdx.x] = rzn;
      st->resid[threadIdx.x] = sres[threadIdx.x];
      beta_out[threadIdx.x] = sbeta[threadIdx.x];
    }
    if (threadIdx.x == 0) {
      float m = 0.f;
      for (int t = 0; t < T; ++t) m += sres[t];
      st->mean_resid = m / (float)T;
    }
  }
}

__global__ __launch_bounds__(256) void k_unnormalise(float *__restrict__ x, const CgState *__restrict__ st, long long N,
                                                     int T) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256)
    for (int t = 0; t < T; ++t) x[i * T + t] *= st->rhs_norm[t];
}

inline int nblocks_for(long long N) {
  long long b = (N + 1023) / 1024;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

int apply_operator(const rpgp_operator *op, const float *V, float *out, int T, void *ws, size_t ws_bytes, void *stream) {
  switch (op->kind) {
    case RPGP_OP_FUSED:
      return rpgp_mvm_sym(op->Z, V, out, op->N, op->ldz, T, op->j0, op->j1, op->scale, op->noise, ws, ws_bytes, stream);
    case RPGP_OP_FUSED_PREPARED:
      return rpgp_mvm_sym_prepared(op->prep, V, out, op->N, op->J, T, op->j0, op->j1, op->scale, op->noise, ws, ws_bytes,
                                   stream);
    case RPGP_OP_SKI:
      return rpgp_ski_mvm(op->Z, op->Z, op->grid_params, V, out, op->N, op->N, op->ldz, op->ldz, op->J, op->G, T,
                          op->scale, op->noise, ws, ws_bytes, stream);
    case RPGP_OP_DENSE:
      return rpgp_dense_mvm(op->Kd, V, out, op->N, op->ldk, T, op->noise, stream);
    default:
      return RPGP_EINVAL;
  }
}

size_t operator_workspace(const rpgp_operator *op, int T) {
  switch (op->kind) {
    case RPGP_OP_FUSED:
    case RPGP_OP_FUSED_PREPARED:
      return rpgp_mvm_sym_workspace_bytes(op->N, T);
    case RPGP_OP_SKI:
      return rpgp_ski_workspace_bytes(op->J, op->G, T);
    case RPGP_OP_DENSE:
      return 256;
    default:
      return 0;
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

size_t rpgp_mbcg_workspace_bytes(const rpgp_operator *op, int T, int precond_rank) {
  if (!op || T <= 0 || T > kMaxT || precond_rank < 0 || precond_rank > kMaxK) return 0;
  const size_t nt = (size_t)op->N * T * sizeof(float);
  size_t total = 4 * align256(nt);                                             // r, p, z, Ap
  total += align256(sizeof(CgState));
  total += 3 * align256((size_t)kMaxBlocks * kMaxT * sizeof(float));           // partial pAp / rr / rz
  total += align256((size_t)kMaxBlocks * kMaxK * kMaxT * sizeof(float));       // partial L^T r
  total += 2 * align256((size_t)(kMaxHist + 1) * kMaxT * sizeof(float));       // alpha / beta history
  total += align256(operator_workspace(op, T));
  return total;
}

int rpgp_mbcg_solve(const rpgp_operator *op, const float *rhs, float *x, int T, int max_iter, int min_iter,
                    int hist_len, int check_every, float tolerance, int precond_rank, const float *L,
                    const double *Cinv, float precond_sigma2, float *alpha_hist_host, float *beta_hist_host,
                    int *iterations_host, float *mean_resid_host, void *workspace, size_t workspace_bytes, void *stream) {
  if (!op || !rhs || !x || T <= 0 || T > kMaxT || max_iter < 0 || hist_len < 0 || check_every <= 0 ||
      precond_rank < 0 || precond_rank > kMaxK || (precond_rank > 0 && (!L || !Cinv)) || op->N <= 0)
    return RPGP_EINVAL;
  if (hist_len > kMaxHist || (hist_len > 0 && (!alpha_hist_host || !beta_hist_host))) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_mbcg_workspace_bytes(op, T, precond_rank)) return RPGP_EWORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long N = op->N;
  const int K = precond_rank;
  const size_t nt = (size_t)N * T * sizeof(float);

  char *w = reinterpret_cast<char *>(workspace);
  float *r = reinterpret_cast<float *>(w); w += align256(nt);
  float *p = reinterpret_cast<float *>(w); w += align256(nt);
  float *z = reinterpret_cast<float *>(w); w += align256(nt);
  float *Ap = reinterpret_cast<float *>(w); w += align256(nt);
  CgState *state = reinterpret_cast<CgState *>(w); w += align256(sizeof(CgState));
  float *part_a = reinterpret_cast<float *>(w); w += align256((size_t)kMaxBlocks * kMaxT * sizeof(float));
  float *part_rr = reinterpret_cast<float *>(w); w += align256((size_t)kMaxBlocks * kMaxT * sizeof(float));
  float *part_rz = reinterpret_cast<float *>(w); w += align256((size_t)kMaxBlocks * kMaxT * sizeof(float));
  float *part_w = reinterpret_cast<float *>(w); w += align256((size_t)kMaxBlocks * kMaxK * kMaxT * sizeof(float));
  float *alpha_d = reinterpret_cast<float *>(w); w += align256((size_t)(kMaxHist + 1) * kMaxT * sizeof(float));
  float *beta_d = reinterpret_cast<float *>(w); w += align256((size_t)(kMaxHist + 1) * kMaxT * sizeof(float));
  void *op_ws = w;
  const size_t op_ws_bytes = operator_workspace(op, T);

  const int nb = nblocks_for(N);
  const long long rows_per_block = (N + nb - 1) / nb;
  const float eps = 1e-30f, stop_after = 1e-10f;

  // normalise right-hand sides, x = 0
  hipLaunchKernelGGL(k_coldot, dim3(nb), dim3(256), 0, st, rhs, rhs, part_a, N, T);
  hipLaunchKernelGGL(k_normalise, dim3(nb), dim3(256), 0, st, rhs, part_a, nb, r, x, state, N, T);
  // z0 = M^-1 r0, p0 = z0, rz0
  if (K > 0) hipLaunchKernelGGL(k_Ltr, dim3(nb), dim3(256), 0, st, L, r, part_w, N, T, K, rows_per_block);
  hipLaunchKernelGGL(k_precond, dim3(nb), dim3(256), 0, st, L, Cinv, part_w, nb, r, z, part_rz, N, T, K, precond_sigma2);
  hipLaunchKernelGGL(k_first_dir, dim3(nb), dim3(256), 0, st, z, p, part_rz, nb, state, N, T);
  CG_CHECK(hipGetLastError());

  int it = 0;
  float mean_resid = 1.0f;
  bool converged = false;
  const int n_iter = max_iter < N ? max_iter : (int)N;
  // device-side history ring: alpha/beta of iteration k are copied to the host arrays asynchronously
  for (it = 0; it < n_iter; ++it) {
    int rc = apply_operator(op, p, Ap, T, op_ws, op_ws_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_coldot, dim3(nb), dim3(256), 0, st, p, Ap, part_a, N, T);
    const int slot = it < hist_len ? it : kMaxHist;      // history row (the last row is a scratch slot)
    hipLaunchKernelGGL(k_update, dim3(nb), dim3(256), 0, st, p, Ap, part_a, nb, x, r, part_rr, state,
                       alpha_d + (size_t)slot * kMaxT, N, T, eps, stop_after, it & 1);
    if (K > 0) hipLaunchKernelGGL(k_Ltr, dim3(nb), dim3(256), 0, st, L, r, part_w, N, T, K, rows_per_block);
    hipLaunchKernelGGL(k_precond, dim3(nb), dim3(256), 0, st, L, Cinv, part_w, nb, r, z, part_rz, N, T, K,
                       precond_sigma2);
    hipLaunchKernelGGL(k_direction, dim3(nb), dim3(256), 0, st, z, p, part_rz, part_rr, nb, state,
                       beta_d + (size_t)slot * kMaxT, N, T, eps, it & 1);
    const bool hist_pending = it < (hist_len < n_iter ? hist_len : n_iter) - 1;
    const int min_it = min_iter < n_iter - 1 ? min_iter : n_iter - 1;
    if (it >= min_it && !hist_pending && ((it - min_it) % check_every == 0 || it == n_iter - 1)) {
      CG_CHECK(hipMemcpyAsync(&mean_resid, &state->mean_resid, sizeof(float), hipMemcpyDeviceToHost, st));
      CG_CHECK(hipStreamSynchronize(st));
      if (mean_resid != mean_resid) {
        if (iterations_host) *iterations_host = it + 1;
        if (mean_resid_host) *mean_resid_host = mean_resid;
        return RPGP_ENUMERIC;
      }
      if (mean_resid < tolerance) {
        converged = true;
        ++it;
        break;
      }
    }
  }
  hipLaunchKernelGGL(k_unnormalise, dim3(nb), dim3(256), 0, st, x, state, N, T);
  const int nh = it < hist_len ? it : hist_len;
  if (nh > 0) {   // history rows are kMaxT wide on the device, [hist_len][kMaxT] on the host
    CG_CHECK(hipMemcpyAsync(alpha_hist_host, alpha_d, (size_t)nh * kMaxT * sizeof(float), hipMemcpyDeviceToHost, st));
    CG_CHECK(hipMemcpyAsync(beta_hist_host, beta_d, (size_t)nh * kMaxT * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  CG_CHECK(hipMemcpyAsync(&mean_resid, &state->mean_resid, sizeof(float), hipMemcpyDeviceToHost, st));
  CG_CHECK(hipStreamSynchronize(st));
  if (iterations_host) *iterations_host = it;
  if (mean_resid_host) *mean_resid_host = mean_resid;
  (void)converged;
  return 0;
}

}  // extern "C"
