// rpgp_step.hip — the small kernels AROUND the solve of one optimiser step of the flagship model.
//
// One step of `-mll(model(X), y); loss.backward(); optimizer.step()` (fitting/optimizing.py:65-76) on the model every served
// `additive_rp` specification builds (training_routines.py:131-189, 325-410; models.py:10-20) is, outside the CG solve and the
// derivative sweep, a chain of ~90 one-to-four-microsecond element-wise launches on d + 3 scalars and a handful of N x 11 blocks:
// softplus and its derivative, the lengthscale division, the probe draw z = L e1 + sigma e2 and its normalisation, the
// concatenations that lay out [probes | y - c] and the two sides of the bilinear derivative, the reductions r.alpha,
// sum(L * R), sum(alpha), the chain rule back to the raw parameters.  The device needs ~0.2 ms for them, the host ~0.9 ms to
// issue them (profiles/r4_step_C2_step_gaps.txt): the step is HOST-bound there.  Each entry point below is one launch for one such stretch; the arithmetic is the one the torch operations perform (fp32, the same formulas), the
// reductions are fixed-order (bitwise reproducible).  rpgp_amd/fused_mll.py is the caller.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/rpgp.h"
#include "rpgp_internal.h"

namespace {

#define STEP_CHECK(expr)                         \
  do {                                           \
    hipError_t _e = (expr);                      \
    if (_e != hipSuccess) return (int)_e;        \
  } while (0)

// F.softplus(x) (beta 1, threshold 20) and its derivative
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// pinned landing zone of the three scalars the C-ABI kernels take by value (one per host thread)
struct HyperHost {
  float *host = nullptr, *dev = nullptr;       // [0] outputscale, [1] noise, [2] mean, [3] stamp
  bool ok = false;
  int init() {
    if (ok) return 0;
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&host), 64, hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) return (int)e;
    e = hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), host, 0);
    if (e != hipSuccess) return (int)e;
    ok = true;
    return 0;
  }
};
thread_local HyperHost g_hyper;

// pinned ring of the objective's posted values (rpgp_step_value with a ticket): slot = {value, stamp}.  The training loop reads
// the loss of EVERY step on the host (fitting/optimizing.py:76 `loss.item()`): a device-to-host copy there waits for the whole
// step — derivative and optimiser update included — although the value exists since the end of the forward pass.  The kernel
// that forms it posts it here as well, and the host picks it up when it wants it without touching the stream.
constexpr int kValueSlots = 16;
struct ValueHost {
  float *host = nullptr, *dev = nullptr;       // kValueSlots x {value, stamp}
  float issued[kValueSlots] = {0.f};           // the stamp of the latest ticket of each slot
  int next = 0;
  float counter = 0.f;
  bool ok = false;
  int init() {
    if (ok) return 0;
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&host), kValueSlots * 2 * sizeof(float),
                                 hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) return (int)e;
    for (int i = 0; i < 2 * kValueSlots; ++i) host[i] = 0.f;
    e = hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), host, 0);
    if (e != hipSuccess) return (int)e;
    ok = true;
    return 0;
  }
};
thread_local ValueHost g_value;

// ---- hyper-parameters -> what the step needs ---------------------------------------------------------------------------------
// dev_out: [0] outputscale  [1] noise  [2] mean  [3] sigmoid(raw_os)  [4] sigmoid(raw_noise)  [8 ..) ls[n_ls]  then sigmoid(raw_ls)[n_ls]
// Peff[i][j] = W[j][i] / ls[i] (prescale, n_ls = d) | / ls[j] (n_ls = J) | / ls[0] (n_ls = 1)       (W: the Linear weight, J x d)
__global__ __launch_bounds__(256) void k_step_hyper(const float *__restrict__ raw_ls, int n_ls, const float *__restrict__ raw_os,
                                                    const float *__restrict__ raw_noise, const float *__restrict__ mean,
                                                    const float *__restrict__ W, int d, int J, int prescale, float min_noise,
                                                    float *__restrict__ Peff, float *__restrict__ dev_out,
                                                    float *__restrict__ host_out, float stamp) {
  __shared__ float sls[1024];
  for (int i = threadIdx.x; i < n_ls; i += 256) {
    const float x = raw_ls[i];
    const float l = softplus_f(x);
    sls[i] = l;
    dev_out[8 + i] = l;
    dev_out[8 + n_ls + i] = sigmoid_f(x);
  }
  if (threadIdx.x == 0) {
    const float os = softplus_f(raw_os[0]), nz = softplus_f(raw_noise[0]) + min_noise, mu = mean[0];
    dev_out[0] = os;
    dev_out[1] = nz;
    dev_out[2] = mu;
    dev_out[3] = sigmoid_f(raw_os[0]);
    dev_out[4] = sigmoid_f(raw_noise[0]);
    host_out[0] = os;
    host_out[1] = nz;
    host_out[2] = mu;
    __threadfence_system();
    *reinterpret_cast<volatile float *>(host_out + 3) = stamp;
    __threadfence_system();
  }
  __syncthreads();
  for (int e = threadIdx.x; e < d * J; e += 256) {
    const int i = e / J, j = e - i * J;
    const float l = n_ls == 1 ? sls[0] : (prescale ? sls[i] : sls[j]);
    Peff[e] = W[(size_t)j * d + i] / l;
  }
}

// ---- probes z = L e1 + sqrt(sigma^2) e2 (WoodburyPreconditioner.sample) and the solve's right-hand sides [z | y - c] ------------
// One ROW per thread (a wave reads 64 consecutive rows of L and e2: whole cache lines), the p <= 16 results in registers, written
// straight into the N x (p + 1) block with y - c in the last column.  The probes are NOT normalised here: the executor
// normalises every right-hand-side column itself (k_init) and hands back Khat^-1 of the columns as given — exactly the
// Khat^-1 z_p the derivative needs (the generic path divides by |z_p| first and multiplies the solves by it afterwards).
constexpr int kMaxP = 16, kMaxKp = 64;
// (round 6) The tiles of L (256 x k), e2 (256 x p) and the result (256 x (p + 1)) go through LDS as FLAT arrays: one row per
// thread straight from memory makes every load instruction of a wave 64 requests 4 k bytes apart (C5, N = 391k: 37 us for 56 MB).
__global__ __launch_bounds__(256) void k_step_probes(const float *__restrict__ L, int k, const float *__restrict__ e1,
                                                     const float *__restrict__ e2, float sqrt_noise,
                                                     const float *__restrict__ y, const float *__restrict__ mean_dev, long long N,
                                                     int p, float *__restrict__ full_rhs) {
  extern __shared__ float smem[];
  float *se1 = smem;                                    // [k][16]: e1 padded to 16 columns
  float *sL = se1 + kMaxKp * kMaxP;                     // [256][k | 1]   (odd row stride: conflict-free row reads)
  const int ldl = k | 1;
  float *sE = sL + 256 * ldl;                           // [256][p | 1], later the result rows [256][(p + 1) | 1]
  const int lde = p | 1, T = p + 1, ldo = T | 1;
  for (int e = threadIdx.x; e < k * kMaxP; e += 256) {
    const int kk = e / kMaxP, c = e - kk * kMaxP;
    se1[e] = c < p ? e1[kk * p + c] : 0.f;
  }
  const float mu = mean_dev[0];
  for (long long r0 = (long long)blockIdx.x * 256; r0 < N; r0 += (long long)gridDim.x * 256) {
    const int rows = (int)((N - r0 < 256) ? N - r0 : 256);
    __syncthreads();
    for (int e = threadIdx.x; e < rows * k; e += 256) {
      const int r = e / k, c = e - r * k;
      sL[r * ldl + c] = L[r0 * k + e];
    }
    for (int e = threadIdx.x; e < rows * p; e += 256) {
      const int r = e / p, c = e - r * p;
      sE[r * lde + c] = e2[r0 * p + e];
    }
    const float yv = (int)threadIdx.x < rows ? y[r0 + threadIdx.x] : 0.f;
    __syncthreads();
    float acc[kMaxP], ev[kMaxP];
#pragma unroll
    for (int c = 0; c < kMaxP; ++c) {
      acc[c] = 0.f;
      ev[c] = c < p ? sE[threadIdx.x * lde + c] : 0.f;
    }
    for (int kk = 0; kk < k; ++kk) {
      const float l = sL[threadIdx.x * ldl + kk];
#pragma unroll
      for (int c = 0; c < kMaxP; ++c) acc[c] = __builtin_fmaf(l, se1[kk * kMaxP + c], acc[c]);
    }
    __syncthreads();                                     // (sE is reused for the result rows)
#pragma unroll
    for (int c = 0; c < kMaxP; ++c)
      if (c < p) sE[threadIdx.x * ldo + c] = acc[c] + sqrt_noise * ev[c];
    sE[threadIdx.x * ldo + p] = yv - mu;
    __syncthreads();
    for (int e = threadIdx.x; e < rows * T; e += 256) {
      const int r = e / T, c = e - r * T;
      full_rhs[r0 * T + e] = sE[r * ldo + c];
    }
  }
}

// ---- value of the objective ------------------------------------------------------------------------------------------------
// inv_quad = sum_i rhs[i][col] * sol[i][col];  out[0] = (inv_quad + logdet) * c1 + c2,  out[1] = inv_quad.  Up to kValueBlocks
// workgroups write a float64 partial each; the LAST one to finish adds them in index order (bitwise reproducible whoever it is)
// — once per optimiser step, so the hand-off (every wave drains its stores, workgroup barrier, agent-scope release, arrive on
// the counter; the last arriver: agent-scope acquire, plain loads — MI355X_MICROARCH.md "valid forms") costs nothing that
// matters.  ws: [0] arrival counter (0 on entry, 0 again on exit), doubles from byte 256.
constexpr int kValueBlocks = 1024;
__device__ __forceinline__ void value_out(double tot, double logdet, double c1, double c2, float *__restrict__ out,
                                          float *__restrict__ host_slot, float stamp) {
  const float val = (float)((tot + logdet) * c1 + c2);
  out[0] = val;
  out[1] = (float)tot;
  if (host_slot) {
    // post {value, stamp} to the host as ONE 8-byte store to the mapped pinned slot: nothing to order, no system-scope fence
    const unsigned long long both = (unsigned long long)__float_as_uint(val) | ((unsigned long long)__float_as_uint(stamp) << 32);
    *reinterpret_cast<volatile unsigned long long *>(host_slot) = both;
  }
}
__global__ __launch_bounds__(256) void k_step_value(const float *__restrict__ rhs, const float *__restrict__ sol, long long N,
                                                    int T, int col, double logdet, double c1, double c2,
                                                    float *__restrict__ out, unsigned *__restrict__ counter,
                                                    double *__restrict__ part, float *__restrict__ host_slot, float stamp) {
  __shared__ double sh[256];
  double s = 0.0;
  // (four rows of a thread in flight: the column of an N x T block is one 4-byte request per row, and a thread walking its
  //  rows one round trip at a time made this 30 us at N = 391k)
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < N; i += 4 * stride) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = rhs[(i + u * stride) * T + col];
      b[u] = sol[(i + u * stride) * T + col];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s += (double)a[u] * (double)b[u];
  }
  for (; i < N; i += stride) s += (double)rhs[i * T + col] * (double)sol[i * T + col];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[blockIdx.x] = sh[0];
    if (counter == nullptr) return;            // two-launch form: k_step_value_finish adds the partials up
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == gridDim.x - 1;
    if (last) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      double tot = 0.0;
      for (unsigned q = 0; q < gridDim.x; ++q) tot += part[q];
      value_out(tot, logdet, c1, c2, out, host_slot, stamp);
    }
  }
}

// Many workgroups (N beyond ~64k): the partials are added by a second launch instead.  An agent-scope release per workgroup is a
// write-back of what the solve left dirty in that XCD's L2, and with hundreds of workgroups arriving the single-launch form
// spent 30 - 44 us in them at N = 391k (6 us at N = 7k, where it stays).
__global__ __launch_bounds__(256) void k_step_value_finish(const double *__restrict__ part, int nparts, double logdet, double c1,
                                                           double c2, float *__restrict__ out, float *__restrict__ host_slot,
                                                           float stamp) {
  __shared__ double sh[256];
  // (index order within a thread, then the tree: a fixed order)
  double s = 0.0;
  for (int q = threadIdx.x; q < nparts; q += 256) s += part[q];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) value_out(sh[0], logdet, c1, c2, out, host_slot, stamp);
}

// ---- the two sides of the bilinear derivative ---------------------------------------------------------------------------------
// left = [Khat^-1 z_p * (g_ld / p) | -g_iq alpha],  right = [M^-1 z_p | alpha]   with Khat^-1 z_p = sol[:, c];
// g_iq = g_ld = g[0] * gscale (the incoming gradient is a device scalar).  Per block: partial sum(left * right), sum(alpha).
constexpr int kLrBlocks = 4096;
__global__ __launch_bounds__(256) void k_step_lr(const float *__restrict__ sol, const float *__restrict__ pre_probes,
                                                 long long ldp, const float *__restrict__ g, float gscale, long long N, int p,
                                                 float *__restrict__ left, float *__restrict__ right,
                                                 float *__restrict__ part) {
  __shared__ float s1[256], s2[256];
  const float gq = g[0] * gscale;
  const float gp = gq / (float)p;
  const int T = p + 1;
  const long long total = N * T;
  float a1 = 0.f, a2 = 0.f;
  // (four elements of a thread in flight, up to 4 096 workgroups: 33 dependent round trips per thread were 30 us at N = 391k)
  const long long stride = (long long)gridDim.x * 256;
  for (long long e0 = (long long)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += 4 * stride) {
    float sv[4], pv[4];
    int cc[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long e = e0 + u * stride;
      ok[u] = e < total;
      const long long ec = ok[u] ? e : total - 1;
      const long long row = ec / T;
      cc[u] = (int)(ec - row * T);
      sv[u] = sol[ec];
      pv[u] = pre_probes[row * ldp + (cc[u] < p ? cc[u] : 0)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      const long long e = e0 + u * stride;
      float l, r;
      if (cc[u] < p) {
        l = sv[u] * gp;
        r = pv[u];
      } else {
        l = -gq * sv[u];
        r = sv[u];
        a2 += sv[u];
      }
      left[e] = l;
      right[e] = r;
      a1 = __builtin_fmaf(l, r, a1);
    }
  }
  s1[threadIdx.x] = a1;
  s2[threadIdx.x] = a2;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      s1[threadIdx.x] += s1[threadIdx.x + w];
      s2[threadIdx.x] += s2[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = s1[0];
    part[2 * blockIdx.x + 1] = s2[0];
  }
}

// ---- chain rule back to the raw parameters -------------------------------------------------------------------------------------
// dPeff: d x J (gradient w.r.t. Peff = P / l);  g_l = -sum(dPeff * P) / l^2 over the row (prescale) / column / everything;
// g_raw_ls = g_l sigmoid(raw_ls);  g_raw_os = gs sigmoid(raw_os);  g_raw_noise = (gn + g[0] dlp_over_n) sigmoid(raw_noise) with
// gn = sum of the partials;  g_mean = -2 g_iq sum(alpha).   hyp = the dev_out block of k_step_hyper.
__global__ __launch_bounds__(256) void k_step_hyper_backward(const float *__restrict__ dPeff, const float *__restrict__ W, int d,
                                                             int J, int n_ls, int prescale, float zfac,
                                                             const float *__restrict__ hyp, const float *__restrict__ gs,
                                                             const float *__restrict__ part, int nparts,
                                                             const float *__restrict__ g, float gscale, float dlp_over_n,
                                                             float gs_scale, float *__restrict__ g_raw_ls,
                                                             float *__restrict__ g_raw_os,
                                                             float *__restrict__ g_raw_noise, float *__restrict__ g_mean) {
  __shared__ double sa[256], sb[256];
  const float *ls = hyp + 8, *sig_ls = hyp + 8 + n_ls;
  if (n_ls > 1) {
    for (int i = threadIdx.x; i < n_ls; i += 256) {
      float s = 0.f;
      if (prescale) {
        for (int j = 0; j < J; ++j) s = __builtin_fmaf(dPeff[(size_t)i * J + j], W[(size_t)j * d + i], s);
      } else {
        for (int r = 0; r < d; ++r) s = __builtin_fmaf(dPeff[(size_t)r * J + i], W[(size_t)i * d + r], s);
      }
      g_raw_ls[i] = -(s * zfac) / (ls[i] * ls[i]) * sig_ls[i];
    }
  } else if (threadIdx.x == 0) {
    float s = 0.f;
    for (int e = 0; e < d * J; ++e) {
      const int i = e / J, j = e - i * J;
      s = __builtin_fmaf(dPeff[e], W[(size_t)j * d + i], s);
    }
    g_raw_ls[0] = -(s * zfac) / (ls[0] * ls[0]) * sig_ls[0];
  }
  // the two sums over the workgroups' partials of k_step_lr: thread t adds entries t, t + 256, ..., then a tree — fixed order
  // (one thread walking ~600 partials was 25 us at N = 7 372: longer than every other kernel of the backward pass but one)
  double a = 0.0, b = 0.0;
  for (int q = threadIdx.x; q < nparts; q += 256) {
    a += (double)part[2 * q];
    b += (double)part[2 * q + 1];
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      sa[threadIdx.x] += sa[threadIdx.x + w];
      sb[threadIdx.x] += sb[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float gq = g[0] * gscale;
    g_raw_os[0] = gs[0] * gs_scale * hyp[3];
    g_raw_noise[0] = ((float)sa[0] + g[0] * dlp_over_n) * hyp[4];
    g_mean[0] = -(2.f * gq * (float)sb[0]);
  }
}

// (wall-time bounded: the single-workgroup kernel behind this stamp finishes within microseconds of its launch unless the
//  queue in front of it is long — then the runtime's synchronisation takes over, rpgp_internal::spin_until)
inline bool spin_stamp(const volatile float *p, float stamp) {
  return rpgp_internal::spin_until([p, stamp] { return *p == stamp; }, 5000);
}

}  // namespace

extern "C" {

int rpgp_step_hyper(const float *raw_ls, int n_ls, const float *raw_os, const float *raw_noise, const float *mean,
                    const float *W, int d, int J, int prescale, float min_noise, float *Peff, float *dev_out,
                    float *outputscale_host, float *noise_host, float *mean_host, void *stream) {
  if (!raw_ls || !raw_os || !raw_noise || !mean || !W || !Peff || !dev_out || n_ls < 1 || n_ls > 1024 || d < 1 || J < 1 ||
      (n_ls != 1 && n_ls != (prescale ? d : J)))
    return RPGP_EINVAL;
  const int rc = g_hyper.init();
  if (rc) return rc;
  static thread_local float counter = 0.f;
  counter = counter >= 1.0e6f ? 1.f : counter + 1.f;           // a stamp no earlier call left behind
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_step_hyper, dim3(1), dim3(256), 0, st, raw_ls, n_ls, raw_os, raw_noise, mean, W, d, J, prescale, min_noise,
                     Peff, dev_out, g_hyper.dev, counter);
  STEP_CHECK(hipGetLastError());
  if (!spin_stamp(g_hyper.host + 3, counter)) STEP_CHECK(hipStreamSynchronize(st));
  if (outputscale_host) *outputscale_host = g_hyper.host[0];
  if (noise_host) *noise_host = g_hyper.host[1];
  if (mean_host) *mean_host = g_hyper.host[2];
  return 0;
}

int rpgp_step_probes(const float *L, int k, const float *e1, const float *e2, float sqrt_noise, const float *y,
                     const float *mean_dev, int64_t N, int p, float *full_rhs, void *stream) {
  if (!L || !e1 || !e2 || !y || !mean_dev || !full_rhs || N < 1 || p < 1 || p > kMaxP || k < 1 || k > kMaxKp) return RPGP_EINVAL;
  long long nb = (N + 255) / 256;
  if (nb > 2048) nb = 2048;
  const size_t lds = ((size_t)kMaxKp * kMaxP + 256 * (size_t)(k | 1) + 256 * (size_t)((p + 1) | 1)) * sizeof(float);
  if (lds > 48 * 1024) {      // (rank-64 factors: 88 KB; the rank-15 preconditioner of a 10-probe block needs 30 KB)
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_step_probes),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k_step_probes, dim3((unsigned)nb), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), L, k, e1, e2,
                     sqrt_noise, y, mean_dev, (long long)N, p, full_rhs);
  return (int)hipGetLastError();
}

size_t rpgp_step_value_workspace_bytes(void) { return 256 + (size_t)kValueBlocks * sizeof(double); }

int rpgp_step_value(const float *full_rhs, const float *solves, int64_t N, int T, int col, double logdet, double c1, double c2,
                    float *out2, void *workspace, size_t workspace_bytes, int *ticket_out, void *stream) {
  if (!full_rhs || !solves || !out2 || N < 1 || T < 1 || col < 0 || col >= T) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_step_value_workspace_bytes()) return RPGP_EWORKSPACE;
  long long nb = (N + 1023) / 1024;                  // ~4 rows per thread (one batch of loads)
  if (nb > kValueBlocks) nb = kValueBlocks;
  float *slot = nullptr;
  float stamp = 0.f;
  if (ticket_out) {
    *ticket_out = -1;
    if (g_value.init() == 0) {
      const int sl = g_value.next;
      g_value.next = (sl + 1) % kValueSlots;
      g_value.counter = g_value.counter >= 1.0e6f ? 1.f : g_value.counter + 1.f;
      stamp = g_value.counter;
      g_value.issued[sl] = stamp;
      slot = g_value.dev + 2 * sl;
      *ticket_out = sl + kValueSlots * (int)stamp;
    }
  }
  double *part = reinterpret_cast<double *>(reinterpret_cast<char *>(workspace) + 256);
  const bool two = nb > 64;
  hipLaunchKernelGGL(k_step_value, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), full_rhs, solves,
                     (long long)N, T, col, logdet, c1, c2, out2, two ? nullptr : reinterpret_cast<unsigned *>(workspace), part,
                     slot, stamp);
  if (two)
    hipLaunchKernelGGL(k_step_value_finish, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), part, (int)nb, logdet,
                       c1, c2, out2, slot, stamp);
  return (int)hipGetLastError();
}

int rpgp_step_value_wait(int ticket, float *value_host) {
  if (ticket < 0 || !value_host || !g_value.ok) return RPGP_EINVAL;
  const int sl = ticket % kValueSlots;
  const float stamp = (float)(ticket / kValueSlots);
  if (g_value.issued[sl] != stamp) return RPGP_EINVAL;        // the slot has been handed to a later call (or another thread's)
  // (one aligned 8-byte load: value and stamp of the same store)
  const volatile unsigned long long *p = reinterpret_cast<const volatile unsigned long long *>(g_value.host + 2 * sl);
  unsigned stamp_bits;
  static_assert(sizeof(float) == sizeof(unsigned), "float bits");
  __builtin_memcpy(&stamp_bits, &stamp, 4);
  unsigned long long got = 0;
  if (!rpgp_internal::spin_until([p, stamp_bits, &got] { got = *p; return (unsigned)(got >> 32) == stamp_bits; }, 2000000))
    return RPGP_EINVAL;
  const unsigned vb = (unsigned)(got & 0xffffffffull);
  __builtin_memcpy(value_host, &vb, 4);
  return 0;
}

size_t rpgp_step_lr_workspace_bytes(void) { return (size_t)kLrBlocks * 2 * sizeof(float); }

int rpgp_step_lr(const float *solves, const float *pre_probes, int64_t ldp, const float *g, float gscale, int64_t N, int p,
                 float *left, float *right, float *partials, int *nparts_out, void *stream) {
  if (!solves || !pre_probes || !g || !left || !right || !partials || !nparts_out || N < 1 || p < 1 || p > kMaxP || ldp < p)
    return RPGP_EINVAL;
  long long nb = (N * (p + 1) + 255) / 256;
  if (nb > kLrBlocks) nb = kLrBlocks;
  hipLaunchKernelGGL(k_step_lr, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), solves, pre_probes,
                     (long long)ldp, g, gscale, (long long)N, p, left, right, partials);
  *nparts_out = (int)nb;
  return (int)hipGetLastError();
}

int rpgp_step_hyper_backward(const float *dPeff, const float *W, int d, int J, int n_ls, int prescale, float zfac,
                             const float *hyper_dev, const float *gs, const float *partials, int nparts, const float *g,
                             float gscale, float dlp_over_n, float gs_scale, float *g_raw_ls, float *g_raw_os,
                             float *g_raw_noise, float *g_mean, void *stream) {
  if (!dPeff || !W || !hyper_dev || !gs || !partials || !g || !g_raw_ls || !g_raw_os || !g_raw_noise || !g_mean || d < 1 ||
      J < 1 || n_ls < 1 || nparts < 1 || (n_ls != 1 && n_ls != (prescale ? d : J)))
    return RPGP_EINVAL;
  hipLaunchKernelGGL(k_step_hyper_backward, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dPeff, W, d, J, n_ls,
                     prescale, zfac, hyper_dev, gs, partials, nparts, g, gscale, dlp_over_n, gs_scale, g_raw_ls, g_raw_os,
                     g_raw_noise, g_mean);
  return (int)hipGetLastError();
}

}  // extern "C"
