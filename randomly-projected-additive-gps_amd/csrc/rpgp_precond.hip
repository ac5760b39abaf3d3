// Host-stack pieces of the Woodbury preconditioner M = L L^T + sigma^2 I that run OUTSIDE the
// native mBCG executor: the float64 capacitance Gram L^T L, L^T R for a block of right-hand sides, and the cancelling
// update (R - L t) / sigma^2.  Round 3 profile of one optimiser step at the C2 shape (N = 7372): two library float64 GEMMs
// for the 15 x 15 / 15 x 10 products took 200 us EACH (one 64 x 64 tile, serial loop over N) out of a 4 ms step, behind a
// float32 -> float64 copy of L; the update was an addmm + div + cast.
//
//   rpgp_gram_f64       out[K x T] = A^T B, A: N x K fp32, B: N x T fp32; products are exact in float64 and the sums are
//                       float64 (v_mfma_f64_16x16x4_f64), fixed order -> bit-reproducible.  K, T <= 64.
//   rpgp_woodbury_apply out = (float)(((double)R - L t) / sigma^2), t: K x T float64 — the subtraction that shrinks the
//                       range-of-L component of R by ~1e-6 stays float64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>

#include "../../include/rpgp.h"
#include "rpgp_internal.h"

namespace {

typedef double doublex4p __attribute__((ext_vector_type(4)));

constexpr int kGramMaxWg = 512;

// Lane l of a wave: c = l % 16 (column inside a 16-wide tile), q = l / 16 (row inside a 4-row step).  One matrix
// instruction per (A tile, B tile, 4 rows): A operand [i = c][k = q] = A[n + q][16 ma + c], B operand [k = q][j = c] =
// B[n + q][16 nb + c]; result register r of lane l = out[16 ma + q + 4 r][16 nb + c].
// `counter` != nullptr (few slabs: N <= 16k): the LAST workgroup to finish adds the slabs up itself — gram_finish_kernel's sums
// in gram_finish_kernel's order (four quarters of the slabs, each in slab order, combined in order: bitwise the same result) —
// instead of a second launch; three Gram products per optimiser step are on the host-bound stretches around the solve
// (DESIGN §3.4).  Hand-off as in k_step_value (rpgp_step.hip): stores drained, workgroup barrier, agent-scope release, arrive;
// the last arriver acquires and reads.  The counter returns to zero.
// NW = waves per workgroup: 4, or 16 for tall inputs — the slab count (= workgroups) is capped at 512, so at N = 391k four waves
// per workgroup leave 8 waves per CU to hide ~12 dependent load round trips each (37 us for 39 MB); sixteen fill the CU.
template <int MA, int NB, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gram_partial_kernel(const float *__restrict__ A, long long lda,
                                                           const float *__restrict__ B, long long ldb, long long N, int K,
                                                           int T, double *__restrict__ part, unsigned *__restrict__ counter,
                                                           double *__restrict__ out64, float *__restrict__ out32) {
  __shared__ double sred[NW][64];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  // kU accumulator sets (steps s, s + nwaves, ...): consecutive matrix instructions never chain on one accumulator, and
  // the loads of kU steps are in flight together
  constexpr int kU = (MA * NB <= 2) ? 4 : 2;
  doublex4p acc[kU][MA][NB];
#pragma unroll
  for (int u = 0; u < kU; ++u)
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[u][a][b] = doublex4p{0.0, 0.0, 0.0, 0.0};
  const long long nwaves = (long long)gridDim.x * NW;
  const long long w = (long long)blockIdx.x * NW + wave;
  const long long nsteps = (N + 3) / 4;
  for (long long s = w; s < nsteps; s += kU * nwaves) {
    float af[kU][MA], bf[kU][NB];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      // clamped row + select: a load under a condition compiles to a branch with its own wait
      const long long n = 4 * (s + u * nwaves) + q;
      const bool rv = n < N;
      const long long nc = rv ? n : N - 1;
#pragma unroll
      for (int a = 0; a < MA; ++a) {
        const int col = 16 * a + c;
        const float v = A[nc * lda + (col < K ? col : K - 1)];
        af[u][a] = (rv && col < K) ? v : 0.f;
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int col = 16 * b + c;
        const float v = B[nc * ldb + (col < T ? col : T - 1)];
        bf[u][b] = (rv && col < T) ? v : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u)
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[u][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)af[u][a], (double)bf[u][b], acc[u][a][b], 0, 0, 0);
  }
  // workgroup sum (waves in order), then one slab per workgroup: part[blockIdx][tile][r][lane]
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      doublex4p v = acc[0][a][b] + acc[1][a][b];
      if constexpr (kU == 4) v = v + (acc[2][a][b] + acc[3][a][b]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        __syncthreads();
        sred[wave][lane] = v[r];
        __syncthreads();
        if (wave == 0) {
          double sum = ((sred[0][lane] + sred[1][lane]) + sred[2][lane]) + sred[3][lane];
          if constexpr (NW == 16) {
#pragma unroll
            for (int g4 = 4; g4 < 16; g4 += 4)
              sum += ((sred[g4][lane] + sred[g4 + 1][lane]) + sred[g4 + 2][lane]) + sred[g4 + 3][lane];
          }
          part[(((size_t)blockIdx.x * (MA * NB) + (a * NB + b)) * 4 + r) * 64 + lane] = sum;
        }
      }
    }
  if (!counter || NW != 4) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == gridDim.x - 1;
    if (last) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  const int nparts = (int)gridDim.x, per = (nparts + 3) / 4, pos = threadIdx.x;
  const size_t stride = (size_t)MA * NB * 256;
  for (int tile = 0; tile < MA * NB; ++tile) {
    const double *src = part + (size_t)tile * 256 + pos;
    double qs[4];
#pragma unroll
    for (int quarter = 0; quarter < 4; ++quarter) {
      const int p0 = quarter * per, p1 = min(nparts, p0 + per);
      double sacc = 0.0;
      int p = p0;
      for (; p + 7 < p1; p += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(p + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc += v[u];
      }
      for (; p < p1; ++p) sacc += src[(size_t)p * stride];
      qs[quarter] = sacc;
    }
    const double sum = ((qs[0] + qs[1]) + qs[2]) + qs[3];
    const int a = tile / NB, b = tile % NB, r = pos >> 6, q = (pos & 63) >> 4, c = pos & 15;
    const int i = 16 * a + q + 4 * r, j = 16 * b + c;
    if (i < K && j < T) {
      if (out64) out64[i * T + j] = sum;
      if (out32) out32[i * T + j] = (float)sum;
    }
  }
}

// One workgroup of 1024 threads per 16 x 16 tile: thread (quarter, position) sums its quarter of the slabs for one of the
// 256 result positions (coalesced across positions, 8 independent loads in flight), the quarters are combined in order
// through LDS.  (First version: one thread per output element walking all 512 slabs — 122 us at the C5 shape.)
__global__ __launch_bounds__(1024) void gram_finish_kernel(const double *__restrict__ part, int nparts, int MA, int NB, int K,
                                                           int T, double *__restrict__ out, float *__restrict__ out32) {
  __shared__ double sq[4][256];
  const int tile = blockIdx.x, pos = threadIdx.x & 255, quarter = threadIdx.x >> 8;
  const size_t stride = (size_t)MA * NB * 256;
  const int per = (nparts + 3) / 4;
  const int p0 = quarter * per, p1 = min(nparts, p0 + per);
  const double *src = part + (size_t)tile * 256 + pos;
  double s = 0.0;
  int p = p0;
  for (; p + 7 < p1; p += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(p + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; p < p1; ++p) s += src[(size_t)p * stride];
  sq[quarter][pos] = s;
  __syncthreads();
  if (quarter == 0) {
    const double sum = ((sq[0][pos] + sq[1][pos]) + sq[2][pos]) + sq[3][pos];
    // position = r * 64 + lane, lane = q * 16 + c: result element (16 a + q + 4 r, 16 b + c)
    const int a = tile / NB, b = tile % NB, r = pos >> 6, q = (pos & 63) >> 4, c = pos & 15;
    const int i = 16 * a + q + 4 * r, j = 16 * b + c;
    if (i < K && j < T) {
      if (out) out[i * T + j] = sum;
      if (out32) out32[i * T + j] = (float)sum;
    }
  }
}

__global__ __launch_bounds__(256) void woodbury_apply_kernel(const float *__restrict__ L, long long ldl,
                                                             const float *__restrict__ R, long long ldr,
                                                             const double *__restrict__ Tm, double inv_noise,
                                                             float *__restrict__ out, long long ldo, long long N, int K,
                                                             int T, const double *__restrict__ Cinv) {
  extern __shared__ double sT[];                    // K x T
  if (Cinv) {                                       // Tm = L^T R: every workgroup forms t = C^-1 Tm itself (K^2 T products)
    for (int e = threadIdx.x; e < K * T; e += 256) {
      const int k = e / T, t = e - k * T;
      double acc = 0.0;
      for (int kk = 0; kk < K; ++kk) acc = fma(Cinv[k * K + kk], Tm[kk * T + t], acc);
      sT[e] = acc;
    }
  } else {
    for (int e = threadIdx.x; e < K * T; e += 256) sT[e] = Tm[e];
  }
  __syncthreads();
  const long long total = N * T;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
    const long long n = g / T;
    const int t = (int)(g - n * T);
    double acc = (double)R[n * ldr + t];
    const float *lrow = L + n * ldl;
    for (int k = 0; k < K; ++k) acc = fma(-(double)lrow[k], sT[k * T + t], acc);
    out[n * ldo + t] = (float)(acc * inv_noise);
  }
}

// Capacitance matrix of the Woodbury preconditioner in ONE launch (K <= 64, one workgroup): C = G + noise I (G = L^T L
// from rpgp_gram_f64), its lower Cholesky factor, C^-1 and log|C|.  Replaces a diagonal add, the library's potrf (+ info
// check = a device synchronisation), log / sum and potri: ~15 launches of 15 x 15 float64 work per optimiser step.
// A non-positive pivot writes NaN into everything (the caller checks log|C| where it synchronises anyway).
__global__ __launch_bounds__(256) void woodbury_setup_kernel(const double *__restrict__ G, double noise, int K,
                                                             double *__restrict__ chol, double *__restrict__ cinv,
                                                             double *__restrict__ logdet, double *__restrict__ logdet_host) {
  __shared__ double sA[64 * 65];      // C, then its Cholesky factor (lower), row stride 65
  __shared__ double sI[64 * 65];      // inverse of the factor (lower)
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  for (int e = tid; e < K * K; e += 256) {
    const int i = e / K, j = e % K;
    sA[i * 65 + j] = G[e] + (i == j ? noise : 0.0);
  }
  __syncthreads();
  // right-looking Cholesky: column k scaled by thread group, trailing update in parallel
  for (int k = 0; k < K; ++k) {
    const double d = sA[k * 65 + k];
    if (tid == 0 && !(d > 0.0)) bad = 1;
    __syncthreads();
    const double r = sqrt(d);
    for (int i = k + tid; i < K; i += 256) sA[i * 65 + k] = (i == k) ? r : sA[i * 65 + k] / r;
    __syncthreads();
    for (int e = tid; e < (K - k - 1) * (K - k - 1); e += 256) {
      const int i = k + 1 + e / (K - k - 1), j = k + 1 + e % (K - k - 1);
      if (j <= i) sA[i * 65 + j] -= sA[i * 65 + k] * sA[j * 65 + k];
    }
    __syncthreads();
  }
  // inverse of the lower factor, one column per thread (forward substitution)
  if (tid < K) {
    const int c = tid;
    for (int i = 0; i < K; ++i) {
      double v = (i == c) ? 1.0 : 0.0;
      for (int m = c; m < i; ++m) v -= sA[i * 65 + m] * sI[m * 65 + c];
      sI[i * 65 + c] = (i < c) ? 0.0 : v / sA[i * 65 + i];
    }
  }
  __syncthreads();
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  for (int e = tid; e < K * K; e += 256) {
    const int i = e / K, j = e % K;
    double acc = 0.0;                                   // C^-1 = Linv^T Linv
    for (int m = (i > j ? i : j); m < K; ++m) acc += sI[m * 65 + i] * sI[m * 65 + j];
    cinv[e] = bad ? nanv : acc;
    chol[e] = bad ? nanv : (j <= i ? sA[i * 65 + j] : 0.0);
  }
  if (tid == 0) {
    double ld = 0.0;
    for (int i = 0; i < K; ++i) ld += log(sA[i * 65 + i]);
    logdet[0] = bad ? nanv : 2.0 * ld;
    if (logdet_host) {                 // (pinned host memory: the value the host reads behind the solve — no copy launch)
      logdet_host[0] = bad ? nanv : 2.0 * ld;
      __threadfence_system();
    }
  }
}

// arrival counters of the folded finish: zero at allocation, left at zero by every launch; 64 per device, handed out round-robin
// (two launches share a counter only if 64 others are in flight between them)
inline unsigned *gram_counter() {
  constexpr int kSlots = 64, kMaxDev = 16;
  static std::atomic<unsigned *> pool[kMaxDev];
  static std::atomic<unsigned> next[kMaxDev];
  static std::mutex init_lock;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
  unsigned *base = pool[dev].load(std::memory_order_acquire);
  if (!base) {
    std::lock_guard<std::mutex> g(init_lock);
    base = pool[dev].load(std::memory_order_acquire);
    if (!base) {
      unsigned *p = nullptr;
      if (hipMalloc(reinterpret_cast<void **>(&p), kSlots * 64) != hipSuccess) return nullptr;
      if (hipMemset(p, 0, kSlots * 64) != hipSuccess) return nullptr;
      pool[dev].store(p, std::memory_order_release);
      base = p;
    }
  }
  const unsigned slot = next[dev].fetch_add(1u, std::memory_order_relaxed) % kSlots;
  return base + slot * 16;
}
inline int tiles16(int n) { return (n + 15) / 16; }
inline int gram_blocks(long long N) {
  long long g = (N + 255) / 256;
  if (g < 1) g = 1;
  if (g > kGramMaxWg) g = kGramMaxWg;
  return (int)g;
}

}  // namespace

namespace rpgp_internal {

size_t gram_part_bytes(int K, int T) {
  if (K <= 0 || T <= 0 || K > 64 || T > 64) return 0;
  int ma = tiles16(K), nb = tiles16(T);
  if (ma == 3) ma = 4;
  if (nb == 3) nb = 4;
  return (size_t)kGramMaxWg * ma * nb * 256 * sizeof(double);
}

// out64 and / or out32 [K x T] = A^T B; `part` holds gram_part_bytes(K, T)
int gram_launch(const float *A, long long lda, const float *B, long long ldb, long long N, int K, int T, double *out64,
                float *out32, double *part, hipStream_t st) {
  const int g = gram_blocks(N);
  int ma = tiles16(K), nb = tiles16(T);
  if (ma == 3) ma = 4;                               // compiled tile counts: 1, 2, 4
  if (nb == 3) nb = 4;
  // few slabs: the last workgroup of the partial kernel finishes (no second launch)
  unsigned *counter = nullptr;
  if (g <= 64) counter = gram_counter();
  const bool tall = N >= 131072 && !counter;           // (sixteen waves per workgroup; the small tile counts only)
#define RPGP_GRAM_CASE(MA_, NB_)                                                                                       \
  if (ma == MA_ && nb == NB_) {                                                                                        \
    if (tall && MA_ * NB_ <= 2)                                                                                        \
      hipLaunchKernelGGL((gram_partial_kernel<MA_, NB_, 16>), dim3(g), dim3(1024), 0, st, A, lda, B, ldb, N, K, T, part,       \
                         counter, out64, out32);                                                                       \
    else                                                                                                               \
      hipLaunchKernelGGL((gram_partial_kernel<MA_, NB_, 4>), dim3(g), dim3(256), 0, st, A, lda, B, ldb, N, K, T, part,        \
                         counter, out64, out32);                                                                       \
  }
  RPGP_GRAM_CASE(1, 1); RPGP_GRAM_CASE(1, 2); RPGP_GRAM_CASE(1, 4);
  RPGP_GRAM_CASE(2, 1); RPGP_GRAM_CASE(2, 2); RPGP_GRAM_CASE(2, 4);
  RPGP_GRAM_CASE(4, 1); RPGP_GRAM_CASE(4, 2); RPGP_GRAM_CASE(4, 4);
#undef RPGP_GRAM_CASE
  if (!counter) hipLaunchKernelGGL(gram_finish_kernel, dim3(ma * nb), dim3(1024), 0, st, part, g, ma, nb, K, T, out64, out32);
  return (int)hipGetLastError();
}

}  // namespace rpgp_internal

extern "C" {

size_t rpgp_gram_f64_workspace_bytes(int K, int T) { return rpgp_internal::gram_part_bytes(K, T); }

int rpgp_gram_f64(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t N, int K, int T, double *out,
                  void *workspace, size_t workspace_bytes, void *stream) {
  if (!A || !B || !out || N < 0 || K <= 0 || T <= 0 || K > 64 || T > 64 || lda < K || ldb < T) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_gram_f64_workspace_bytes(K, T)) return RPGP_EWORKSPACE;
  return rpgp_internal::gram_launch(A, (long long)lda, B, (long long)ldb, (long long)N, K, T, out, nullptr,
                                    reinterpret_cast<double *>(workspace), reinterpret_cast<hipStream_t>(stream));
}

int rpgp_woodbury_setup(const double *gram, double noise, int K, double *chol, double *cinv, double *logdet, void *stream) {
  return rpgp_woodbury_setup_pinned(gram, noise, K, chol, cinv, logdet, nullptr, stream);
}

int rpgp_woodbury_setup_pinned(const double *gram, double noise, int K, double *chol, double *cinv, double *logdet,
                               double *logdet_pinned_host, void *stream) {
  if (!gram || !chol || !cinv || !logdet || K <= 0 || K > 64 || !(noise > 0.0)) return RPGP_EINVAL;
  double *hdev = nullptr;
  if (logdet_pinned_host &&
      hipHostGetDevicePointer(reinterpret_cast<void **>(&hdev), logdet_pinned_host, 0) != hipSuccess)
    return RPGP_EINVAL;
  hipLaunchKernelGGL(woodbury_setup_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), gram, noise, K,
                     chol, cinv, logdet, hdev);
  return (int)hipGetLastError();
}

static int woodbury_apply_impl(const float *L, int64_t ldl, const float *R, int64_t ldr, const double *Tm, const double *Cinv,
                               double noise, float *out, int64_t ldo, int64_t N, int K, int T, void *stream) {
  if (!L || !R || !Tm || !out || N < 0 || K <= 0 || T <= 0 || K > 64 || T > 64 || ldl < K || ldr < T || ldo < T ||
      !(noise > 0.0))
    return RPGP_EINVAL;
  if (N == 0) return 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  long long blocks = (N * T + 255) / 256;
  // (every workgroup of the fused form pays K^2 T products at its head — 2 250 for the rank-15 preconditioner of a 10-probe
  //  block, nine per thread; a cap of 1 024 workgroups made a thread of the C5 operator walk 15 elements one round trip at a time)
  if (blocks > (Cinv ? 4096 : 8192)) blocks = Cinv ? 4096 : 8192;
  hipLaunchKernelGGL(woodbury_apply_kernel, dim3((int)blocks), dim3(256), (size_t)K * T * sizeof(double), st, L,
                     (long long)ldl, R, (long long)ldr, Tm, 1.0 / noise, out, (long long)ldo, (long long)N, K, T, Cinv);
  return (int)hipGetLastError();
}

int rpgp_woodbury_apply(const float *L, int64_t ldl, const float *R, int64_t ldr, const double *Tm, double noise,
                        float *out, int64_t ldo, int64_t N, int K, int T, void *stream) {
  return woodbury_apply_impl(L, ldl, R, ldr, Tm, nullptr, noise, out, ldo, N, K, T, stream);
}

int rpgp_woodbury_apply_cinv(const float *L, int64_t ldl, const float *R, int64_t ldr, const double *gram_LR,
                             const double *Cinv, double noise, float *out, int64_t ldo, int64_t N, int K, int T,
                             void *stream) {
  if (!Cinv) return RPGP_EINVAL;
  return woodbury_apply_impl(L, ldl, R, ldr, gram_LR, Cinv, noise, out, ldo, N, K, T, stream);
}

}  // extern "C"
