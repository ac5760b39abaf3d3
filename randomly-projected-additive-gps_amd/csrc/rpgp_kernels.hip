// rpgp_kernels.hip — hand-written gfx950 (CDNA4) kernels + the C-ABI of include/rpgp.h.
//
// Hot path: the additive randomly-projected RBF kernel
//     K[i,i'] = scale * sum_j exp(-0.5 (Z[i,j]-Z[i',j])^2)
// applied to a block of vectors without ever storing K.  gfx950 only: wave64, DPP wave rotates,
// packed fp32 VALU, 160 KB LDS.  No CUDA/compat paths.
//
// Tiling of the fused MVM (mvm_tile_kernel):
//   * a 256-thread workgroup owns BR = 256*R rows; each lane keeps R rows' JT projected coordinates
//     (pre-multiplied by sqrt(0.5*log2 e) so the inner loop is sub, mul, v_exp_f32, add) and its
//     R*TT row accumulators in VGPRs;
//   * the workgroup sweeps one chunk of columns, staged 256 at a time in LDS with coalesced loads;
//   * inside a 64-column subtile, lane l visits column (l + s) mod 64 at step s (a per-lane LDS read,
//     bank-conflict free for the padded strides used).  Because every lane holds a DIFFERENT column,
//     the transposed product  out[i'] += K[i,i'] v[i]  needs no cross-lane reduction: its accumulator
//     is rotated one lane per step with a DPP wave rotate, so after 64 steps lane l holds column l.
//     Each unordered pair is therefore evaluated once (SYM=true), halving the exp count;
//   * partial results go to per-(row-block / chunk) slabs in a caller-provided workspace and a small
//     reduce kernel sums them in a fixed order: deterministic, no float atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stddef.h>
#include <atomic>
#include <type_traits>

#include "../../include/rpgp.h"
#include "rpgp_internal.h"

namespace {

constexpr float kExp2Scale = 0.8493218002880191f;  // sqrt(0.5 * log2(e)):  exp(-d^2/2) = exp2(-(c d)^2)
constexpr int kSC = 256;                           // columns staged in LDS per sub-chunk (T <= 4)
// wide right-hand sides stage fewer columns so that LDS (sT = 4*SC*TT floats) still admits several workgroups per CU
template <int TT> struct StageCols { static constexpr int v = (TT > 4) ? 64 : 256; };

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float floatx4m __attribute__((ext_vector_type(4)));

// LDS row stride (floats) for a column's JT coordinates: (stride/4) odd keeps the per-lane
// ds_read_b128 of 16 consecutive columns on 16 distinct 4-bank slots (MI355X_MICROARCH.md §LDS).
template <int JT> struct ColStride { static constexpr int v = (JT % 8 == 0) ? JT + 4 : JT; };

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// DPP wave rotate by one lane (gfx9: wave_rol:1 = 0x134).  Direction is probed once on the host
// (rpgp_init) and passed to kernels as `rotdir`, so correctness does not rest on the mnemonic.
__device__ __forceinline__ float wave_rotate1(float x) {
  int xi = __builtin_bit_cast(int, x);
  // bound_ctrl = true: every lane of a wave rotate has a valid source, and the compiler no longer has to materialise the
  // `old` operand with an extra v_mov_b32 per rotation (24 per two steps in the T = 12 kernel)
  int r = __builtin_amdgcn_update_dpp(0, xi, 0x134, 0xf, 0xf, true);
  return __builtin_bit_cast(float, r);
}

__global__ void probe_rotate_kernel(int *out) {
  float x = (float)threadIdx.x;
  float y = wave_rotate1(x);
  out[threadIdx.x] = (int)y;
}

// sum_j exp2(-(a_j - b_j)^2), written on float2 so hipcc emits v_pk_add_f32 / v_pk_mul_f32.
template <int JT>
__device__ __forceinline__ float pair_kernel_sum(const float (&a)[JT], const float (&b)[JT]) {
  if constexpr (JT % 2 == 0) {
    float2v acc = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < JT; j += 2) {
      float2v av = {a[j], a[j + 1]};
      float2v bv = {b[j], b[j + 1]};
      float2v d = av - bv;
      float2v m = -(d * d);
      float2v e = {fast_exp2(m.x), fast_exp2(m.y)};
      acc += e;
    }
    float s;                      // one v_add_f32 (asm): keeps the SLP vectoriser from packing neighbouring rows' adds behind moves
    asm("v_add_f32 %0, %1, %2" : "=v"(s) : "v"(acc.x), "v"(acc.y));
    return s;
  } else {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      float d = a[j] - b[j];
      acc += fast_exp2(-(d * d));
    }
    return acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Kernel-function policies.  The hot path is KfBase (unweighted sum of 1-D RBFs).  The other members of the
// reference's additive family (`kernel_type` of training_routines.py:47-88, k > 1 sub-kernels of
// :172-174, per-component weights of polynomial_projection_kernels.py:88-98) run through the SAME tile / dense kernels
// with a different policy:
//     K[i,i'] = scale * sum_c w[c] * phi(columns [c*G, (c+1)*G) of Z)
// `pre` is the factor applied to Z when it is loaded (so the inner loop works on prescaled differences dd), `val` the
// 1-D function of dd, `val_grad` additionally returns g with  d phi / d z_i = mulG * g.
// ---------------------------------------------------------------------------------------------
template <int KIND> struct Phi;
template <> struct Phi<RPGP_KIND_RBF> {                 // exp(-d^2 / 2) = exp2(-(c d)^2)
  static constexpr float pre = kExp2Scale;
  static constexpr float mulG = -1.0f / kExp2Scale;
  static __device__ __forceinline__ float val(float dd) { return fast_exp2(-(dd * dd)); }
  static __device__ __forceinline__ void val_grad(float dd, float &phi, float &g) {
    phi = fast_exp2(-(dd * dd));
    g = phi * dd;
  }
};
template <> struct Phi<RPGP_KIND_MATERN15> {            // (1 + sqrt3 r) exp(-sqrt3 r), dd = sqrt3 (z - z')
  static constexpr float pre = 1.7320508075688772f;
  static constexpr float mulG = -1.7320508075688772f;
  static __device__ __forceinline__ float val(float dd) {
    const float u = __builtin_fabsf(dd);
    return (1.0f + u) * fast_exp2(-1.4426950408889634f * u);
  }
  static __device__ __forceinline__ void val_grad(float dd, float &phi, float &g) {
    const float u = __builtin_fabsf(dd);
    const float e = fast_exp2(-1.4426950408889634f * u);
    phi = (1.0f + u) * e;
    g = dd * e;                                         // d phi / d dd = -dd e
  }
};
template <> struct Phi<RPGP_KIND_IMQ> {                 // (1 + r^2)^(-1/2)   (imq_kernel.py:8-9)
  static constexpr float pre = 1.0f;
  static constexpr float mulG = -1.0f;
  static __device__ __forceinline__ float val(float dd) { return __builtin_amdgcn_rsqf(__builtin_fmaf(dd, dd, 1.0f)); }
  static __device__ __forceinline__ void val_grad(float dd, float &phi, float &g) {
    phi = __builtin_amdgcn_rsqf(__builtin_fmaf(dd, dd, 1.0f));
    g = dd * phi * phi * phi;
  }
};
template <> struct Phi<RPGP_KIND_COSINE> {              // cos(pi r) (period 1); v_cos/v_sin take revolutions: dd = (z - z') / 2
  static constexpr float pre = 0.5f;
  static constexpr float mulG = -3.14159265358979323846f;
  static __device__ __forceinline__ float val(float dd) { return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(dd)); }
  static __device__ __forceinline__ void val_grad(float dd, float &phi, float &g) {
    const float f = __builtin_amdgcn_fractf(dd);
    phi = __builtin_amdgcn_cosf(f);
    g = __builtin_amdgcn_sinf(f);
  }
};

struct KfBase {                                         // the hot path: sum_j exp2(-(a_j - b_j)^2), weights folded in `scale`
  static constexpr float pre = kExp2Scale;
  static constexpr int group = 1;
  static constexpr bool weighted = false;
  template <int JT>
  static __device__ __forceinline__ float pair_sum(const float (&a)[JT], const float (&b)[JT], const float *) {
    return pair_kernel_sum<JT>(a, b);
  }
};

template <int KIND>
struct KfPhi {                                          // weighted sum of 1-D functions
  static constexpr float pre = Phi<KIND>::pre;
  static constexpr float mulG = Phi<KIND>::mulG;
  static constexpr int group = 1;
  static constexpr bool weighted = true;
  template <int JT>
  static __device__ __forceinline__ float pair_sum(const float (&a)[JT], const float (&b)[JT], const float *w) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < JT; ++j) acc = __builtin_fmaf(w[j], Phi<KIND>::val(a[j] - b[j]), acc);
    return acc;
  }
  // per-column gradient factors and per-component values for the bilinear derivative
  template <int JT>
  static __device__ __forceinline__ void pair_grad(const float (&a)[JT], const float *b, const float *w, float S,
                                                   float (&accG)[JT], float (&accC)[JT]) {
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      float phi, g;
      Phi<KIND>::val_grad(a[j] - b[j], phi, g);
      accG[j] = __builtin_fmaf(S * w[j], g, accG[j]);
      accC[j] = __builtin_fmaf(S, phi, accC[j]);
    }
  }
};

template <int G>
struct KfGroupRbf {                                     // weighted sum of G-dimensional RBFs (products of G 1-D RBFs)
  static constexpr float pre = kExp2Scale;
  static constexpr float mulG = -1.0f / kExp2Scale;
  static constexpr int group = G;
  static constexpr bool weighted = true;
  template <int JT>
  static __device__ __forceinline__ float pair_sum(const float (&a)[JT], const float (&b)[JT], const float *w) {
    static_assert(JT % G == 0, "column pieces must hold whole groups");
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < JT / G; ++c) {
      float r2 = 0.f;
#pragma unroll
      for (int m = 0; m < G; ++m) {
        const float dd = a[c * G + m] - b[c * G + m];
        r2 = __builtin_fmaf(dd, dd, r2);
      }
      acc = __builtin_fmaf(w[c], fast_exp2(-r2), acc);
    }
    return acc;
  }
  template <int JT>
  static __device__ __forceinline__ void pair_grad(const float (&a)[JT], const float *b, const float *w, float S,
                                                   float (&accG)[JT], float (&accC)[JT / G]) {
#pragma unroll
    for (int c = 0; c < JT / G; ++c) {
      float dd[G];
      float r2 = 0.f;
#pragma unroll
      for (int m = 0; m < G; ++m) {
        dd[m] = a[c * G + m] - b[c * G + m];
        r2 = __builtin_fmaf(dd[m], dd[m], r2);
      }
      const float phi = fast_exp2(-r2);
      const float sw = S * w[c] * phi;
#pragma unroll
      for (int m = 0; m < G; ++m) accG[c * G + m] = __builtin_fmaf(sw, dd[m], accG[c * G + m]);
      accC[c] = __builtin_fmaf(S, phi, accC[c]);
    }
  }
};

template <int JT>
__device__ __forceinline__ void lds_load_cols(const float *sB, int idx, float (&b)[JT]) {
  constexpr int STR = ColStride<JT>::v;
  const float *p = sB + idx * STR;
  if constexpr (JT % 4 == 0) {
#pragma unroll
    for (int j = 0; j < JT; j += 4) {
      float4v q = *reinterpret_cast<const float4v *>(p + j);
      b[j] = q.x; b[j + 1] = q.y; b[j + 2] = q.z; b[j + 3] = q.w;
    }
  } else if constexpr (JT % 2 == 0) {
#pragma unroll
    for (int j = 0; j < JT; j += 2) {
      float2v q = *reinterpret_cast<const float2v *>(p + j);
      b[j] = q.x; b[j + 1] = q.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < JT; ++j) b[j] = p[j];
  }
}

template <int TT>
__device__ __forceinline__ void lds_load_vec(const float *sV, int idx, float (&v)[TT]) {
  const float *p = sV + idx * TT;
  if constexpr (TT % 4 == 0) {
#pragma unroll
    for (int t = 0; t < TT; t += 4) {
      float4v q = *reinterpret_cast<const float4v *>(p + t);
      v[t] = q.x; v[t + 1] = q.y; v[t + 2] = q.z; v[t + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int t = 0; t < TT; ++t) v[t] = p[t];
  }
}

// ---------------------------------------------------------------------------------------------
// Fused MVM tile kernel.  grid = (max chunks per row block, row blocks); block = 256.
//   SYM : Z1 == Z2 (N x N), columns of row block rb start at its own first row (upper triangle incl.
//         the diagonal block); the transposed product is produced for columns beyond the diagonal block.
//   !SYM: rectangular M x N, every row block sweeps all columns.
// slabR[k][row][T]  : partial row products of chunk k           (k < chunks of that row block)
// slabT[rb][col][T] : partial transposed products of row block rb (only col >= (rb+1)*BR written)
// ---------------------------------------------------------------------------------------------
// Linear workgroup index -> (row block, column chunk).  Row block b owns ceil((N - cbase(b)) / chunk) chunks (cbase = its
// own first row for the symmetric sweep, 0 otherwise); workgroups are numbered row block by row block.  A rank's
// launch covers the contiguous range [w0, w0 + gridDim.x) of that numbering (pair-sharding at workgroup granularity).
__device__ __forceinline__ void wg_to_tile(int lin, int N, int BR, int chunk, bool sym, int &rb, int &kchunk) {
  int b = 0, acc = 0;
  for (;;) {
    const int cbase = sym ? b * BR : 0;
    const int cb = (N - cbase + chunk - 1) / chunk;
    if (lin < acc + cb) break;
    acc += cb;
    ++b;
  }
  rb = b;
  kchunk = lin - acc;
}

template <int JT, int TT, int R, bool SYM, class KF = KfBase>
__global__ __launch_bounds__(256) void mvm_tile_kernel(
    const float *__restrict__ Z1, const float *__restrict__ Z2, const float *__restrict__ V,
    float *__restrict__ slabR, float *__restrict__ slabT, int M, int N, int ldz1, int ldz2, int ldv,
    int j0, int t0, int tcnt, int chunk_cols, int rotdir, int accumulate, int w0, int rb_first, int slab_row0,
    int slab_rows, const float *__restrict__ wts) {
  constexpr int BR = 256 * R;
  constexpr int SC = StageCols<TT>::v;
  constexpr int STR = ColStride<JT>::v;
  __shared__ __attribute__((aligned(16))) float sB[SC * STR];
  __shared__ __attribute__((aligned(16))) float sV[SC * TT];
  __shared__ __attribute__((aligned(16))) float sT[SYM ? 4 * SC * TT : 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  int rb, kchunk;
  wg_to_tile(blockIdx.x + w0, SYM ? N : N, BR, chunk_cols, SYM, rb, kchunk);
  const int r0 = rb * BR;
  const int cbase = SYM ? r0 : 0;
  const long long cb = (long long)cbase + (long long)kchunk * chunk_cols;
  if (cb >= N) return;
  const int c_begin = (int)cb;
  const int c_end = (c_begin + chunk_cols < N) ? c_begin + chunk_cols : N;

  // component weights of this column piece (uniform; unused by the hot-path policy)
  float wreg[KF::weighted ? JT / KF::group : 1];
  if constexpr (KF::weighted) {
#pragma unroll
    for (int c = 0; c < JT / KF::group; ++c) wreg[c] = wts[j0 / KF::group + c];
  } else {
    wreg[0] = 0.f;
  }

  float a[R][JT];
  float vrow[R][TT];
  float accR[R][TT];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = r0 + wave * (64 * R) + r * 64 + lane;
    const bool valid = row < M;
    // unconditional loads from a clamped row (a load under a condition compiles to a branch + s_waitcnt vmcnt(0) per
    // element: 40 serial round trips here); rows beyond M carry a valid row's coordinates, are never stored, and their
    // right-hand side is masked to zero, so they contribute nothing
    const int rowc = valid ? row : M - 1;
#pragma unroll
    for (int j = 0; j < JT; ++j) a[r][j] = Z1[(size_t)rowc * ldz1 + j0 + j] * KF::pre;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if constexpr (SYM) vrow[r][t] = V[(size_t)rowc * ldv + t0 + (t < tcnt ? t : 0)] * ((valid && t < tcnt) ? 1.f : 0.f);
      else vrow[r][t] = 0.f;
      accR[r][t] = 0.f;
    }
  }

  for (int c0 = c_begin; c0 < c_end; c0 += SC) {
    __syncthreads();
    if (tid < SC) {
      const int col = c0 + tid;
      const bool cv = col < c_end;
      const int colc = cv ? col : N - 1;            // (same: clamped column, its right-hand side masked to zero)
      float zc[JT], vc[TT];
#pragma unroll
      for (int j = 0; j < JT; ++j) zc[j] = Z2[(size_t)colc * ldz2 + j0 + j];
#pragma unroll
      for (int t = 0; t < TT; ++t) vc[t] = V[(size_t)colc * ldv + t0 + (t < tcnt ? t : 0)];
#pragma unroll
      for (int j = 0; j < JT; ++j) sB[tid * STR + j] = zc[j] * KF::pre;
#pragma unroll
      for (int t = 0; t < TT; ++t) sV[tid * TT + t] = vc[t] * ((cv && t < tcnt) ? 1.f : 0.f);
    }
    __syncthreads();
    const int ncol = c_end - c0;
    const int nsub = ncol >= SC ? SC / 64 : (ncol + 63) / 64;
    for (int sub = 0; sub < nsub; ++sub) {
      const bool doT = SYM && (c0 + sub * 64 >= r0 + BR);
      float accT[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) accT[t] = 0.f;
      if (doT) {
#pragma unroll 2
        for (int s = 0; s < 64; ++s) {
          const int idx = sub * 64 + ((lane + rotdir * s) & 63);
          float b[JT], v[TT];
          lds_load_cols<JT>(sB, idx, b);
          lds_load_vec<TT>(sV, idx, v);
          float tsum[TT];
#pragma unroll
          for (int t = 0; t < TT; ++t) tsum[t] = accT[t];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float ks = KF::template pair_sum<JT>(a[r], b, wreg);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
              accR[r][t] = __builtin_fmaf(ks, v[t], accR[r][t]);
              tsum[t] = __builtin_fmaf(ks, vrow[r][t], tsum[t]);
            }
          }
#pragma unroll
          for (int t = 0; t < TT; ++t) accT[t] = wave_rotate1(tsum[t]);
        }
      } else {
#pragma unroll 2
        for (int s = 0; s < 64; ++s) {
          const int idx = sub * 64 + ((lane + s) & 63);
          float b[JT], v[TT];
          lds_load_cols<JT>(sB, idx, b);
          lds_load_vec<TT>(sV, idx, v);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float ks = KF::template pair_sum<JT>(a[r], b, wreg);
#pragma unroll
            for (int t = 0; t < TT; ++t) accR[r][t] = __builtin_fmaf(ks, v[t], accR[r][t]);
          }
        }
      }
      if constexpr (SYM) {
#pragma unroll
        for (int t = 0; t < TT; ++t) sT[(wave * SC + sub * 64 + lane) * TT + t] = accT[t];
      }
    }
    if constexpr (SYM) {
      __syncthreads();
      const int col = c0 + tid;
      if (tid < SC && col < c_end && col >= r0 + BR) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          if (t < tcnt) {
            float sum = sT[(0 * SC + tid) * TT + t] + sT[(1 * SC + tid) * TT + t] +
                        sT[(2 * SC + tid) * TT + t] + sT[(3 * SC + tid) * TT + t];
            float *dst = slabT + ((size_t)(rb - rb_first) * N + col) * ldv + t0 + t;
            *dst = accumulate ? *dst + sum : sum;
          }
        }
      }
    }
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = r0 + wave * (64 * R) + r * 64 + lane;
    if (row < M) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (t < tcnt) {
          float *dst = slabR + ((size_t)kchunk * slab_rows + (row - slab_row0)) * ldv + t0 + t;
          *dst = accumulate ? *dst + accR[r][t] : accR[r][t];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Factorised fast path ("prepared" MVM).
//   exp2(-(a-b)^2) = exp2(-a^2) * exp2(2ab - b^2)
// so per pair-term the loop is  t = fma(a, 2b, -b^2);  e = v_exp_f32(t);  K = fma(e, exp2(-a^2), K)
// (3 VALU issues instead of 4: the subtract disappears).  2ab - b^2 <= a^2 must stay below the fp32 exponent range,
// so rpgp_prepare centres every projection at the midpoint of its range and records max|a|; the host only takes this
// path when max a^2 < kFactMaxSq, otherwise the exact direct kernel above is used.
// prep buffer: [header 256 B][mid J floats, padded to 256 B][rowdat N*J float2 {a, Ea}][coldat N*J float2 {2a, -a^2}]
// ---------------------------------------------------------------------------------------------
constexpr float kFactMaxSq = 100.0f;   // a^2 bound (|z - mid| < 11.8): exp2(100) is far inside fp32 range
constexpr int kPrepHeaderFloats = 64;
constexpr int kPrepMidFloats = 64;     // J <= 64 projections per prepared buffer

struct PrepLayout {
  float *header;
  float *mid;
  float2v *rowdat;
  float2v *coldat;
};

__host__ __device__ inline PrepLayout prep_layout(void *prep, long long N, int J) {
  PrepLayout L;
  L.header = reinterpret_cast<float *>(prep);
  L.mid = L.header + kPrepHeaderFloats;
  L.rowdat = reinterpret_cast<float2v *>(L.mid + kPrepMidFloats);
  L.coldat = L.rowdat + (size_t)N * J;
  return L;
}

// NaN-propagating min / max so that non-finite inputs are caught by the range guard
__device__ __forceinline__ float min_nan(float a, float b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
__device__ __forceinline__ float max_nan(float a, float b) { return (a != a) ? a : ((b != b) ? b : (b > a ? b : a)); }

// per-block partial min/max of every projection.  Thread t handles projection t % J of rows n0 + t / J, n0 + t / J + 256 / J, ...;
// eight rows' loads are requested together (clamped to a valid row, masked afterwards: a loop of one load per trip is a chain of
// dependent round trips — 60 us at C4 in its first form, VERDICT r5 #6a)
__global__ __launch_bounds__(256) void prep_minmax_kernel(const float *__restrict__ Z, float *__restrict__ part,
                                                          long long N, int ldz, int J, long long rows_per_block) {
  __shared__ float smin[256], smax[256];
  const long long n0 = (long long)blockIdx.x * rows_per_block;
  const long long n1 = (n0 + rows_per_block < N) ? n0 + rows_per_block : N;
  const int j = threadIdx.x % J;
  const int rstep = 256 / J;
  float mn = 3.4e38f, mx = -3.4e38f;
  if ((int)threadIdx.x < rstep * J && n0 < n1) {
    for (long long n = n0 + threadIdx.x / J; n < n1; n += 8LL * rstep) {
      float z[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long r = n + (long long)u * rstep;
        z[u] = Z[(r < n1 ? r : n1 - 1) * ldz + j];            // (a clamped row repeats a value of this thread's own column)
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        mn = min_nan(mn, z[u]);
        mx = max_nan(mx, z[u]);
      }
    }
  }
  smin[threadIdx.x] = mn;
  smax[threadIdx.x] = mx;
  __syncthreads();
  if ((int)threadIdx.x < J) {
    for (int t = threadIdx.x + J; t < rstep * J; t += J) {
      mn = min_nan(mn, smin[t]);
      mx = max_nan(mx, smax[t]);
    }
    part[((size_t)blockIdx.x * J + threadIdx.x) * 2 + 0] = mn;
    part[((size_t)blockIdx.x * J + threadIdx.x) * 2 + 1] = mx;
  }
}

__global__ __launch_bounds__(256) void prep_finish_kernel(const float *__restrict__ part, int nparts, int J,
                                                          float *__restrict__ header, float *__restrict__ mid) {
  __shared__ float smin[256], smax[256], shalf[64];
  // thread t folds the partials p = t / J, t / J + 256 / J, ... of projection t % J (min / max are exact: any order), eight
  // records in flight at a time
  const int j = threadIdx.x % J, pstep = 256 / J;
  float mn = 3.4e38f, mx = -3.4e38f;
  if ((int)threadIdx.x < pstep * J) {
    const float2v *pr = reinterpret_cast<const float2v *>(part);
    for (int p = threadIdx.x / J; p < nparts; p += 8 * pstep) {
      float2v v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = p + u * pstep;
        v[u] = pr[(size_t)(q < nparts ? q : nparts - 1) * J + j];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        mn = min_nan(mn, v[u].x);
        mx = max_nan(mx, v[u].y);
      }
    }
  }
  smin[threadIdx.x] = mn;
  smax[threadIdx.x] = mx;
  __syncthreads();
  if ((int)threadIdx.x < J) {
    for (int t = threadIdx.x + J; t < pstep * J; t += J) {
      mn = min_nan(mn, smin[t]);
      mx = max_nan(mx, smax[t]);
    }
    mid[j] = 0.5f * (mn + mx);
    shalf[j] = 0.5f * (mx - mn) * kExp2Scale;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = 0.f;
    bool finite = true;
    for (int q = 0; q < J; ++q) {
      finite = finite && (shalf[q] == shalf[q]) && (shalf[q] < 3.0e38f);
      m = shalf[q] > m ? shalf[q] : m;
    }
    header[1] = m;                                             // max |a|
    reinterpret_cast<int *>(header)[0] = (finite && m * m < kFactMaxSq) ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void prep_build_kernel(const float *__restrict__ Z, const float *__restrict__ mid,
                                                         float2v *__restrict__ rowdat, float2v *__restrict__ coldat,
                                                         long long N, int ldz, int J) {
  const long long total = N * J;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
    const long long n = g / J;
    const int j = (int)(g % J);
    const float a = (Z[n * ldz + j] - mid[j]) * kExp2Scale;
    const float na2 = -(a * a);
    rowdat[g] = float2v{a, fast_exp2(na2)};
    coldat[g] = float2v{2.0f * a, na2};
  }
}

// LDS record of one column: NP = ceil(JT/2) float4 {b2_even, b2_odd, nb2_even, nb2_odd}; stride padded so that
// (stride/4) is odd (conflict-free per-lane ds_read_b128)
template <int JT> struct FactStride {
  static constexpr int np = (JT + 1) / 2;
  static constexpr int v = (np % 2 == 0) ? 4 * np + 4 : 4 * np;
};

// K(i, c) partial sum over JT projections in factorised form; q points at this column's packed LDS record
// [pair p: b2_{2p}, b2_{2p+1}, nb2_{2p}, nb2_{2p+1}] (JT even) or [b2, nb2] (JT == 1)
// ASMADD: the horizontal add of the packed sum as ONE v_add_f32 written in asm.  Left to the compiler, the SLP vectoriser
// packs the two rows' adds of the T = 1 loop into a v_pk_add_f32 behind three v_mov_b32 each (6 moves + 2 packed adds per two
// steps); the asm form measured 2.352 against 2.365-2.388 ms alternating in one process on one box (4 of 4 pairs).
template <int JT, bool ASMADD = false>
__device__ __forceinline__ float fact_pair_sum(const float2v (&ap)[(JT + 1) / 2], const float2v (&ea)[(JT + 1) / 2],
                                               const float *q) {
  if constexpr (JT != 1) {
    // odd JT: the last record's second slot is padding (b2 = 0, nb2 = -1e30 -> e = 0, and Ea = 0 on the row side)
    float2v acc = {0.f, 0.f};
#pragma unroll
    for (int p = 0; p < (JT + 1) / 2; ++p) {
      const float4v c = *reinterpret_cast<const float4v *>(q + 4 * p);
      const float2v b2 = {c.x, c.y};
      const float2v nb2 = {c.z, c.w};
      const float2v t = __builtin_elementwise_fma(ap[p], b2, nb2);
      const float2v e = {fast_exp2(t.x), fast_exp2(t.y)};
      acc = __builtin_elementwise_fma(e, ea[p], acc);
    }
    if constexpr (ASMADD) {
      float s;
      asm("v_add_f32 %0, %1, %2" : "=v"(s) : "v"(acc.x), "v"(acc.y));
      return s;
    } else {
      return acc.x + acc.y;
    }
  } else {
    const float t = __builtin_fmaf(ap[0].x, q[0], q[1]);
    return fast_exp2(t) * ea[0].x;
  }
}

template <int JT, int TT, int R>
__global__ __launch_bounds__(256) void mvm_fact_kernel(const float2v *__restrict__ rowdat,
                                                       const float2v *__restrict__ coldat,
                                                       const float *__restrict__ V, float *__restrict__ slabR,
                                                       float *__restrict__ slabT, int N, int J, int ldv, int j0, int t0,
                                                       int tcnt, int chunk_cols, int rotdir, int accumulate,
                                                       int w0, int rb_first, int slab_row0, int slab_rows) {
  constexpr int BR = 256 * R;
  constexpr int SC = StageCols<TT>::v;
  constexpr int NP = (JT + 1) / 2;
  constexpr int STR = (JT == 1) ? 2 : FactStride<JT>::v;
  __shared__ __attribute__((aligned(16))) float sB[SC * STR];
  __shared__ __attribute__((aligned(16))) float sV[SC * TT];
  __shared__ __attribute__((aligned(16))) float sT[4 * SC * TT];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  int rb, kchunk;
  wg_to_tile(blockIdx.x + w0, N, BR, chunk_cols, true, rb, kchunk);
  const int r0 = rb * BR;
  const long long cb = (long long)r0 + (long long)kchunk * chunk_cols;
  if (cb >= N) return;
  const int c_begin = (int)cb;
  const int c_end = (c_begin + chunk_cols < N) ? c_begin + chunk_cols : N;

  float2v ap[R][NP], ea[R][NP];
  float vrow[R][TT], accR[R][TT];
  // Unconditional loads from clamped addresses, masked by a multiply: a load under a condition — or a select next to the
  // load — compiles to a branch with its own s_waitcnt vmcnt(0), i.e. 20+ SERIAL round trips in this prologue and 10 per
  // stage below (round 3, same-box A/B: T = 1 2.314 -> 2.278 ms, T = 11 block 3.26 -> 3.19 ms, N = 7372 T = 11 125 -> 117 us).
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = r0 + wave * (64 * R) + r * 64 + lane;
    const bool valid = row < N;
    const int rowc = valid ? row : N - 1;
    const float vm = valid ? 1.f : 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const float2v x0 = rowdat[(size_t)rowc * J + j0 + 2 * p];
      float2v x1 = {0.f, 0.f};
      if (2 * p + 1 < JT) x1 = rowdat[(size_t)rowc * J + j0 + 2 * p + 1];
      ap[r][p] = float2v{x0.x, x1.x};
      ea[r][p] = float2v{x0.y * vm, x1.y * vm};     // invalid rows: Ea = 0 -> K = 0
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const int tc = t < tcnt ? t : 0;
      vrow[r][t] = V[(size_t)rowc * ldv + t0 + tc] * ((valid && t < tcnt) ? 1.f : 0.f);
      accR[r][t] = 0.f;
    }
  }

  for (int c0 = c_begin; c0 < c_end; c0 += SC) {
    __syncthreads();
    if (tid < SC) {
      const int col = c0 + tid;
      const bool cv = col < c_end;
      if constexpr (JT == 1) {
        float2v x = {0.f, -1.0e30f};
        if (cv) x = coldat[(size_t)col * J + j0];
        sB[tid * STR + 0] = x.x;
        sB[tid * STR + 1] = x.y;
      } else {
        const int colc = cv ? col : N - 1;
        const float cm = cv ? 1.f : 0.f, padv = cv ? 0.f : -1.0e30f;     // padded columns: b2 = 0, nb2 = -1e30 -> exp2 = 0
        float2v y0[NP], y1[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          y0[p] = coldat[(size_t)colc * J + j0 + 2 * p];
          y1[p] = float2v{0.f, -1.0e30f};
          if (2 * p + 1 < JT) y1[p] = coldat[(size_t)colc * J + j0 + 2 * p + 1];
        }
        float vv[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) vv[t] = V[(size_t)colc * ldv + t0 + (t < tcnt ? t : 0)];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const bool has1 = 2 * p + 1 < JT;
          *reinterpret_cast<float4v *>(&sB[tid * STR + 4 * p]) =
              float4v{y0[p].x * cm, has1 ? y1[p].x * cm : 0.f, __builtin_fmaf(y0[p].y, cm, padv),
                      has1 ? __builtin_fmaf(y1[p].y, cm, padv) : -1.0e30f};
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) sV[tid * TT + t] = vv[t] * ((cv && t < tcnt) ? 1.f : 0.f);
      }
      if constexpr (JT == 1) {
#pragma unroll
        for (int t = 0; t < TT; ++t)
          sV[tid * TT + t] = (cv && t < tcnt) ? V[(size_t)col * ldv + t0 + t] : 0.f;
      }
    }
    __syncthreads();
    const int ncol = c_end - c0;
    const int nsub = ncol >= SC ? SC / 64 : (ncol + 63) / 64;
    for (int sub = 0; sub < nsub; ++sub) {
      const bool doT = (c0 + sub * 64 >= r0 + BR);
      float accT[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) accT[t] = 0.f;
      if (doT) {
#pragma unroll 2
        for (int s = 0; s < 64; ++s) {
          const int idx = sub * 64 + ((lane + rotdir * s) & 63);
          float v[TT];
          lds_load_vec<TT>(sV, idx, v);
          float tsum[TT];
#pragma unroll
          for (int t = 0; t < TT; ++t) tsum[t] = accT[t];
          // (24-bit multiply: v_mul_lo_u32 is a quarter-rate instruction, one per step in this loop)
          const float *colrec = sB + __mul24(idx, STR);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float ks = fact_pair_sum<JT, true>(ap[r], ea[r], colrec);     // (T = 11 block: 3.32 -> 3.29 ms as well)
#pragma unroll
            for (int t = 0; t < TT; ++t) {
              accR[r][t] = __builtin_fmaf(ks, v[t], accR[r][t]);
              tsum[t] = __builtin_fmaf(ks, vrow[r][t], tsum[t]);
            }
          }
#pragma unroll
          for (int t = 0; t < TT; ++t) accT[t] = wave_rotate1(tsum[t]);
        }
      } else {
#pragma unroll 2
        for (int s = 0; s < 64; ++s) {
          const int idx = sub * 64 + ((lane + s) & 63);
          float v[TT];
          lds_load_vec<TT>(sV, idx, v);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float ks = fact_pair_sum<JT>(ap[r], ea[r], sB + idx * STR);
#pragma unroll
            for (int t = 0; t < TT; ++t) accR[r][t] = __builtin_fmaf(ks, v[t], accR[r][t]);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) sT[(wave * SC + sub * 64 + lane) * TT + t] = accT[t];
    }
    __syncthreads();
    {
      const int col = c0 + tid;
      if (tid < SC && col < c_end && col >= r0 + BR) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          if (t < tcnt) {
            float sum = sT[(0 * SC + tid) * TT + t] + sT[(1 * SC + tid) * TT + t] +
                        sT[(2 * SC + tid) * TT + t] + sT[(3 * SC + tid) * TT + t];
            float *dst = slabT + ((size_t)(rb - rb_first) * N + col) * ldv + t0 + t;
            *dst = accumulate ? *dst + sum : sum;
          }
        }
      }
    }
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = r0 + wave * (64 * R) + r * 64 + lane;
    if (row < N) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (t < tcnt) {
          float *dst = slabR + ((size_t)kchunk * slab_rows + (row - slab_row0)) * ldv + t0 + t;
          *dst = accumulate ? *dst + accR[r][t] : accR[r][t];
        }
      }
    }
  }
}

// out[row][t] = scale * (sum_k slabR[k][row][t] + sum_{rb < row/BR} slabT[rb][row][t]) + noise * V[row][t]
// A workgroup owns 32 consecutive outputs; its 8 half-waves split the slab entries (entry k goes to group k % 8, each
// read is a 128-byte contiguous segment) and the 8 group sums are added in a fixed order: deterministic, and short
// enough for the small problems (N < 10k) where a one-thread-per-output loop over ~150 slabs was latency-bound.
template <int kRedOutputs>
__global__ __launch_bounds__(256) void mvm_reduce_kernel(const float *__restrict__ slabR, const float *__restrict__ slabT,
                                  const float *__restrict__ V, float *__restrict__ out, int M, int N, int T,
                                  int BR, int chunk_cols, int sym, float scale, float noise,
                                  const int *__restrict__ guard, int rb0, int rb1, int slab_row0, int slab_rows,
                                  rpgp_internal::Taper taper = rpgp_internal::Taper{0x7fffffff, 0x7fffffff, 0x7fffffff}) {
  constexpr int kRedGroups = 256 / kRedOutputs;
  __shared__ double sacc[kRedGroups][kRedOutputs];   // float64: ~150 slab entries per output, free at this size
  const int o = threadIdx.x & (kRedOutputs - 1), g = threadIdx.x / kRedOutputs;
  const size_t gid = (size_t)blockIdx.x * kRedOutputs + o;
  const bool valid = gid < (size_t)M * T;
  double acc = 0.0;
  if (valid) {
    const int row = (int)(gid / T);
    const int rb = row / BR;
    const int cbase = sym ? rb * BR : 0;
    const int chunk_b = rpgp_internal::taper_chunk(rb, chunk_cols, taper.tb1, taper.tb2, taper.tb3);
    const int nk = (N - cbase + chunk_b - 1) / chunk_b;
    if (rb >= rb0 && rb < rb1) {                    // row products exist only for this call's row blocks
      const size_t lid = gid - (size_t)slab_row0 * T;
      for (int k = g; k < nk; k += kRedGroups) acc += (double)slabR[(size_t)k * slab_rows * T + lid];
    }
    if (sym) {
      const int bend = rb < rb1 ? rb : rb1;         // transposed products written by row blocks rb0 <= b < min(rb, rb1)
      int b = rb0 + g;                              // four entries in flight, added in the same order as the plain loop
      for (; b + 3 * kRedGroups < bend; b += 4 * kRedGroups) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = slabT[(size_t)(b + u * kRedGroups - rb0) * N * T + gid];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (double)v[u];
      }
      for (; b < bend; b += kRedGroups) acc += (double)slabT[(size_t)(b - rb0) * N * T + gid];
    }
  }
  sacc[g][o] = acc;
  __syncthreads();
  if (g != 0 || !valid) return;
  if (guard && *guard == 0) {   // prepared (factorised) path used although rpgp_prepare flagged the range as unsafe
    out[gid] = __builtin_nanf("");
    return;
  }
  double tot = sacc[0][o];
#pragma unroll
  for (int q = 1; q < kRedGroups; ++q) tot += sacc[q][o];
  float r = (float)((double)scale * tot);
  if (noise != 0.f) r = __builtin_fmaf(noise, V[gid], r);
  out[gid] = r;
}

// Outputs per workgroup of the slab reduce: 32 (eight groups share the slab entries: the latency-bound small problems) up to
// 256 (one group, 1 KB contiguous per slab entry: the large ones, where 128-byte pieces 2 MB apart waste the HBM rate).
inline int red_outputs(size_t total) {
  // measured (tools/r5_symk_red_prof.sh, T = 11 packed-cache product): 64 outputs beat 32 at every size measured (N = 7k,
  // 81k outputs: 7.2 -> 6.5 us; 15k: 10.4 -> 9.4; 50k: 36.6 -> 29.6); 128 only from N ~ 15k (8.3 us) and lose at 7k (10.0 us);
  // below the measured range the round-2 width stays
  return total >= (size_t)65536 ? 64 : 32;
}
#define RPGP_LAUNCH_MVM_REDUCE(total_, st_, ...)                                                                          \
  do {                                                                                                                    \
    const size_t tot__ = (total_);                                                                                        \
    switch (red_outputs(tot__)) {                                                                                         \
      case 64: hipLaunchKernelGGL(mvm_reduce_kernel<64>, dim3((unsigned)((tot__ + 63) / 64)), dim3(256), 0, st_, __VA_ARGS__); break;    \
      default: hipLaunchKernelGGL(mvm_reduce_kernel<32>, dim3((unsigned)((tot__ + 31) / 32)), dim3(256), 0, st_, __VA_ARGS__); break;    \
    }                                                                                                                     \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Packed symmetric cache ("symcache"): the cached-K mode without the lower triangle.
//   The fused symmetric sweep evaluates every unordered pair once and uses the value twice (row product and, through
//   the rotating accumulators, the transposed product).  The cache stores exactly the values that sweep consumes, in
//   the order it consumes them, so that the cached product is the same sweep with a 16-byte load in place of the J
//   exponentials: HALF the bytes of the N x N matrix per product (and half the HBM footprint).
//   Unit of storage: one 64-column subtile of one row block = BR x 64 floats, laid out
//       [wave 0..3][s4 0..15][r 0..R-1][lane 0..63][u 0..3]      (float4 per lane: rotation steps s = 4 s4 + u)
//   value (wave, s4, r, lane, u) = sum_j exp2(-(a - b)^2) for row r0 + wave*64R + r*64 + lane and column
//   csub + ((lane + rotdir*s) & 63).  Subtiles are numbered row block by row block, left to right from the diagonal
//   block's first column (subtile g of row block rb: columns rb*BR + 64 g ...), so the layout depends on (N, R) only.
//   A wave's loads are 1 KB contiguous; it streams 16 R KB per subtile.  Pairs outside the matrix hold 0.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ long long symk_first_subtile(int rb, int N, int BR) {
  const long long nsN = (N + 63) / 64, q = BR / 64;
  return (long long)rb * nsN - q * ((long long)rb * (rb - 1) / 2);
}

// JT scaled coordinates of point `n` (clamped to a valid row), or `pad` in every slot when !ok — ALWAYS loaded: a load under a
// condition (or a select next to it) compiles to a branch with its own s_waitcnt vmcnt(0), one serial round trip per element.
// The masking is arithmetic (z * m + padv) so that the optimiser cannot turn it back into a branch.
template <int JT>
__device__ __forceinline__ void load_coords_masked(const float *__restrict__ Z, int n, int N, int ldz, int j0, bool ok,
                                                   float pad, float (&out)[JT]) {
  const int nc = (ok && n < N) ? n : N - 1;
  const float m = ok ? kExp2Scale : 0.f, padv = ok ? 0.f : pad;
  float z[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) z[j] = Z[(size_t)nc * ldz + j0 + j];
#pragma unroll
  for (int j = 0; j < JT; ++j) out[j] = __builtin_fmaf(z[j], m, padv);
}

template <int JT, int R>
__global__ __launch_bounds__(256) void symk_build_kernel(const float *__restrict__ Z, float4v *__restrict__ cache, int N,
                                                         int ldz, int j0, int chunk_cols, int rotdir, int accumulate,
                                                         int w0, long long sub0) {
  constexpr int BR = 256 * R;
  constexpr int STR = ColStride<JT>::v;
  __shared__ __attribute__((aligned(16))) float sB[64 * STR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int rb, kchunk;
  wg_to_tile(blockIdx.x + w0, N, BR, chunk_cols, true, rb, kchunk);
  const int r0 = rb * BR;
  const long long cb = (long long)r0 + (long long)kchunk * chunk_cols;
  if (cb >= N) return;
  const int c_begin = (int)cb;
  const int c_end = (c_begin + chunk_cols < N) ? c_begin + chunk_cols : N;
  // rows / columns outside the matrix sit at +3e18 / +1e18: every pair with one of them is exp2(-huge) = 0
  float a[R][JT];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = r0 + wave * (64 * R) + r * 64 + lane;
    load_coords_masked<JT>(Z, row, N, ldz, j0, row < N, 3.0e18f, a[r]);
  }
  const long long g0 = symk_first_subtile(rb, N, BR) + (long long)kchunk * (chunk_cols / 64) - sub0;
  int sub = 0;
  for (int c0 = c_begin; c0 < c_end; c0 += 64, ++sub) {
    __syncthreads();
    if (tid < 64) {
      const int col = c0 + tid;
      float bc[JT];
      load_coords_masked<JT>(Z, col, N, ldz, j0, col < c_end, 1.0e18f, bc);
#pragma unroll
      for (int j = 0; j < JT; ++j) sB[tid * STR + j] = bc[j];
    }
    __syncthreads();
    float4v *dst = cache + ((size_t)((g0 + sub) * 4 + wave) * 16) * R * 64 + lane;
    for (int s4 = 0; s4 < 16; ++s4) {
      float4v kq[R];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = (lane + rotdir * (4 * s4 + u)) & 63;
        float b[JT];
        lds_load_cols<JT>(sB, idx, b);
#pragma unroll
        for (int r = 0; r < R; ++r) kq[r][u] = pair_kernel_sum<JT>(a[r], b);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float4v *q = dst + (size_t)(s4 * R + r) * 64;
        if (accumulate) {
          const float4v old = *q;
          kq[r] += old;
        }
        *q = kq[r];
      }
    }
  }
}

// Cached symmetric product: the loop nest of mvm_fact_kernel with the kernel value taken from the cache.  The wave's
// stream is requested D steps (of 4 rotation steps x R rows = R KB) ahead through a register ring, across subtile and
// LDS-stage boundaries.
// Loads of the packed cache: nontemporal for a cache that is streamed once per product; DEFAULT policy for one that fits the
// 256 MB Infinity Cache, where the products of one solve re-read it back to back (symk_nt_loads below; measured
// tools/r5_symk_nt_ab.py, T = 11: 39 MB 24.6 -> 21.9 us, 118 MB 37.7 -> 35.4, 234 MB 59.9 -> 52.4, thin T = 1 46.1 -> 41.4; the other
// way from ~300 MB: 464 MB 106 -> 113 us, 5 GB 947 -> 1039).
template <bool NT>
__device__ __forceinline__ float4v symk_tile_load(const float4v *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

template <int TT, int R, bool NTL = true>
__global__ __launch_bounds__(256) void symk_mvm_kernel(const float4v *__restrict__ cache, const float *__restrict__ V,
                                                       float *__restrict__ slabR, float *__restrict__ slabT, int N, int ldv,
                                                       int t0, int tcnt, int chunk_cols, int rotdir, int accumulate, int w0,
                                                       int rb_first, int slab_row0, int slab_rows, long long sub0) {
  constexpr int BR = 256 * R;
  constexpr int SC = StageCols<TT>::v;
  constexpr int D = (TT > 4) ? 8 : 4;               // ring depth (steps): the wide blocks run 2 waves per SIMD and need the distance
  __shared__ __attribute__((aligned(16))) float sV[SC * TT];
  __shared__ __attribute__((aligned(16))) float sT[4 * SC * TT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int rb, kchunk;
  wg_to_tile(blockIdx.x + w0, N, BR, chunk_cols, true, rb, kchunk);
  const int r0 = rb * BR;
  const long long cb = (long long)r0 + (long long)kchunk * chunk_cols;
  if (cb >= N) return;
  const int c_begin = (int)cb;
  const int c_end = (c_begin + chunk_cols < N) ? c_begin + chunk_cols : N;

  float vrow[R][TT], accR[R][TT];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = r0 + wave * (64 * R) + r * 64 + lane;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      vrow[r][t] = (row < N && t < tcnt) ? V[(size_t)row * ldv + t0 + t] : 0.f;
      accR[r][t] = 0.f;
    }
  }
  const long long g0 = symk_first_subtile(rb, N, BR) + (long long)kchunk * (chunk_cols / 64) - sub0;
  const int total = ((c_end - c_begin + 63) / 64) * 16;                 // steps of this workgroup
  const float4v *wp = cache + ((size_t)(g0 * 4 + wave) * 16) * R * 64 + lane;
  // step n = 16 * subtile + s4  ->  wp + subtile * (4 waves * 16 * R * 64) + s4 * R * 64
  auto step_ptr = [&](int n) {
    n = n < total ? n : total - 1;                                      // tail: re-request the last step (no branch)
    return wp + (size_t)(n >> 4) * (size_t)(4 * 16 * R * 64) + (size_t)(n & 15) * (R * 64);
  };
  float4v ring[D][R];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float4v *q = step_ptr(d);
#pragma unroll
    for (int r = 0; r < R; ++r) ring[d][r] = symk_tile_load<NTL>(q + r * 64);
  }
  int n = 0;
  // the stage's right-hand sides are requested one stage ahead, from clamped addresses (a load under a condition compiles to
  // a branch with `s_waitcnt vmcnt(0)` behind it, which drains the ring at the head of every stage)
  float vpre[TT];
  auto v_request = [&](int c0n) {
    const int col = c0n + (tid < SC ? tid : 0);
    const int colc = col < N ? col : N - 1;
#pragma unroll
    for (int t = 0; t < TT; ++t) vpre[t] = V[(size_t)colc * ldv + t0 + (t < tcnt ? t : 0)];
  };
  v_request(c_begin);
  for (int c0 = c_begin; c0 < c_end; c0 += SC) {
    __syncthreads();
    if (tid < SC) {
#pragma unroll
      for (int t = 0; t < TT; ++t) sV[tid * TT + t] = (c0 + tid < c_end && t < tcnt) ? vpre[t] : 0.f;
    }
    v_request(c0 + SC < c_end ? c0 + SC : c0);
    __syncthreads();
    const int ncol = c_end - c0;
    const int nsub = ncol >= SC ? SC / 64 : (ncol + 63) / 64;
    for (int sub = 0; sub < nsub; ++sub) {
      const bool doT = (c0 + sub * 64 >= r0 + BR);
      float accT[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) accT[t] = 0.f;
#pragma nounroll
      for (int g = 0; g < 16 / D; ++g) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
          float4v cur[R];
#pragma unroll
          for (int r = 0; r < R; ++r) cur[r] = ring[k][r];
          {
            const float4v *q = step_ptr(n + D);
#pragma unroll
            for (int r = 0; r < R; ++r) ring[k][r] = symk_tile_load<NTL>(q + r * 64);
          }
          ++n;
          if (doT) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int idx = sub * 64 + ((lane + rotdir * (4 * (g * D + k) + u)) & 63);
              float v[TT];
              lds_load_vec<TT>(sV, idx, v);
              if constexpr (TT % 2 == 0) {
                // both products on float2 so that hipcc emits v_pk_fma_f32 for the transposed sums as well (left to
                // the SLP vectoriser, the chains through the DPP rotation stayed scalar: 96 of 192 FMAs per step)
                float2v ts2[TT / 2];
#pragma unroll
                for (int q = 0; q < TT / 2; ++q) ts2[q] = float2v{accT[2 * q], accT[2 * q + 1]};
#pragma unroll
                for (int r = 0; r < R; ++r) {
                  const float2v ks2 = {cur[r][u], cur[r][u]};
#pragma unroll
                  for (int q = 0; q < TT / 2; ++q) {
                    const float2v a2 = __builtin_elementwise_fma(ks2, float2v{v[2 * q], v[2 * q + 1]},
                                                                 float2v{accR[r][2 * q], accR[r][2 * q + 1]});
                    accR[r][2 * q] = a2.x;
                    accR[r][2 * q + 1] = a2.y;
                    ts2[q] = __builtin_elementwise_fma(ks2, float2v{vrow[r][2 * q], vrow[r][2 * q + 1]}, ts2[q]);
                  }
                }
#pragma unroll
                for (int q = 0; q < TT / 2; ++q) {
                  accT[2 * q] = wave_rotate1(ts2[q].x);
                  accT[2 * q + 1] = wave_rotate1(ts2[q].y);
                }
              } else {
                float tsum[TT];
#pragma unroll
                for (int t = 0; t < TT; ++t) tsum[t] = accT[t];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                  const float ks = cur[r][u];
#pragma unroll
                  for (int t = 0; t < TT; ++t) {
                    accR[r][t] = __builtin_fmaf(ks, v[t], accR[r][t]);
                    tsum[t] = __builtin_fmaf(ks, vrow[r][t], tsum[t]);
                  }
                }
#pragma unroll
                for (int t = 0; t < TT; ++t) accT[t] = wave_rotate1(tsum[t]);
              }
            }
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int idx = sub * 64 + ((lane + rotdir * (4 * (g * D + k) + u)) & 63);
              float v[TT];
              lds_load_vec<TT>(sV, idx, v);
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const float ks = cur[r][u];
#pragma unroll
                for (int t = 0; t < TT; ++t) accR[r][t] = __builtin_fmaf(ks, v[t], accR[r][t]);
              }
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) sT[(wave * SC + sub * 64 + lane) * TT + t] = accT[t];
    }
    __syncthreads();
    {
      const int col = c0 + tid;
      if (tid < SC && col < c_end && col >= r0 + BR) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          if (t < tcnt) {
            const float sum = sT[(0 * SC + tid) * TT + t] + sT[(1 * SC + tid) * TT + t] +
                              sT[(2 * SC + tid) * TT + t] + sT[(3 * SC + tid) * TT + t];
            float *dst = slabT + ((size_t)(rb - rb_first) * N + col) * ldv + t0 + t;
            *dst = accumulate ? *dst + sum : sum;
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = r0 + wave * (64 * R) + r * 64 + lane;
    if (row < N) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (t < tcnt) {
          float *dst = slabR + ((size_t)kchunk * slab_rows + (row - slab_row0)) * ldv + t0 + t;
          *dst = accumulate ? *dst + accR[r][t] : accR[r][t];
        }
      }
    }
  }
}

// ---- wide layout of the packed cache (blocks of 5..16 right-hand sides on the matrix cores) -----------------------------
// Same subtiles, same size; inside a wave's 16 R KB the values are stored as 16 x 16 tiles in the A/B operand order of
// v_mfma_f32_16x16x4_f32:  [wave][slot(rt 0..4R-1, ct 0..3)][lane]  float4 = K[row0 + 16 rt + (l % 16)][csub + 16 ct + 4 (l / 16) + i]
// (slots in the pair order the product consumes: (2m, p), (2m + 1, (p + 1) % 4), p = 0..3).
// One tile feeds two groups of four exact-fp32 MFMAs:
//   row product        D[t][rho]   += sum_i  A = V[csub + 16 ct + 4k + i][t]   x  B = tile_i             (contracts the columns)
//   transposed product D'[t][gam]  += sum_j  A = V[row0 + 16 rt + 4k + j][t]   x  B = L2_j               (contracts the rows)
// where L2_j[lane] = K[16 rt + 4 (l/16) + j][16 ct + l % 16] is the tile transposed through a per-wave LDS scratch (one
// 16-byte store, four 4-byte loads per lane, conflict-free at row stride 20; a wave's DS operations execute in order, so no
// barrier).  (First version: transposed in registers by four more MFMAs against identity slices — exact, but 12 instead
// of 8 matrix-pipe issues per tile: 1.14 ms against 1.00 ms for the T = 11 block at N = 50k.)
// The rotating accumulators of the thin layout cost T DPP moves per 64 pairs; here the transposed sums of a column tile
// stay in one accumulator across the wave's 4 R row tiles.
template <int JT, int R>
__global__ __launch_bounds__(256) void symk_build_tile_kernel(const float *__restrict__ Z, float4v *__restrict__ cache,
                                                              int N, int ldz, int j0, int chunk_cols, int accumulate,
                                                              int w0, long long sub0) {
  constexpr int BR = 256 * R;
  constexpr int STR = ColStride<JT>::v;
  __shared__ __attribute__((aligned(16))) float sB[64 * STR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rho = lane & 15, kap = lane >> 4;
  int rb, kchunk;
  wg_to_tile(blockIdx.x + w0, N, BR, chunk_cols, true, rb, kchunk);
  const int r0 = rb * BR;
  const long long cb = (long long)r0 + (long long)kchunk * chunk_cols;
  if (cb >= N) return;
  const int c_begin = (int)cb;
  const int c_end = (c_begin + chunk_cols < N) ? c_begin + chunk_cols : N;
  const long long g0 = symk_first_subtile(rb, N, BR) + (long long)kchunk * (chunk_cols / 64) - sub0;
  int sub = 0;
  for (int c0 = c_begin; c0 < c_end; c0 += 64, ++sub) {
    __syncthreads();
    if (tid < 64) {
      const int col = c0 + tid;
      float bc[JT];
      load_coords_masked<JT>(Z, col, N, ldz, j0, col < c_end, 1.0e18f, bc);
#pragma unroll
      for (int j = 0; j < JT; ++j) sB[tid * STR + j] = bc[j];
    }
    __syncthreads();
    float4v *dst = cache + ((size_t)((g0 + sub) * 4 + wave) * 16) * R * 64 + lane;
#pragma nounroll
    for (int rt = 0; rt < 4 * R; ++rt) {
      const int row = r0 + wave * (64 * R) + 16 * rt + rho;
      float a[JT];
      load_coords_masked<JT>(Z, row, N, ldz, j0, row < N, 3.0e18f, a);
#pragma nounroll
      for (int ct = 0; ct < 4; ++ct) {
        float4v kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float b[JT];
          lds_load_cols<JT>(sB, 16 * ct + 4 * kap + i, b);
          kq[i] = pair_kernel_sum<JT>(a, b);
        }
        // slot of tile (rt, ct) in the product's pair order: pairs A = (2 m, p), B = (2 m + 1, (p + 1) % 4)
        const int slot = (rt >> 1) * 8 + (((rt & 1) ? ((ct + 3) & 3) : ct) << 1) + (rt & 1);
        float4v *q = dst + (size_t)slot * 64;
        if (accumulate) {
          const float4v old = *q;
          kq += old;
        }
        *q = kq;
      }
    }
  }
}

// One pair of row tiles (2 RT2, 2 RT2 + 1) of a subtile, then the next pair (compile-time recursion: a `#pragma unroll`
// loop over the pairs was kept rolled by hipcc, which then rotated the accumulator arrays through registers with ~1000
// v_mov per subtile).
template <int R, bool DOT, int RT2, bool NT = true>
__device__ __forceinline__ void symk_tile_rows(const float4v *wp, float4v (&ring)[8], const float (&acol)[4][4],
                                               const float (&arow)[4 * R][4], float *scr, int woff, int roff,
                                               floatx4m (&accR)[4 * R], floatx4m (&accT)[4]) {
  constexpr int NRT = 4 * R, NTL = 4 * NRT, D = 8;
  constexpr size_t SUB = (size_t)4 * NTL * 64;
  if constexpr (RT2 < NRT / 2) {
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int rtA = 2 * RT2, ctA = pp, rtB = 2 * RT2 + 1, ctB = (pp + 1) & 3;
      const float4v kA = ring[2 * pp], kB = ring[2 * pp + 1];
      {
        const int m0 = RT2 * 8 + 2 * pp + D, m1 = m0 + 1;            // slots requested now (>= NT: next subtile)
        ring[2 * pp] = symk_tile_load<NT>(wp + (m0 < NTL ? (size_t)m0 * 64 : SUB + (size_t)(m0 - NTL) * 64));
        ring[2 * pp + 1] = symk_tile_load<NT>(wp + (m1 < NTL ? (size_t)m1 * 64 : SUB + (size_t)(m1 - NTL) * 64));
      }
      __builtin_amdgcn_sched_barrier(0);       // keep the requests D tiles ahead (hipcc sinks them to their first use)
      if constexpr (DOT) {
        // transpose of the two tiles through the wave's LDS scratch: row-major 16 x 20 (16-byte stores and the 4-byte
        // loads below are conflict-free at that stride); a wave's DS operations execute in order, so neither a barrier nor
        // a second buffer is needed; the eight row-product MFMAs cover the round trip
        *reinterpret_cast<float4v *>(scr + woff) = kA;
        *reinterpret_cast<float4v *>(scr + 320 + woff) = kB;
        floatx4m lA, lB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          lA[j] = scr[roff + 20 * j];
          lB[j] = scr[320 + roff + 20 * j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          accR[rtA] = __builtin_amdgcn_mfma_f32_16x16x4f32(acol[ctA][i], kA[i], accR[rtA], 0, 0, 0);
          accR[rtB] = __builtin_amdgcn_mfma_f32_16x16x4f32(acol[ctB][i], kB[i], accR[rtB], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          accT[ctA] = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[rtA][j], lA[j], accT[ctA], 0, 0, 0);
          accT[ctB] = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[rtB][j], lB[j], accT[ctB], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          accR[rtA] = __builtin_amdgcn_mfma_f32_16x16x4f32(acol[ctA][i], kA[i], accR[rtA], 0, 0, 0);
          accR[rtB] = __builtin_amdgcn_mfma_f32_16x16x4f32(acol[ctB][i], kB[i], accR[rtB], 0, 0, 0);
        }
      }
    }
    symk_tile_rows<R, DOT, RT2 + 1, NT>(wp, ring, acol, arow, scr, woff, roff, accR, accT);
  }
}

// The wide-layout product: ONE workgroup barrier per 64-column subtile (round 5; the three-barrier first form it replaced is
// tools/experiments/r6_removed_forms.patch).
//  * The staged right-hand sides and the four waves' transposed sums live in DOUBLE-BUFFERED LDS images.  Subtile s reads its
//    A operands from sV[s & 1], requests the block of subtile s + 1 at its head (four registers per thread) and writes it to
//    sV[(s + 1) & 1] at its end; the waves' transposed sums go to sT[s & 1]; ONE barrier; then the fixed-order cross-wave
//    sum and the slab store (bit-identical results to the three-barrier form).
//  * Why one barrier is enough: an image written in subtile s was last read in subtile s - 1 (sV) or s - 2 (sT), and every
//    reader of those passed the barrier of subtile s - 1 only after its reads — which the writer has passed too.
//  (A first form without any staging — every wave requesting its own sixteen A operands from global memory one subtile
//  ahead — needs 16 more registers than the 256 there are: the reloads put an `s_waitcnt vmcnt(0)` at the head of every
//  subtile, which drains the tile ring.)
template <int R, bool NTLOAD = true>
__global__ __launch_bounds__(256, 2) void symk_mvm_tile2_kernel(const float4v *__restrict__ cache, const float *__restrict__ V,
                                                             float *__restrict__ slabR, float *__restrict__ slabT, int N,
                                                             int ldv, int t0, int tcnt, int chunk_cols, int w0, int rb_first,
                                                             int slab_row0, int slab_rows, long long sub0) {
  constexpr int BR = 256 * R;
  constexpr int NRT = 4 * R;
  constexpr int NT = 4 * NRT;
  constexpr int D = 8;
  constexpr int SVS = 17;
  __shared__ __attribute__((aligned(16))) float sV[2][64 * SVS];
  __shared__ __attribute__((aligned(16))) float sT[2][4 * 64 * 16];
  __shared__ __attribute__((aligned(16))) float sX[4 * 640];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tn = lane & 15, kap = lane >> 4;
  int rb, kchunk;
  wg_to_tile(blockIdx.x + w0, N, BR, chunk_cols, true, rb, kchunk);
  const int r0 = rb * BR;
  const long long cb = (long long)r0 + (long long)kchunk * chunk_cols;
  if (cb >= N) return;
  const int c_begin = (int)cb;
  const int c_end = (c_begin + chunk_cols < N) ? c_begin + chunk_cols : N;
  const int rw0 = r0 + wave * (64 * R);

  float arow[NRT][4];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = rw0 + 16 * rt + 4 * kap + j;
      arow[rt][j] = (row < N && tn < tcnt) ? V[(size_t)row * ldv + t0 + tn] : 0.f;
    }
  float *scr = sX + wave * 640;
  const int woff = tn * 20 + 4 * kap;
  const int roff = 4 * kap * 20 + tn;
  floatx4m accR[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) accR[rt] = floatx4m{0.f, 0.f, 0.f, 0.f};

  const long long g0 = symk_first_subtile(rb, N, BR) + (long long)kchunk * (chunk_cols / 64) - sub0;
  const float4v *wp = cache + ((size_t)(g0 * 4 + wave) * 16) * R * 64 + lane;
  constexpr size_t SUB = (size_t)4 * NT * 64;
  float4v ring[D];
#pragma unroll
  for (int d = 0; d < D; ++d) ring[d] = symk_tile_load<NTLOAD>(wp + (size_t)d * 64);
  float vpre[4];
  auto v_request = [&](int c0n) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + 256 * it;
      const int c = e >> 4, t = e & 15;
      const int col = c0n + c;
      const int colc = col < N ? col : N - 1;
      const int tc = t < tcnt ? t : 0;
      vpre[it] = V[(size_t)colc * ldv + t0 + tc];
    }
  };
  auto v_stage = [&](float *dst, int c0n) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + 256 * it;
      const int c = e >> 4, t = e & 15;
      dst[c * SVS + t] = (c0n + c < c_end && t < tcnt) ? vpre[it] : 0.f;
    }
  };
  // contiguous slab stores: element e = c * tcnt + t of a subtile's block (when the block is one run: t0 == 0, tcnt == ldv)
  const bool dense_slab = (t0 == 0 && tcnt == ldv);
  int lidx[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = tid + 256 * it;
    const int c = e / tcnt, t = e - c * tcnt;
    lidx[it] = (e < 64 * tcnt) ? c * 16 + t : -1;
  }
  v_request(c_begin);
  v_stage(sV[0], c_begin);
  __syncthreads();
  int par = 0;
  for (int c0 = c_begin; c0 < c_end; c0 += 64, wp += SUB, par ^= 1) {
    float acol[4][4];
    {
      const float *sv = sV[par];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) acol[ct][i] = sv[(16 * ct + 4 * kap + i) * SVS + tn];
    }
    // The next subtile's block is requested here and written to LDS at the END of this subtile: in straight-line code hipcc
    // counts the 16 R tile requests issued in between and waits with `vmcnt(8)` — the ring stays in flight.  (Consumed at the
    // head of the next iteration the wait becomes `vmcnt(3..0)`: across the back edge the counter model is conservative, and
    // that drains the ring once per subtile — the three-barrier form above does exactly that.)
    v_request(c0 + 64 < c_end ? c0 + 64 : c0);
    const bool doT = (c0 >= r0 + BR);
    floatx4m accT[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) accT[ct] = floatx4m{0.f, 0.f, 0.f, 0.f};
    float *st = sT[par];
    // ONE instruction stream for every subtile: the diagonal block's subtiles (no transposed product: the block is stored
    // whole) run the transposed MFMAs too and drop their sums — BR / N of the matrix work (1 % at N = 50k, 7 % at 7k) on a
    // pipe that is not the bound, against two copies of the tile code whose register maps hipcc reconciles with moves of the
    // whole ring (and an `s_waitcnt vmcnt(0)`) on the loop's back edge.
    symk_tile_rows<R, true, 0, NTLOAD>(wp, ring, acol, arow, scr, woff, roff, accR, accT);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[(wave * 64 + 16 * ct + tn) * 16 + 4 * kap + r] = accT[ct][r];
    v_stage(sV[par ^ 1], c0 + 64);                       // (past the chunk: zeros, never read)
    __syncthreads();
    if (doT) {
      if (dense_slab) {
        // the subtile's 64 x tcnt sums are one contiguous run of the slab: thread e writes element e (whole 256-byte
        // wave stores instead of 176-byte fragments with five idle lanes in sixteen)
        float *dstT = slabT + ((size_t)(rb - rb_first) * N + c0) * ldv;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int e = tid + 256 * it;
          const int li = lidx[it];
          if (li >= 0 && c0 + (li >> 4) < c_end) {
            const float sum = st[0 * 1024 + li] + st[1 * 1024 + li] + st[2 * 1024 + li] + st[3 * 1024 + li];
            dstT[e] = sum;
          }
        }
      } else {
      for (int e = tid; e < 64 * 16; e += 256) {
        const int c = e >> 4, t = e & 15;
        const int col = c0 + c;
        if (col < c_end && t < tcnt) {
          const float sum = st[(0 * 64 + c) * 16 + t] + st[(1 * 64 + c) * 16 + t] + st[(2 * 64 + c) * 16 + t] +
                            st[(3 * 64 + c) * 16 + t];
          slabT[((size_t)(rb - rb_first) * N + col) * ldv + t0 + t] = sum;
        }
      }
      }
    }
  }
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    const int row = rw0 + 16 * rt + tn;
    if (row < N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = 4 * kap + r;
        if (t < tcnt) slabR[((size_t)kchunk * slab_rows + (row - slab_row0)) * ldv + t0 + t] = accR[rt][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Dense block: out[m][n] = scale * sum_j exp(-0.5 (Z1[m,j]-Z2[n,j])^2).  One thread per output column
// (coalesced stores), RT rows per workgroup held in LDS.
// ---------------------------------------------------------------------------------------------
template <int JT, class KF = KfBase>
__global__ __launch_bounds__(256) void dense_kernel(const float *__restrict__ Z1, const float *__restrict__ Z2,
                                                    float *__restrict__ out, int M, int N, int ldz1, int ldz2,
                                                    long long ldo, int j0, float scale, int accumulate,
                                                    const float *__restrict__ wts) {
  constexpr int RT = 16;
  __shared__ float sA[RT][JT];
  float wreg[KF::weighted ? JT / KF::group : 1];
  if constexpr (KF::weighted) {
#pragma unroll
    for (int c = 0; c < JT / KF::group; ++c) wreg[c] = wts[j0 / KF::group + c];
  } else {
    wreg[0] = 0.f;
  }
  const int tid = threadIdx.x;
  const int col = blockIdx.x * 256 + tid;
  const int m0 = blockIdx.y * RT;
  for (int e = tid; e < RT * JT; e += 256) {
    const int r = e / JT, j = e % JT;
    sA[r][j] = (m0 + r < M) ? Z1[(size_t)(m0 + r) * ldz1 + j0 + j] * KF::pre : 0.f;
  }
  __syncthreads();
  if (col >= N) return;
  float b[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) b[j] = Z2[(size_t)col * ldz2 + j0 + j] * KF::pre;
#pragma unroll 4
  for (int r = 0; r < RT; ++r) {
    if (m0 + r >= M) break;
    const float acc = KF::template pair_sum<JT>(sA[r], b, wreg);
    float *dst = out + (size_t)(m0 + r) * ldo + col;
    *dst = accumulate ? __builtin_fmaf(scale, acc, *dst) : scale * acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Projection Z = X @ Peff and its backward dPeff = X^T @ G (thin GEMMs; plain VALU version).
// ---------------------------------------------------------------------------------------------
// Z = X @ Peff on the matrix cores: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, MI355X_MICROARCH.md "FP32-input
// MFMA").  One wave64 owns a 16-row stripe of Z and walks the column tiles; the K dimension (d) advances 4 per MFMA.
//   A (16x4): lane l holds X[row0 + l%16][k0 + l/16]        B (4x16): lane l holds Peff[k0 + l/16][col0 + l%16]
//   D (16x16): lane l holds Z[row0 + 4*(l/16) + r][col0 + l%16], r = 0..3
__global__ __launch_bounds__(256) void project_kernel(const float *__restrict__ X, const float *__restrict__ Peff,
                                                      float *__restrict__ Z, long long N, int d, int J) {
  extern __shared__ float sP[];  // d4 x J16 zero-padded copy of Peff (d4 = d rounded up to 4, J16 = J rounded up to 16)
  const int d4 = (d + 3) & ~3, J16 = (J + 15) & ~15;
  for (int e = threadIdx.x; e < d4 * J16; e += 256) {
    const int k = e / J16, j = e % J16;
    sP[e] = (k < d && j < J) ? Peff[k * J + j] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  const long long stripes = (N + 15) / 16;
  for (long long sidx = (long long)blockIdx.x * 4 + wave; sidx < stripes; sidx += (long long)gridDim.x * 4) {
    const long long row = sidx * 16 + m;
    const bool rv = row < N;
    for (int col0 = 0; col0 < J16; col0 += 16) {
      floatx4m acc = {0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < d4; k0 += 4) {
        const int k = k0 + q;
        const float a = (rv && k < d) ? X[row * d + k] : 0.f;
        const float b = sP[k * J16 + col0 + m];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long orow = sidx * 16 + 4 * q + r;
        const int ocol = col0 + m;
        if (orow < N && ocol < J) Z[orow * J + ocol] = acc[r];
      }
    }
  }
}

// dPeff[k][j] = sum_n X[n][k] G[n][j]; one workgroup per (k, j-tile) would be overkill: d*J <= ~1024 outputs,
// each workgroup reduces a slice of n for all (k,j) handled by its threads, then atomics-free two-pass.
__global__ __launch_bounds__(256) void project_grad_partial_kernel(const float *__restrict__ X,
                                                                   const float *__restrict__ G,
                                                                   float *__restrict__ part, long long N, int d,
                                                                   int J, long long rows_per_block) {
  // thread e handles output element (k,j) = (e / J, e % J) for e < d*J (loop if d*J > 256)
  const long long n0 = (long long)blockIdx.x * rows_per_block;
  const long long n1 = (n0 + rows_per_block < N) ? n0 + rows_per_block : N;
  for (int e = threadIdx.x; e < d * J; e += 256) {
    const int k = e / J, j = e % J;
    float acc = 0.f;
    for (long long n = n0; n < n1; ++n) acc = __builtin_fmaf(X[n * d + k], G[n * J + j], acc);
    part[(size_t)blockIdx.x * d * J + e] = acc;
  }
}

__global__ void sum_partials_kernel(const float *__restrict__ part, float *__restrict__ out, int count, int nparts,
                                    float mul) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  float acc = 0.f;
  for (int p = 0; p < nparts; ++p) acc += part[(size_t)p * count + e];
  out[e] = acc * mul;
}

// ---------------------------------------------------------------------------------------------
// Bilinear derivative.  Lane owns a row; columns broadcast from LDS.
//   S = sum_t L[i,t] R[i',t] + R[i,t] L[i',t]
//   gz[i][j] += S * e_j * d_j   (d in exp2-scaled units; caller multiplies by -scale/kExp2Scale... see host)
//   gs[i]    += S * sum_j e_j
// grid = (column splits, row blocks of 256); partial slabs reduced by bilinear_reduce_kernel.
// ---------------------------------------------------------------------------------------------
template <int JT, int TT>
__global__ __launch_bounds__(256) void bilinear_kernel(const float *__restrict__ Z, const float *__restrict__ L,
                                                       const float *__restrict__ Rm, float *__restrict__ slabG,
                                                       float *__restrict__ slabS, int N, int ldz, int T, int j0,
                                                       int cols_per_split) {
  constexpr int STR = JT + 2 * TT;
  __shared__ __attribute__((aligned(16))) float sC[64 * STR];
  const int tid = threadIdx.x;
  const int row = blockIdx.y * 256 + tid;
  const bool valid = row < N;
  const int c_begin = blockIdx.x * cols_per_split;
  if (c_begin >= N) return;
  const int c_end = (c_begin + cols_per_split < N) ? c_begin + cols_per_split : N;

  float a[JT], li[TT], ri[TT], accG[JT];
  float accS = 0.f;
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    a[j] = valid ? Z[(size_t)row * ldz + j0 + j] * kExp2Scale : 0.f;
    accG[j] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    li[t] = (valid && t < T) ? L[(size_t)row * T + t] : 0.f;
    ri[t] = (valid && t < T) ? Rm[(size_t)row * T + t] : 0.f;
  }
  for (int c0 = c_begin; c0 < c_end; c0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * STR; e += 256) {
      const int c = e / STR, q = e % STR;
      const int col = c0 + c;
      float val = 0.f;
      if (col < c_end) {
        if (q < JT) val = Z[(size_t)col * ldz + j0 + q] * kExp2Scale;
        else if (q < JT + TT) { const int t = q - JT; val = t < T ? L[(size_t)col * T + t] : 0.f; }
        else { const int t = q - JT - TT; val = t < T ? Rm[(size_t)col * T + t] : 0.f; }
      }
      sC[e] = val;
    }
    __syncthreads();
    const int nc = (c_end - c0 < 64) ? c_end - c0 : 64;
    for (int c = 0; c < nc; ++c) {
      const float *p = sC + c * STR;
      float S = 0.f;
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        S = __builtin_fmaf(li[t], p[JT + TT + t], S);
        S = __builtin_fmaf(ri[t], p[JT + t], S);
      }
      float ks = 0.f;
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const float dd = a[j] - p[j];
        const float e = fast_exp2(-(dd * dd));
        ks += e;
        accG[j] = __builtin_fmaf(S * e, dd, accG[j]);
      }
      accS = __builtin_fmaf(S, ks, accS);
    }
  }
  if (valid) {
#pragma unroll
    for (int j = 0; j < JT; ++j) slabG[((size_t)blockIdx.x * N + row) * JT + j] = accG[j];
    slabS[(size_t)blockIdx.x * N + row] = accS;
  }
}

// ---------------------------------------------------------------------------------------------
// Symmetric form of the bilinear derivative (what training runs at N >= 2048): S_ii' = S_i'i, so every unordered pair
// is evaluated ONCE and feeds both rows,
//     gz[i][j] += S e_j d_j      gz[i'][j] -= S e_j d_j      gs[i] += S sum_j e_j      gs[i'] += S sum_j e_j
// — half the exponentials of the full sweep above.  Same tile decomposition as the fused MVM (a workgroup owns BR = 512
// rows and a chunk of columns starting at its diagonal block; lane l visits column (l + s) mod 64 at step s; the JT + 1
// transposed accumulators travel with their column by a DPP wave rotate per step); two rows per lane halve the LDS
// traffic per pair (the one-row full sweep spends 11 ds_read_b128 per pair: 93 % of the LDS issue rate).
// slabR[kchunk][row][JT + 1] : row sums of a column chunk         slabT[rb][col][JT + 1] : transposed sums of row block rb
// ---------------------------------------------------------------------------------------------
// LDS-DMA of one dword per lane (wave-instruction: 64 consecutive floats at `lds_dst`); hipcc does not count it: the caller
// waits with an explicit s_waitcnt vmcnt.  M0 (the DMA's LDS base) is written in the statement that uses it.
__device__ __forceinline__ void glds_dword(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// Zs[n][q] = Z[n][j0 + q] * kExp2Scale (the derivative's DMA staging cannot scale on the way into LDS)
__global__ __launch_bounds__(256) void scale_columns_kernel(const float *__restrict__ Z, float *__restrict__ Zs, long long N,
                                                            int ldz, int j0, int JT) {
  const long long total = N * JT;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
    const long long n = g / JT;
    const int q = (int)(g - n * JT);
    Zs[g] = Z[n * ldz + j0 + q] * kExp2Scale;
  }
}

// DMA = true (JT + 2 TT a multiple of 4 with an odd quarter, so that the LDS records are unpadded and the subtile image is
// linear, and 64 records a whole number of 256-element pieces): Z is the pre-scaled copy Zs (row stride JT, j0 = 0) and the
// subtile's 64 column records are written by LDS-DMA — the register staging loop compiles to global_load; s_waitcnt vmcnt(0);
// ds_write per element (eleven serial round trips per subtile with all four waves idle), and unrolling it into registers
// spills (the kernel already holds 252 VGPRs: 106 spills, 7.0 ms).
template <int JT, int TT, bool DMA = false>
__global__ __launch_bounds__(256, 2) void bilinear_sym_kernel(const float *__restrict__ Z, const float *__restrict__ L,
                                                              const float *__restrict__ Rm, float *__restrict__ slabR,
                                                              float *__restrict__ slabT, int N, int ldz, int T, int j0,
                                                              int chunk_cols, int rotdir, int w0, int rb_first,
                                                              int slab_row0, int slab_rows) {
  constexpr int R = 2;
  constexpr int BR = 256 * R;
  constexpr int W = JT + 1;                        // slab width: JT gradient columns + the scale column
  constexpr int STRQ = JT + 2 * TT;
  constexpr int STR = (STRQ % 4 == 0) ? (((STRQ / 4) % 2 == 1) ? STRQ : STRQ + 4) : STRQ;   // (STR/4) odd: conflict-free b128
  __shared__ __attribute__((aligned(16))) float sC[64 * STR];
  __shared__ __attribute__((aligned(16))) float sT[4 * 64 * W];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  int rb, kchunk;
  wg_to_tile(blockIdx.x + w0, N, BR, chunk_cols, true, rb, kchunk);
  const int r0 = rb * BR;
  const long long cb = (long long)r0 + (long long)kchunk * chunk_cols;
  if (cb >= N) return;
  const int c_begin = (int)cb;
  const int c_end = (c_begin + chunk_cols < N) ? c_begin + chunk_cols : N;

  float a[R][JT], li[R][TT], ri[R][TT], accG[R][JT], accS[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = r0 + wave * (64 * R) + r * 64 + lane;
    const bool valid = row < N;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      if constexpr (DMA) a[r][j] = valid ? Z[(size_t)row * JT + j] : 0.f;
      else a[r][j] = valid ? Z[(size_t)row * ldz + j0 + j] * kExp2Scale : 0.f;
      accG[r][j] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {               // rows beyond N: L = R = 0 -> S = 0 -> no contribution
      li[r][t] = (valid && t < T) ? L[(size_t)row * T + t] : 0.f;
      ri[r][t] = (valid && t < T) ? Rm[(size_t)row * T + t] : 0.f;
    }
    accS[r] = 0.f;
  }

  for (int c0 = c_begin; c0 < c_end; c0 += 64) {
    __syncthreads();
    if constexpr (DMA) {
      static_assert(STR == STRQ, "linear (unpadded) subtile image");
      constexpr int NST = (64 * STRQ + 255) / 256;             // (the last piece may be partial: its tail lanes are masked off)
      const unsigned sc_bytes = (unsigned)(size_t)sC;
      const unsigned uw = __builtin_amdgcn_readfirstlane((unsigned)wave);
#pragma unroll 1                                                 // (rolled: unrolled, the eleven address computations are hoisted
      for (int it = 0; it < NST; ++it) {                         //  together and spill; the DMAs need no wait between them)
        const int e = tid + 256 * it;
        if (e >= 64 * STRQ) break;                              // (EXEC-masked lanes of a DMA do not write)
        const int c = e / STRQ, q = e % STRQ;
        const int col = c0 + c;
        const int colc = col < N ? col : N - 1;                 // clamped: every lane of the DMA reads a valid address
        const bool isz = q < JT, isl = q < JT + TT;
        int t = isz ? 0 : (isl ? q - JT : q - JT - TT);
        t = t < T ? t : T - 1;                                  // (slot t >= T: the row side's L / R are zero there)
        const float *src = isz ? Z : (isl ? L : Rm);
        const size_t off = isz ? (size_t)colc * JT + q : (size_t)colc * T + t;
        glds_dword(src + off, __builtin_amdgcn_readfirstlane(sc_bytes + (unsigned)(it * 256 + (int)uw * 64) * 4u));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (c0 + 64 > c_end) {                                    // ragged last subtile: columns beyond the end get L = R = 0
#pragma unroll 1
        for (int it = 0; it < NST; ++it) {
          const int e = tid + 256 * it;
          const int c = e / STRQ, q = e % STRQ;
          if (e < 64 * STRQ && c0 + c >= c_end && q >= JT) sC[e] = 0.f;
        }
      }
    } else {
      for (int e = tid; e < 64 * STRQ; e += 256) {
        const int c = e / STRQ, q = e % STRQ;
        const int col = c0 + c;
        float val = 0.f;                            // columns beyond the chunk: L = R = 0 -> S = 0
        if (col < c_end) {
          if (q < JT) val = Z[(size_t)col * ldz + j0 + q] * kExp2Scale;
          else if (q < JT + TT) { const int t = q - JT; val = t < T ? L[(size_t)col * T + t] : 0.f; }
          else { const int t = q - JT - TT; val = t < T ? Rm[(size_t)col * T + t] : 0.f; }
        }
        sC[c * STR + q] = val;
      }
    }
    __syncthreads();
    const bool doT = (c0 >= r0 + BR);
    float accT[W];
#pragma unroll
    for (int q = 0; q < W; ++q) accT[q] = 0.f;
#pragma unroll 1
    for (int s = 0; s < 64; ++s) {
      const float *p = sC + __mul24((lane + rotdir * s) & 63, STR);       // (24-bit multiply: v_mul_lo_u32 is quarter rate)
      float pz[JT], pl[TT], pr[TT];
#pragma unroll
      for (int j = 0; j < JT; ++j) pz[j] = p[j];
#pragma unroll
      for (int t = 0; t < TT; ++t) { pl[t] = p[JT + t]; pr[t] = p[JT + TT + t]; }
      float tg[W];
#pragma unroll
      for (int q = 0; q < W; ++q) tg[q] = accT[q];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float S = 0.f;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          S = __builtin_fmaf(li[r][t], pr[t], S);
          S = __builtin_fmaf(ri[r][t], pl[t], S);
        }
        float ks = 0.f;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          const float dd = a[r][j] - pz[j];
          const float e = fast_exp2(-(dd * dd));
          ks += e;
          const float se = S * e;                       // g = (S e) d feeds both sides as an FMA (was mul + add + sub)
          accG[r][j] = __builtin_fmaf(se, dd, accG[r][j]);
          tg[j] = __builtin_fmaf(-se, dd, tg[j]);
        }
        const float sk = S * ks;
        accS[r] += sk;
        tg[JT] += sk;
      }
      if (doT) {
#pragma unroll
        for (int q = 0; q < W; ++q) accT[q] = wave_rotate1(tg[q]);
      }
    }
    if (doT) {
#pragma unroll
      for (int q = 0; q < W; ++q) sT[(wave * 64 + lane) * W + q] = accT[q];
    }
    __syncthreads();
    if (doT) {
      for (int e = tid; e < 64 * W; e += 256) {
        const int c = e / W, q = e % W;
        const int col = c0 + c;
        if (col < c_end) {
          const float sum = sT[(0 * 64 + c) * W + q] + sT[(1 * 64 + c) * W + q] + sT[(2 * 64 + c) * W + q] +
                            sT[(3 * 64 + c) * W + q];
          slabT[((size_t)(rb - rb_first) * N + col) * W + q] = sum;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = r0 + wave * (64 * R) + r * 64 + lane;
    if (row < N) {
      float *dst = slabR + ((size_t)kchunk * slab_rows + (row - slab_row0)) * W;
#pragma unroll
      for (int j = 0; j < JT; ++j) dst[j] = accG[r][j];
      dst[JT] = accS[r];
    }
  }
}

// gZ[row][j0 + q] = mulG * (sum_k slabR[k][row][q] + sum_{b < row / BR} slabT[b][row][q]),  q < JT;   column JT -> rowS
__global__ __launch_bounds__(256) void bilinear_sym_reduce_kernel(const float *__restrict__ slabR,
                                                                  const float *__restrict__ slabT,
                                                                  float *__restrict__ gZ, float *__restrict__ rowS, int N,
                                                                  int JT, int ldg, int j0, int BR, int chunk_cols,
                                                                  int slab_rows, float mulG, int accumulate) {
  const int W = JT + 1;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)N * W) return;
  const int row = (int)(gid / W), q = (int)(gid % W);
  const int rb = row / BR;
  const int nk = (N - rb * BR + chunk_cols - 1) / chunk_cols;
  double acc = 0.0;
  // four slab entries requested together, added in slab order (a rolled `acc += load` loop waits for every load: up to
  // ~110 serial round trips per output at N = 50k)
  int k = 0;
  for (; k + 3 < nk; k += 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = slabR[(size_t)(k + u) * slab_rows * W + gid];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += (double)v[u];
  }
  for (; k < nk; ++k) acc += (double)slabR[(size_t)k * slab_rows * W + gid];
  int b = 0;
  for (; b + 3 < rb; b += 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = slabT[(size_t)(b + u) * N * W + gid];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += (double)v[u];
  }
  for (; b < rb; ++b) acc += (double)slabT[(size_t)b * N * W + gid];
  if (q < JT) gZ[(size_t)row * ldg + j0 + q] = mulG * (float)acc;
  else rowS[row] = accumulate ? rowS[row] + (float)acc : (float)acc;
}

// Dense-weight variant for the Cholesky regime (N <= max_cholesky_size): S is an explicit symmetric N x N matrix
// (e.g. Khat^-1 - alpha alpha^T).  Lane owns row i and reads S[c][i] (coalesced thanks to symmetry).
template <int JT>
__global__ __launch_bounds__(256) void bilinear_dense_kernel(const float *__restrict__ Z, const float *__restrict__ S,
                                                             float *__restrict__ slabG, float *__restrict__ slabS,
                                                             int N, int ldz, long long lds_, int j0,
                                                             int cols_per_split) {
  __shared__ __attribute__((aligned(16))) float sC[64 * JT];
  const int tid = threadIdx.x;
  const int row = blockIdx.y * 256 + tid;
  const bool valid = row < N;
  const int c_begin = blockIdx.x * cols_per_split;
  if (c_begin >= N) return;
  const int c_end = (c_begin + cols_per_split < N) ? c_begin + cols_per_split : N;
  float a[JT], accG[JT];
  float accS = 0.f;
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    a[j] = valid ? Z[(size_t)row * ldz + j0 + j] * kExp2Scale : 0.f;
    accG[j] = 0.f;
  }
  for (int c0 = c_begin; c0 < c_end; c0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * JT; e += 256) {
      const int c = e / JT, q = e % JT;
      const int col = c0 + c;
      sC[e] = (col < c_end) ? Z[(size_t)col * ldz + j0 + q] * kExp2Scale : 0.f;
    }
    __syncthreads();
    const int nc = (c_end - c0 < 64) ? c_end - c0 : 64;
    for (int c = 0; c < nc; ++c) {
      const float *p = sC + c * JT;
      const float Sv = valid ? S[(size_t)(c0 + c) * lds_ + row] : 0.f;
      float ks = 0.f;
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const float dd = a[j] - p[j];
        const float e = fast_exp2(-(dd * dd));
        ks += e;
        accG[j] = __builtin_fmaf(Sv * e, dd, accG[j]);
      }
      accS = __builtin_fmaf(Sv, ks, accS);
    }
  }
  if (valid) {
#pragma unroll
    for (int j = 0; j < JT; ++j) slabG[((size_t)blockIdx.x * N + row) * JT + j] = accG[j];
    slabS[(size_t)blockIdx.x * N + row] = accS;
  }
}

// gZ[row][j0+j] = mulG * sum_split slabG ; rowS[row] = sum_split slabS  (+= if accumulate)
__global__ void bilinear_reduce_kernel(const float *__restrict__ slabG, const float *__restrict__ slabS,
                                       float *__restrict__ gZ, float *__restrict__ rowS, int N, int JT, int ldg,
                                       int j0, int nsplit, float mulG, int accumulate) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)N * (JT + 1)) return;
  const int row = (int)(gid / (JT + 1));
  const int q = (int)(gid % (JT + 1));
  float acc = 0.f;
  if (q < JT) {
    for (int s = 0; s < nsplit; ++s) acc += slabG[((size_t)s * N + row) * JT + q];
    gZ[(size_t)row * ldg + j0 + q] = mulG * acc;
  } else {
    for (int s = 0; s < nsplit; ++s) acc += slabS[(size_t)s * N + row];
    rowS[row] = accumulate ? rowS[row] + acc : acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Bilinear derivative for the generalised family (weighted policies): same sweep as bilinear_kernel, but the
// per-row by-products are the UNWEIGHTED per-component sums  rowC[i][c] = sum_i' S_ii' phi_c(i,i')  (the gradients of
// the component weights and of the outer scale follow from them on the host).
// ---------------------------------------------------------------------------------------------
template <int JT, int TT, class KF>
__global__ __launch_bounds__(256) void family_bilinear_kernel(const float *__restrict__ Z, const float *__restrict__ L,
                                                              const float *__restrict__ Rm, float *__restrict__ slabG,
                                                              float *__restrict__ slabC, int N, int ldz, int T, int j0,
                                                              int cols_per_split, const float *__restrict__ wts) {
  constexpr int STR = JT + 2 * TT;
  constexpr int NC = JT / KF::group;
  __shared__ __attribute__((aligned(16))) float sC[64 * STR];
  const int tid = threadIdx.x;
  const int row = blockIdx.y * 256 + tid;
  const bool valid = row < N;
  const int c_begin = blockIdx.x * cols_per_split;
  if (c_begin >= N) return;
  const int c_end = (c_begin + cols_per_split < N) ? c_begin + cols_per_split : N;

  float wreg[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) wreg[c] = wts[j0 / KF::group + c];
  float a[JT], li[TT], ri[TT], accG[JT], accC[NC];
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    a[j] = valid ? Z[(size_t)row * ldz + j0 + j] * KF::pre : 0.f;
    accG[j] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) accC[c] = 0.f;
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    li[t] = (valid && t < T) ? L[(size_t)row * T + t] : 0.f;
    ri[t] = (valid && t < T) ? Rm[(size_t)row * T + t] : 0.f;
  }
  for (int c0 = c_begin; c0 < c_end; c0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * STR; e += 256) {
      const int c = e / STR, q = e % STR;
      const int col = c0 + c;
      float val = 0.f;
      if (col < c_end) {
        if (q < JT) val = Z[(size_t)col * ldz + j0 + q] * KF::pre;
        else if (q < JT + TT) { const int t = q - JT; val = t < T ? L[(size_t)col * T + t] : 0.f; }
        else { const int t = q - JT - TT; val = t < T ? Rm[(size_t)col * T + t] : 0.f; }
      }
      sC[e] = val;
    }
    __syncthreads();
    const int nc = (c_end - c0 < 64) ? c_end - c0 : 64;
    for (int c = 0; c < nc; ++c) {
      const float *p = sC + c * STR;
      float S = 0.f;
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        S = __builtin_fmaf(li[t], p[JT + TT + t], S);
        S = __builtin_fmaf(ri[t], p[JT + t], S);
      }
      KF::template pair_grad<JT>(a, p, wreg, S, accG, accC);
    }
  }
  if (valid) {
#pragma unroll
    for (int j = 0; j < JT; ++j) slabG[((size_t)blockIdx.x * N + row) * JT + j] = accG[j];
#pragma unroll
    for (int c = 0; c < NC; ++c) slabC[((size_t)blockIdx.x * N + row) * NC + c] = accC[c];
  }
}

// explicit symmetric weight matrix S (Cholesky regime), cf. bilinear_dense_kernel
template <int JT, class KF>
__global__ __launch_bounds__(256) void family_bilinear_dense_kernel(const float *__restrict__ Z,
                                                                    const float *__restrict__ S,
                                                                    float *__restrict__ slabG, float *__restrict__ slabC,
                                                                    int N, int ldz, long long lds_, int j0,
                                                                    int cols_per_split, const float *__restrict__ wts) {
  constexpr int NC = JT / KF::group;
  __shared__ __attribute__((aligned(16))) float sC[64 * JT];
  const int tid = threadIdx.x;
  const int row = blockIdx.y * 256 + tid;
  const bool valid = row < N;
  const int c_begin = blockIdx.x * cols_per_split;
  if (c_begin >= N) return;
  const int c_end = (c_begin + cols_per_split < N) ? c_begin + cols_per_split : N;
  float wreg[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) wreg[c] = wts[j0 / KF::group + c];
  float a[JT], accG[JT], accC[NC];
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    a[j] = valid ? Z[(size_t)row * ldz + j0 + j] * KF::pre : 0.f;
    accG[j] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) accC[c] = 0.f;
  for (int c0 = c_begin; c0 < c_end; c0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * JT; e += 256) {
      const int c = e / JT, q = e % JT;
      const int col = c0 + c;
      sC[e] = (col < c_end) ? Z[(size_t)col * ldz + j0 + q] * KF::pre : 0.f;
    }
    __syncthreads();
    const int nc = (c_end - c0 < 64) ? c_end - c0 : 64;
    for (int c = 0; c < nc; ++c) {
      const float Sv = valid ? S[(size_t)(c0 + c) * lds_ + row] : 0.f;
      KF::template pair_grad<JT>(a, sC + c * JT, wreg, Sv, accG, accC);
    }
  }
  if (valid) {
#pragma unroll
    for (int j = 0; j < JT; ++j) slabG[((size_t)blockIdx.x * N + row) * JT + j] = accG[j];
#pragma unroll
    for (int c = 0; c < NC; ++c) slabC[((size_t)blockIdx.x * N + row) * NC + c] = accC[c];
  }
}

// gZ[row][j0+j] = mulG * sum_split slabG ; rowC[row][c0 + c] = sum_split slabC
__global__ void family_bilinear_reduce_kernel(const float *__restrict__ slabG, const float *__restrict__ slabC,
                                              float *__restrict__ gZ, float *__restrict__ rowC, int N, int JT, int NC,
                                              int ldg, int j0, int c0, int ncomp, int nsplit, float mulG) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)N * (JT + NC)) return;
  const int row = (int)(gid / (JT + NC));
  const int q = (int)(gid % (JT + NC));
  float acc = 0.f;
  if (q < JT) {
    for (int s = 0; s < nsplit; ++s) acc += slabG[((size_t)s * N + row) * JT + q];
    gZ[(size_t)row * ldg + j0 + q] = mulG * acc;
  } else {
    const int c = q - JT;
    for (int s = 0; s < nsplit; ++s) acc += slabC[((size_t)s * N + row) * NC + c];
    rowC[(size_t)row * ncomp + c0 + c] = acc;
  }
}

// out[c] = mul * sum_row x[row][c]  (one workgroup per column, fixed order)
__global__ __launch_bounds__(1024) void sum_columns_kernel(const float *__restrict__ x, float *__restrict__ out, int n,
                                                           int ncols, float mul) {
  __shared__ float sh[1024];
  const int c = blockIdx.x;
  float a4[4] = {0.f, 0.f, 0.f, 0.f}, tail = 0.f;        // independent chains: the loads of four strides are in flight together
  int i = threadIdx.x;
  for (; i + 3 * 1024 < n; i += 4 * 1024) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x[(size_t)(i + u * 1024) * ncols + c];
#pragma unroll
    for (int u = 0; u < 4; ++u) a4[u] += v[u];
  }
  for (; i < n; i += 1024) tail += x[(size_t)i * ncols + c];
  const float acc = ((a4[0] + a4[1]) + (a4[2] + a4[3])) + tail;
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = sh[0] * mul;
}

// deterministic single-block sum: out[0] = mul * sum_i x[i]
__global__ __launch_bounds__(1024) void sum_vector_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                          int n, float mul) {
  __shared__ float sh[1024];
  // eight independent partial sums per thread (a single dependent chain of n / 1024 loads took 98 us at n = 391k); fixed order
  float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int i = threadIdx.x;
  for (; i + 7 * 1024 < n; i += 8 * 1024) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) a8[u] += v[u];
  }
  float tail = 0.f;
  for (; i < n; i += 1024) tail += x[i];
  const float acc = (((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]))) + tail;
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0] * mul;
}

// ---------------------------------------------------------------------------------------------
// Cached-K mode: out = Kd @ V + noise V for a dense N x N fp32 matrix in HBM and a thin block V (T <= 16 columns per
// pass).  HBM-bound stream of Kd; the multiply-accumulate runs on the matrix cores (v_mfma_f32_16x16x4_f32, exact
// fp32) so the VALU only issues loads.  A 256-thread workgroup owns 128 rows (32 per wave = two 16-row MFMA tiles);
// V is staged 256 rows at a time in LDS and shared by the four waves.
//   A tile: lane (m = l%16, q = l/16) loads float4 Kd[row0+m][c + 4q .. 4q+3]; MFMA i uses component i with
//   B_i[q][n] = V[c + 4q + i][n]  (any consistent ordering of the K dimension is valid);
//   D: lane holds out[row0 + 4*(l/16) + r][l%16], r = 0..3.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_gemm_kernel(const float *__restrict__ Kd, const float *__restrict__ V,
                                                         float *__restrict__ slab, int N, long long ldk, int T, int t0,
                                                         int tcnt, int rows_per_split) {
  // Kd is symmetric, so out[c][t] = sum_r Kd[r][c] V[r][t]: every wave owns 64 consecutive OUTPUT indices c and
  // streams down the rows r of Kd, 4 rows per step.  Lane (k = l/16, n = l%16) loads float4 Kd[r+k][c0 + 4n .. 4n+3]
  // (256 contiguous bytes per row per wave-instruction) and feeds its 4 components to 4 MFMAs:
  //   D_i[t][n] += sum_k V[r+k][t] * Kd[r+k][c0 + 4n + i]        (A: lane (t = l%16, k = l/16) holds V[r+k][t])
  constexpr int VC = 256;                        // V rows staged per step
  __shared__ float sV[VC * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, k = lane >> 4;
  const int c0 = blockIdx.x * 256 + wave * 64;   // first output index of this wave
  const int cl = c0 + 4 * n;                     // this lane's 4 columns
  // (padded rows: the last column group's 16-byte load may reach into the row's own padding — those values only feed
  // output columns >= N, which are not stored)
  const bool vec_ok = ((ldk & 3) == 0) && ((((uintptr_t)Kd) & 15) == 0) && cl < N && (cl + 3 < ldk);
  floatx4m acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = floatx4m{0.f, 0.f, 0.f, 0.f};
  const int rs = blockIdx.y * rows_per_split;
  const int re = (rs + rows_per_split < N) ? rs + rows_per_split : N;
  for (int rb = rs; rb < re; rb += VC) {
    __syncthreads();
    for (int e = tid; e < VC * 16; e += 256) {
      const int rr = e >> 4, t = e & 15;
      const int row = rb + rr;
      sV[e] = (row < re && t < tcnt) ? V[(size_t)row * T + t0 + t] : 0.f;
    }
    __syncthreads();
    const int rend = (rb + VC < re) ? VC : re - rb;
#pragma unroll 4
    for (int rr = 0; rr < rend; rr += 4) {
      const int row = rb + rr + k;
      float4v a = {0.f, 0.f, 0.f, 0.f};
      if (row < re) {
        const float *kp = Kd + (size_t)row * ldk + cl;
        if (vec_ok) {
          a = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(kp));
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (cl + i < N) a[i] = kp[i];
        }
      }
      const float v = sV[(rr + k) * 16 + n];     // A operand: V[row][t = n] for k = l/16 (zero beyond the split)
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, a.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, a.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, a.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, a.w, acc[3], 0, 0, 0);
    }
  }
  // D_i: lane holds out^T[t = 4k + r][c = c0 + 4n + i]
  float *sl = slab + (size_t)blockIdx.y * N * 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = cl + i;
    if (c < N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sl[(size_t)c * 16 + 4 * k + r] = acc[i][r];
    }
  }
}

// VALU form for T <= 12: the padded 16-column MFMA above spends 128 matrix-pipe cycles per KB of Kd per SIMD — as much
// as HBM allows a SIMD to receive (1 KB / 132 clk at 8 TB/s), so it cannot overlap its way to the HBM roofline.  Here a
// wave owns 256 output columns (lane = 4 consecutive columns) and streams down the rows of the symmetric Kd: every
// wave-instruction loads ONE contiguous KB of a row (perfectly coalesced), V[row][0..T) is wave-uniform (scalar loads,
// SGPR multiplier), and the update is 4 T v_fma_f32 per lane (8.8 T cycles per KB).  Split-K slabs + the same
// fixed-order reduce as the MFMA form.
// The workgroup owns 256 columns and its four waves take alternate 8-row batches of the split's rows; their accumulators
// are added through LDS in a fixed order ((w0 + w2) + (w1 + w3)) before ONE slab record is written.  (Four waves side by
// side on 1024 columns, each writing its own slab record, needed 4x the slabs for the same number of resident waves: at
// N = 15k, T = 11 the slab write burst and its re-read by the reduce were a quarter of the product's time; the split form
// is faster at every size measured, 4k <= N <= 50k — profiles/r2_gemv_wave_split_sweep.txt.)
template <int TT>
__global__ __launch_bounds__(256) void dense_gemv_valu_kernel(const float *__restrict__ Kd, const float *__restrict__ V,
                                                              float *__restrict__ slab, int N, long long ldk,
                                                              int rows_per_split) {
  // TT is the EXACT number of right-hand sides (V is N x TT, row-major): the V loads are unconditional wave-uniform
  // scalar loads that hipcc merges into s_load_dwordx4/x8, and the row loop is branch-free.  (The first version
  // guarded every load with `t < T` / per-lane alignment tests: one branch per load, 2.4 TB/s at N = 15k.)
  __shared__ float4v sRed[2 * TT * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR: keeps the V row pointer scalar
  const int cl = blockIdx.x * 256 + 4 * lane;                      // this lane's 4 output columns
  const bool base_ok = ((ldk & 3) == 0) && ((((uintptr_t)Kd) & 15) == 0);
  float acc[4][TT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[i][t] = 0.f;
  const int rs = blockIdx.y * rows_per_split;
  const int re = (rs + rows_per_split < N) ? rs + rows_per_split : N;
  constexpr int RB = 8;
  constexpr int WSTEP = 4;                        // batches between two of a wave's own
  const int wfirst = wave;
  // The 16-byte path also serves the ragged right edge when the rows are padded (cl + 3 < ldk: the pad belongs to the
  // row's own allocation; what it holds only reaches accumulators of columns >= N, which are never stored).  Sending
  // those lanes down the element-wise path made one wave per split walk its rows one dependent load at a time — at
  // N = 14 939 (N % 4 = 3) that straggler set the whole kernel's time: 208 us against 143 us for T = 11.
  if (base_ok && cl < N && cl + 3 < ldk) {
    // batches of 8 rows: 8 x 16 B per lane in flight before the first FMA (the bytes in flight per CU come from the depth
    // of each thread's own load queue).  A rolling form that re-requests row q of the next batch right after row q's FMAs
    // (sched_barrier-pinned) measured no better at any size and cost 28 VGPRs.
    int row = rs + wfirst * RB;
    const float *kp = Kd + (size_t)row * ldk + cl;
    const float *vp = V + (size_t)row * TT;
    for (; row + RB <= re; row += WSTEP * RB) {
      float4v a[RB];
#pragma unroll
      for (int q = 0; q < RB; ++q) a[q] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(kp + (size_t)q * ldk));
#pragma unroll
      for (int q = 0; q < RB; ++q) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const float v = vp[q * TT + t];                         // wave-uniform address: scalar load
          acc[0][t] = __builtin_fmaf(a[q].x, v, acc[0][t]);
          acc[1][t] = __builtin_fmaf(a[q].y, v, acc[1][t]);
          acc[2][t] = __builtin_fmaf(a[q].z, v, acc[2][t]);
          acc[3][t] = __builtin_fmaf(a[q].w, v, acc[3][t]);
        }
      }
      kp += (size_t)WSTEP * RB * ldk;
      vp += WSTEP * RB * TT;
    }
    // the ragged last rows of the last split (fewer than 8): the wave whose batch they would have started
    if (row < re) {
      for (; row < re; ++row) {
        const float4v a = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(kp));
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const float v = vp[t];
          acc[0][t] = __builtin_fmaf(a.x, v, acc[0][t]);
          acc[1][t] = __builtin_fmaf(a.y, v, acc[1][t]);
          acc[2][t] = __builtin_fmaf(a.z, v, acc[2][t]);
          acc[3][t] = __builtin_fmaf(a.w, v, acc[3][t]);
        }
        kp += ldk;
        vp += TT;
      }
    }
  } else if (cl < N) {
    // unaligned matrix (or an unpadded ragged right edge): element-wise loads
    for (int row = rs + wfirst; row < re; row += WSTEP) {
      const float *kp = Kd + (size_t)row * ldk + cl;
      float a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = (cl + i < N) ? kp[i] : 0.f;
      const float *vp = V + (size_t)row * TT;
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const float v = vp[t];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][t] = __builtin_fmaf(a[i], v, acc[i][t]);
      }
    }
  }
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
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const float4v x = sRed[t * 64 + lane];
      acc[0][t] += x.x; acc[1][t] += x.y; acc[2][t] += x.z; acc[3][t] += x.w;
    }
  }
  // compact slab, column-fastest: [split][t][Npad] — a lane's 4 columns are one 16-byte store per t and a wave's store
  // is one contiguous KB (the [column][t] layout scattered 44 dword stores per lane: 14 M partial-line writes per
  // product at N = 15k)
  const size_t npad = ((size_t)N + 3) & ~(size_t)3;
  float *sl = slab + (size_t)blockIdx.y * npad * TT;
  if (cl + 3 < N) {
#pragma unroll
    for (int t = 0; t < TT; ++t)
      *reinterpret_cast<float4v *>(sl + (size_t)t * npad + cl) = float4v{acc[0][t], acc[1][t], acc[2][t], acc[3][t]};
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = cl + i;
      if (c < N) {
#pragma unroll
        for (int t = 0; t < TT; ++t) sl[(size_t)t * npad + c] = acc[i][t];
      }
    }
  }
}

// out[row][t0+n] = sum_split slab[split][row][n] + noise * V[row][t0+n]   (fixed order: deterministic)
// `sw` = slab width (floats per (split, row)): 16 for the MFMA form, the exact T for the VALU form.
// A workgroup owns 32 consecutive slab entries; its 8 thread groups split the row-split slabs (slab s goes to group
// s % 8, 4 loads in flight per thread) and the 8 partial sums are added in a fixed order — deterministic.  (One thread
// per entry looping over ~100 slabs took 18-33 us: a third of the whole cached-K product at N = 7k.)
__global__ __launch_bounds__(256) void dense_gemm_reduce_kernel(const float *__restrict__ slab,
                                                                const float *__restrict__ V, float *__restrict__ out,
                                                                int N, int T, int t0, int tcnt, int nsplit, float noise,
                                                                int sw) {
  __shared__ float part[8][32];
  const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
  const size_t per = (size_t)N * sw;
  const size_t gid = (size_t)blockIdx.x * 32 + o;
  float acc = 0.f;
  if (gid < per) {
    int sidx = g;
    for (; sidx + 24 < nsplit; sidx += 32) {
      const float x0 = slab[(size_t)sidx * per + gid], x1 = slab[(size_t)(sidx + 8) * per + gid];
      const float x2 = slab[(size_t)(sidx + 16) * per + gid], x3 = slab[(size_t)(sidx + 24) * per + gid];
      acc += x0;
      acc += x1;
      acc += x2;
      acc += x3;
    }
    for (; sidx < nsplit; sidx += 8) acc += slab[(size_t)sidx * per + gid];
  }
  part[g][o] = acc;
  __syncthreads();
  if (g != 0 || gid >= per) return;
  const int row = (int)(gid / sw), n = (int)(gid % sw);
  if (n >= tcnt) return;
  float tot = part[0][o];
#pragma unroll
  for (int q = 1; q < 8; ++q) tot += part[q][o];
  const size_t oidx = (size_t)row * T + t0 + n;
  out[oidx] = __builtin_fmaf(noise, V[oidx], tot);
}

// Reduce of the VALU form's column-fastest slabs [split][t][npad]: out[col][t] = sum_split slab + noise * V[col][t].
// Same 8-group fixed-order scheme as above; a workgroup owns 32 consecutive columns of one t.
__global__ __launch_bounds__(256) void dense_gemv_reduce_kernel(const float *__restrict__ slab,
                                                                const float *__restrict__ V, float *__restrict__ out,
                                                                int N, int T, int nsplit, float noise) {
  __shared__ float part[8][32];
  const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
  const size_t npad = ((size_t)N + 3) & ~(size_t)3;
  const int t = blockIdx.y;
  const int col = blockIdx.x * 32 + o;
  const size_t per = npad * T;
  const size_t e = (size_t)t * npad + col;
  float acc = 0.f;
  if (col < N) {
    int sidx = g;
    for (; sidx + 24 < nsplit; sidx += 32) {
      const float x0 = slab[(size_t)sidx * per + e], x1 = slab[(size_t)(sidx + 8) * per + e];
      const float x2 = slab[(size_t)(sidx + 16) * per + e], x3 = slab[(size_t)(sidx + 24) * per + e];
      acc += x0;
      acc += x1;
      acc += x2;
      acc += x3;
    }
    for (; sidx < nsplit; sidx += 8) acc += slab[(size_t)sidx * per + e];
  }
  part[g][o] = acc;
  __syncthreads();
  if (g != 0 || col >= N) return;
  float tot = part[0][o];
#pragma unroll
  for (int q = 1; q < 8; ++q) tot += part[q][o];
  const size_t oidx = (size_t)col * T + t;
  out[oidx] = __builtin_fmaf(noise, V[oidx], tot);
}

#include "rpgp_ski_common.h"   // ski_grid_of, ski_taps, ski_wj: the grid-interpolation rows of pivchol_step_kernel

// ---------------------------------------------------------------------------------------------
// Rank-k pivoted Cholesky of K = scale * sum_j exp(-0.5 (z_ij - z_i'j)^2) (preconditioner) as ONE
// single-workgroup launch: k greedy steps of {argmax of the residual diagonal, one kernel row, rank-1 downdate}.
// L is N x k row-major (what the native mBCG executor consumes).  Work per step is O(N (J + k)): negligible next to
// one MVM, so a single CU is enough and no host round-trips are needed (the torch version costs ~10 launches/step).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void pivchol_kernel(const float *__restrict__ Z, float *__restrict__ L,
                                                       float *__restrict__ dwork, int N, int ldz, int J, int k,
                                                       float scale) {
  __shared__ float sval[16];
  __shared__ int sidx[16];
  __shared__ float szp[64];
  __shared__ float slp[64];
  __shared__ float sdp;
  __shared__ int spiv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float d0 = scale * (float)J;
  for (int i = tid; i < N; i += 1024) dwork[i] = d0;
  __syncthreads();
  for (int m = 0; m < k; ++m) {
    // argmax of the residual diagonal (ties -> smallest index: deterministic)
    float bv = -1.f;
    int bi = 0x7fffffff;
    for (int i = tid; i < N; i += 1024) {
      const float v = dwork[i];
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { sval[wave] = bv; sidx[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      float v = sval[0];
      int ix = sidx[0];
      for (int w = 1; w < 16; ++w)
        if (sval[w] > v || (sval[w] == v && sidx[w] < ix)) { v = sval[w]; ix = sidx[w]; }
      sdp = v;
      spiv = ix;
    }
    __syncthreads();
    const int piv = spiv;
    const float dp = sdp;
    const bool ok = dp > 1e-10f * d0;
    if (tid < J) szp[tid] = Z[(size_t)piv * ldz + tid] * kExp2Scale;
    if (tid < m) slp[tid] = L[(size_t)piv * k + tid];
    __syncthreads();
    const float inv_sq = ok ? 1.0f / sqrtf(dp) : 0.f;
    for (int i = tid; i < N; i += 1024) {
      float l = 0.f;
      if (ok) {
        float row = 0.f;
        for (int j = 0; j < J; ++j) {
          const float dd = Z[(size_t)i * ldz + j] * kExp2Scale - szp[j];
          row += fast_exp2(-(dd * dd));
        }
        row *= scale;
        float corr = 0.f;
        for (int q = 0; q < m; ++q) corr = __builtin_fmaf(L[(size_t)i * k + q], slp[q], corr);
        l = (row - corr) * inv_sq;
      }
      L[(size_t)i * k + m] = l;
      float nd = dwork[i] - l * l;
      nd = nd < 0.f ? 0.f : nd;
      dwork[i] = (i == piv) ? 0.f : nd;
    }
    __syncthreads();
  }
}

// Multi-workgroup form for N > 2048: one launch per greedy step.  The launch of step m first reduces the per-workgroup
// argmax partials left by step m-1 (every workgroup redundantly, same fixed order -> same pivot everywhere), then
// evaluates the pivot's kernel row on its own rows, writes column m of L, downdates the residual diagonal and leaves
// its argmax partial for step m+1 (ping-pong buffers).  k + 1 launches spread over the whole chip instead of one CU:
// 6.5 ms -> ~0.3 ms at N = 50k.  The entry K(i, piv) is evaluated by a runtime-shaped routine that also covers the other
// family members (kind / group / per-component weights; the hot path is kind RBF, group 1, unit weights).
__device__ __forceinline__ float pivchol_entry(const float *__restrict__ zi, const float *__restrict__ szp, int kind,
                                               int group, int ncomp, const float *__restrict__ wts) {
  float acc = 0.f;
  for (int c = 0; c < ncomp; ++c) {
    float phi;
    if (kind == RPGP_KIND_RBF) {
      float r2 = 0.f;
      for (int q = 0; q < group; ++q) {
        const float dd = zi[c * group + q] * kExp2Scale - szp[c * group + q];
        r2 = __builtin_fmaf(dd, dd, r2);
      }
      phi = fast_exp2(-r2);
    } else if (kind == RPGP_KIND_MATERN15) {
      phi = Phi<RPGP_KIND_MATERN15>::val(zi[c] * Phi<RPGP_KIND_MATERN15>::pre - szp[c]);
    } else if (kind == RPGP_KIND_IMQ) {
      phi = Phi<RPGP_KIND_IMQ>::val(zi[c] * Phi<RPGP_KIND_IMQ>::pre - szp[c]);
    } else {
      phi = Phi<RPGP_KIND_COSINE>::val(zi[c] * Phi<RPGP_KIND_COSINE>::pre - szp[c]);
    }
    acc = wts ? __builtin_fmaf(wts[c], phi, acc) : acc + phi;
  }
  return acc;
}

__device__ __forceinline__ float pivchol_pre(int kind) {
  return kind == RPGP_KIND_RBF ? kExp2Scale
                               : (kind == RPGP_KIND_MATERN15 ? Phi<RPGP_KIND_MATERN15>::pre
                                                             : (kind == RPGP_KIND_IMQ ? Phi<RPGP_KIND_IMQ>::pre
                                                                                      : Phi<RPGP_KIND_COSINE>::pre));
}

// block-wide argmax (ties -> smallest index); result valid in thread 0
__device__ __forceinline__ void block_argmax(float &bv, int &bi, float *sval, int *sidx) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sval[wave] = bv; sidx[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
      if (sval[w] > bv || (sval[w] == bv && sidx[w] < bi)) { bv = sval[w]; bi = sidx[w]; }
  }
}

// argmax partials of an already filled residual diagonal (SKI: the diagonal is not constant)
__global__ __launch_bounds__(256) void pivchol_init_from_diag_kernel(const float *__restrict__ dwork,
                                                                     float *__restrict__ pval, int *__restrict__ pidx,
                                                                     int N) {
  __shared__ float sval[4];
  __shared__ int sidx[4];
  float bv = -1.f;
  int bi = 0x7fffffff;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const float v = dwork[i];
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
  block_argmax(bv, bi, sval, sidx);
  if (threadIdx.x == 0) {
    pval[blockIdx.x] = bv;
    pidx[blockIdx.x] = bi;
  }
}

__global__ __launch_bounds__(256) void pivchol_step_kernel(const float *__restrict__ Z, float *__restrict__ L,
                                                           float *__restrict__ dwork, const float *__restrict__ pval_in,
                                                           const int *__restrict__ pidx_in, float *__restrict__ pval_out,
                                                           int *__restrict__ pidx_out, int nparts, int N, int ldz,
                                                           int ncols, int k, int m, float scale, float d0, int kind,
                                                           int group, int ncomp, const float *__restrict__ wts,
                                                           const float *__restrict__ gp, int G, int const_diag = 0,
                                                           float *__restrict__ Lt = nullptr) {
  // Lt (rank x N, optional): the factor is kept COLUMN-major while it is built — a row of the N x k factor is 4 k bytes, so
  // the correction sum below reads every 64-byte line of it whatever m is and the new column is one 4-byte store per line;
  // column-major, step m reads m coalesced vectors and writes one.  L is then written once at the end (pivchol_untranspose).
  __shared__ float sval[4];
  __shared__ int sidx[4];
  __shared__ float szp[64];
  __shared__ float slp[64];
  __shared__ float spw[64][4];     // SKI mode (gp != nullptr): the pivot's interpolation weights / first tap per column
  __shared__ int spidx[64];
  __shared__ float sdp;
  __shared__ int spiv;
  // pivot of this step from the previous launch's partials; const_diag (step 0 of a stationary kernel): the residual diagonal
  // is d0 everywhere and the first pivot is row 0 (ties -> smallest index) — no initialisation launch, no partials to read
  float bv = -1.f;
  int bi = 0x7fffffff;
  if (const_diag) {
    if (threadIdx.x == 0) { sdp = d0; spiv = 0; }
  } else {
    for (int q = threadIdx.x; q < nparts; q += 256) {
      const float v = pval_in[q];
      const int ix = pidx_in[q];
      if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
    block_argmax(bv, bi, sval, sidx);
    if (threadIdx.x == 0) { sdp = bv; spiv = bi; }
  }
  __syncthreads();
  const int piv = spiv;
  const float dp = sdp;
  const bool ok = dp > 1e-10f * d0;
  const float pre = pivchol_pre(kind);
  if ((int)threadIdx.x < ncols) {
    const float zp = Z[(size_t)piv * ldz + threadIdx.x];
    szp[threadIdx.x] = zp * pre;
    if (gp) {
      float w[4], dw[4];
      const float *gj = ski_grid_of(gp, ncols, threadIdx.x);
      spidx[threadIdx.x] = ski_taps<false>(zp, gj[0], gj[2], G, w, dw);
#pragma unroll
      for (int q = 0; q < 4; ++q) spw[threadIdx.x][q] = w[q];
    }
  }
  if ((int)threadIdx.x < m)
    slp[threadIdx.x] = Lt ? Lt[(size_t)threadIdx.x * N + piv] : L[(size_t)piv * k + threadIdx.x];
  __syncthreads();
  const float inv_sq = ok ? 1.0f / sqrtf(dp) : 0.f;
  bv = -1.f;
  bi = 0x7fffffff;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    float l = 0.f;
    if (ok) {
      float row;
      if (gp) {
        // K_ski(i, piv) = sum_j sum_{q,q'} w_q(z_ij) w_q'(z_pj) Toep[(idx_i + q) - (idx_p + q')]: 7 distinct lags per column
        float acc = 0.f;
        for (int j = 0; j < ncols; ++j) {
          float w[4], dw[4];
          const float *gj = ski_grid_of(gp, ncols, j);
          const int idx = ski_taps<false>(Z[(size_t)i * ldz + j], gj[0], gj[2], G, w, dw);
          const int delta = idx - spidx[j];
          float tl[7];
#pragma unroll
          for (int u = 0; u < 7; ++u) {
            const int lag = delta + u - 3;            // (the sub-kernel of the grid block: RBF unless flags say otherwise)
            tl[u] = ski_radial_f32(ski_kind(gp), lag < 0 ? -lag : lag, gj[1]);
          }
          float aj = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) aj = __builtin_fmaf(w[q] * spw[j][qq], tl[q - qq + 3], aj);
          acc = __builtin_fmaf(ski_wj(gp, j), aj, acc);
        }
        row = scale * acc;
      } else {
        row = scale * pivchol_entry(Z + (size_t)i * ldz, szp, kind, group, ncomp, wts);
      }
      float corr = 0.f;
      if (Lt) {
        for (int q = 0; q < m; ++q) corr = __builtin_fmaf(Lt[(size_t)q * N + i], slp[q], corr);
      } else {
        for (int q = 0; q < m; ++q) corr = __builtin_fmaf(L[(size_t)i * k + q], slp[q], corr);
      }
      l = (row - corr) * inv_sq;
    }
    if (Lt) Lt[(size_t)m * N + i] = l;
    else L[(size_t)i * k + m] = l;
    float nd = (const_diag ? d0 : dwork[i]) - l * l;
    nd = nd < 0.f ? 0.f : nd;
    nd = (i == piv) ? 0.f : nd;
    dwork[i] = nd;
    if (nd > bv || (nd == bv && i < bi)) { bv = nd; bi = i; }
  }
  __syncthreads();
  block_argmax(bv, bi, sval, sidx);
  if (threadIdx.x == 0) {
    pval_out[blockIdx.x] = bv;
    pidx_out[blockIdx.x] = bi;
  }
}

// The grid-interpolation operator's step with few projections (the C5 shape: J = 3), one row per thread, the factor column-major:
// as in pivchol_step_fast_kernel below, what does not depend on the pivot — the thread's coordinates and with them its taps
// and weights, its residual diagonal entry, its entries of the factor's first m columns — is requested at the head of the kernel
// and lands while the pivot is being found.  Same arithmetic in the same order as pivchol_step_kernel.
__global__ __launch_bounds__(256) void pivchol_step_ski_fast_kernel(const float *__restrict__ Z, float *__restrict__ dwork,
                                                                    const float *__restrict__ pval_in,
                                                                    const int *__restrict__ pidx_in,
                                                                    float *__restrict__ pval_out, int *__restrict__ pidx_out,
                                                                    int nparts, int N, int ldz, int ncols, int m, float scale,
                                                                    float d0, const float *__restrict__ gp, int G,
                                                                    float *__restrict__ Lt) {
  __shared__ float sval[4];
  __shared__ int sidx[4];
  __shared__ float slp[16];
  __shared__ float spw[4][4];
  __shared__ int spidx[4];
  __shared__ float sdp;
  __shared__ int spiv;
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  const bool own = i < N;
  const int ic = own ? i : N - 1;
  // the partials of the previous step (nparts <= 2 048: eight per thread, clamped, always loaded)
  float pv[8];
  int pi[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int q = tid + 256 * u;
    const int qc = q < nparts ? q : nparts - 1;
    pv[u] = pval_in[qc];
    pi[u] = pidx_in[qc];
  }
  // own-row operands
  float zr[4], lr[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) zr[j] = Z[(size_t)ic * ldz + (j < ncols ? j : ncols - 1)];
#pragma unroll
  for (int q = 0; q < 16; ++q) lr[q] = Lt[(size_t)(q < m ? q : 0) * N + ic];
  const float dprev = dwork[ic];
  const int kind = ski_kind(gp);
  float ow[4][4], ohs[4], owj[4];
  int oidx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float dw[4];
    const float *gj = ski_grid_of(gp, ncols, j < ncols ? j : ncols - 1);
    oidx[j] = ski_taps<false>(zr[j], gj[0], gj[2], G, ow[j], dw);
    ohs[j] = gj[1];
    owj[j] = ski_wj(gp, j < ncols ? j : ncols - 1);
  }
  float bv = -1.f;
  int bi = 0x7fffffff;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int q = tid + 256 * u;
    if (q < nparts && (pv[u] > bv || (pv[u] == bv && pi[u] < bi))) { bv = pv[u]; bi = pi[u]; }
  }
  block_argmax(bv, bi, sval, sidx);
  if (tid == 0) { sdp = bv; spiv = bi; }
  __syncthreads();
  const int piv = spiv;
  const float dp = sdp;
  const bool ok = dp > 1e-10f * d0;
  if (tid < ncols) {
    const float zp = Z[(size_t)piv * ldz + tid];
    float w[4], dw[4];
    const float *gj = ski_grid_of(gp, ncols, tid);
    spidx[tid] = ski_taps<false>(zp, gj[0], gj[2], G, w, dw);
#pragma unroll
    for (int q = 0; q < 4; ++q) spw[tid][q] = w[q];
  }
  if (tid < m) slp[tid] = Lt[(size_t)tid * N + piv];
  __syncthreads();
  const float inv_sq = ok ? 1.0f / sqrtf(dp) : 0.f;
  bv = -1.f;
  bi = 0x7fffffff;
  if (own) {
    float l = 0.f;
    if (ok) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < ncols) {
          const int delta = oidx[j] - spidx[j];
          float tl[7];
#pragma unroll
          for (int u = 0; u < 7; ++u) {
            const int lag = delta + u - 3;
            tl[u] = ski_radial_f32(kind, lag < 0 ? -lag : lag, ohs[j]);
          }
          float aj = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) aj = __builtin_fmaf(ow[j][q] * spw[j][qq], tl[q - qq + 3], aj);
          acc = __builtin_fmaf(owj[j], aj, acc);
        }
      }
      const float row = scale * acc;
      float corr = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (q < m) corr = __builtin_fmaf(lr[q], slp[q], corr);
      l = (row - corr) * inv_sq;
    }
    Lt[(size_t)m * N + i] = l;
    float nd = dprev - l * l;
    nd = nd < 0.f ? 0.f : nd;
    nd = (i == piv) ? 0.f : nd;
    dwork[i] = nd;
    bv = nd;
    bi = i;
  }
  __syncthreads();
  block_argmax(bv, bi, sval, sidx);
  if (tid == 0) {
    pval_out[blockIdx.x] = bv;
    pidx_out[blockIdx.x] = bi;
  }
}

// L (N x k, row-major) from the column-major scratch the steps filled: a workgroup owns 256 consecutive rows, reads 16
// coalesced vectors at a time into LDS and writes them as 64-byte runs of its rows.
__global__ __launch_bounds__(256) void pivchol_untranspose_kernel(const float *__restrict__ Lt, float *__restrict__ L, int N,
                                                                  int k) {
  __shared__ float tile[16 * 257];
  const int r0 = blockIdx.x * 256;
  const int rows = (N - r0 < 256) ? N - r0 : 256;
  const int i = r0 + threadIdx.x;
  const int ic = i < N ? i : N - 1;
  for (int q0 = 0; q0 < k; q0 += 16) {
    const int cw = (k - q0 < 16) ? k - q0 : 16;
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = Lt[(size_t)(q0 + (q < cw ? q : cw - 1)) * N + ic];       // 16 loads in flight
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[q * 257 + threadIdx.x] = v[q];
    __syncthreads();
    for (int e = threadIdx.x; e < rows * cw; e += 256) {
      const int r = e / cw, q = e - r * cw;
      L[(size_t)(r0 + r) * k + q0 + q] = tile[q * 257 + r];
    }
  }
}

// Fast path of the per-step form for the flagship operator (plain RBF projections, <= 32 columns, rank <= 16, one row per
// thread: N <= 131072).  A greedy step is a chain of dependent round trips — argmax partials -> pivot -> the pivot's row ->
// this thread's own row -> entry — of which the LAST operands do not depend on the pivot at all: the thread's coordinates,
// its row of L and its residual diagonal entry are requested at the head of the kernel, behind the partials, and land while
// the pivot is being found.  Same arithmetic in the same order as pivchol_step_kernel (padded terms add exact zeros).
__global__ __launch_bounds__(256) void pivchol_step_fast_kernel(const float *__restrict__ Z, float *__restrict__ L,
                                                                float *__restrict__ dwork,
                                                                const float *__restrict__ pval_in,
                                                                const int *__restrict__ pidx_in,
                                                                float *__restrict__ pval_out, int *__restrict__ pidx_out,
                                                                int nparts, int N, int ldz, int ncols, int k, int m,
                                                                float scale, float d0, int const_diag) {
  __shared__ float sval[4];
  __shared__ int sidx[4];
  __shared__ float szp[32];
  __shared__ float slp[16];
  __shared__ float sdp;
  __shared__ int spiv;
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  const bool own = i < N;
  const size_t ic = (size_t)(own ? i : N - 1);
  float pv[2];
  int pi[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {                     // (nparts <= 512; clamped, always loaded)
    const int q = tid + 256 * u;
    const int qc = q < nparts ? q : nparts - 1;
    pv[u] = const_diag ? -1.f : pval_in[qc];
    pi[u] = const_diag ? 0x7fffffff : pidx_in[qc];
  }
  float zr[32], lr[16];
#pragma unroll
  for (int j = 0; j < 32; ++j) zr[j] = Z[ic * ldz + (j < ncols ? j : ncols - 1)];
#pragma unroll
  for (int q = 0; q < 16; ++q) lr[q] = L[ic * k + (q < k ? q : k - 1)];
  const float dprev = const_diag ? d0 : dwork[ic];
  float bv = -1.f;
  int bi = 0x7fffffff;
  if (const_diag) {
    if (tid == 0) { sdp = d0; spiv = 0; }
  } else {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = tid + 256 * u;
      if (q < nparts && (pv[u] > bv || (pv[u] == bv && pi[u] < bi))) { bv = pv[u]; bi = pi[u]; }
    }
    block_argmax(bv, bi, sval, sidx);
    if (tid == 0) { sdp = bv; spiv = bi; }
  }
  __syncthreads();
  const int piv = spiv;
  const float dp = sdp;
  const bool ok = dp > 1e-10f * d0;
  if (tid < ncols) szp[tid] = Z[(size_t)piv * ldz + tid] * kExp2Scale;
  if (tid < m) slp[tid] = L[(size_t)piv * k + tid];
  __syncthreads();
  const float inv_sq = ok ? 1.0f / sqrtf(dp) : 0.f;
  bv = -1.f;
  bi = 0x7fffffff;
  if (own) {
    float l = 0.f;
    if (ok) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        if (j < ncols) {
          const float dd = zr[j] * kExp2Scale - szp[j];
          float r2 = 0.f;
          r2 = __builtin_fmaf(dd, dd, r2);
          acc = acc + fast_exp2(-r2);
        }
      }
      const float row = scale * acc;
      float corr = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (q < m) corr = __builtin_fmaf(lr[q], slp[q], corr);
      l = (row - corr) * inv_sq;
    }
    L[(size_t)i * k + m] = l;
    float nd = dprev - l * l;
    nd = nd < 0.f ? 0.f : nd;
    nd = (i == piv) ? 0.f : nd;
    dwork[i] = nd;
    bv = nd;
    bi = i;
  }
  __syncthreads();
  block_argmax(bv, bi, sval, sidx);
  if (tid == 0) {
    pval_out[blockIdx.x] = bv;
    pidx_out[blockIdx.x] = bi;
  }
}

// ---------------------------------------------------------------------------------------------
// DPA-GP diversification (rp.space_equally, rp.py:241-266) as ONE single-workgroup launch:
// `niter` plain gradient steps of size lr on  L(P) = sum_{a != b} cos^4(angle(P_a, P_b)),  then row normalisation.
//   c_ab = P_a.P_b / (n_a n_b);   dL/dP_a = sum_{b != a} 8 c_ab^3 ( P_b / (n_a n_b) - c_ab P_a / n_a^2 )
// J x d <= 64 x 64 lives in LDS for the whole optimisation (5000 steps ~ 10 ms instead of seconds of autograd steps on
// the host, which dominated the wall time of a DPA-GP experiment on the GPU box).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void space_equally_kernel(float *__restrict__ P, int J, int d, float lr, int niter,
                                                            float *__restrict__ final_loss) {
  __shared__ float sA[64 * 65];
  __shared__ float sB[64 * 65];      // ping-pong copies of P
  __shared__ float sC[64 * 65];      // cos_ab (diagonal zeroed)
  __shared__ float sN[64];
  __shared__ float sred[256];
  const int tid = threadIdx.x;
  float *cur = sA, *nxt = sB;
  for (int e = tid; e < J * d; e += 256) cur[(e / d) * 65 + e % d] = P[e];
  __syncthreads();
  for (int it = 0; it <= niter; ++it) {
    if (tid < J) {
      float s2 = 0.f;
      for (int q = 0; q < d; ++q) s2 = __builtin_fmaf(cur[tid * 65 + q], cur[tid * 65 + q], s2);
      sN[tid] = sqrtf(s2);
    }
    __syncthreads();
    for (int e = tid; e < J * J; e += 256) {
      const int a = e / J, b = e % J;
      float dot = 0.f;
      for (int q = 0; q < d; ++q) dot = __builtin_fmaf(cur[a * 65 + q], cur[b * 65 + q], dot);
      sC[a * 65 + b] = a == b ? 0.f : dot / (sN[a] * sN[b]);
    }
    __syncthreads();
    if (it == niter) break;          // the last pass only refreshes norms and cosines for the loss / normalisation
    for (int e = tid; e < J * d; e += 256) {
      const int a = e / d, q = e % d;
      const float na = sN[a], pa = cur[a * 65 + q];
      float g = 0.f;
      for (int b = 0; b < J; ++b) {
        const float c = sC[a * 65 + b];
        const float c3 = c * c * c;
        g = __builtin_fmaf(8.0f * c3, cur[b * 65 + q] / (na * sN[b]) - c * pa / (na * na), g);
      }
      nxt[a * 65 + q] = pa - lr * g;
    }
    __syncthreads();
    float *tmp = cur;
    cur = nxt;
    nxt = tmp;
  }
  float part = 0.f;
  for (int e = tid; e < J * J; e += 256) {
    const float c = sC[(e / J) * 65 + e % J];
    part += c * c * c * c;
  }
  sred[tid] = part;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) sred[tid] += sred[tid + w];
    __syncthreads();
  }
  if (tid == 0 && final_loss) final_loss[0] = sred[0];
  for (int e = tid; e < J * d; e += 256) P[e] = cur[(e / d) * 65 + e % d] / sN[e / d];
}

// ------------------------------- host-side helpers -------------------------------------------

int g_num_cus = 256;       // compute units of the current device (set by rpgp_init)
int g_rotdir = 0;  // +1: wave_rotate1 delivers lane l+1's value to lane l; -1: lane l-1's.  0 = not probed.

// Optional measurement hook (bench.py): HIP-event pairs around the dominant (tile) kernel launches.
constexpr int kProfMax = 4096;
bool g_prof_on = false;
int g_prof_n = 0;
hipEvent_t g_prof_ev[2 * kProfMax];

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define RPGP_CHECK(expr)                          \
  do {                                            \
    hipError_t _e = (expr);                       \
    if (_e != hipSuccess) return (int)_e;         \
  } while (0)

inline int launch_status() { return (int)hipGetLastError(); }

// decomposition of a j-range into the compiled JT pieces
// 10 / 5 / 3 make the J-shards of 2 / 4 / 8 ranks (J = 20) a single sweep per MVM
const int kJPieces[] = {20, 10, 8, 5, 4, 3, 2, 1};
inline int next_j_piece(int remaining) {
  for (int p : kJPieces)
    if (p <= remaining) return p;
  return 1;
}
const int kTPieces[] = {12, 4, 1};
inline int next_t_piece(int remaining) {
  // prefer the smallest compiled piece that covers the remainder, else the largest
  if (remaining >= 12) return 12;
  if (remaining > 4) return 12;
  if (remaining > 1) return 4;
  return 1;
}

struct TilePlan {
  int R;           // rows per lane
  int BR;          // rows per workgroup
  int nrb;         // row blocks
  int chunk_cols;  // columns per workgroup (multiple of 64)
  int total_wg;    // workgroups of the whole problem (row block by row block)
  int w0, w1;      // workgroup range of this call (pair-sharding); default = all
  int rb0, rb1;    // row blocks touched by that range
  int row0, rows;  // first row / number of rows of those row blocks (slab addressing)
  int maxchunks;   // chunk slabs per row (the most chunks any touched row block has)
  bool partial;    // the range is a strict subset: slabs are zero-initialised before the sweep
  rpgp_internal::Taper taper;   // first row blocks of the half / quarter / eighth-size chunks (nrb = none; rpgp_internal.h)
};

inline int plan_chunk(double pairs, int BR, bool big, double target_wgs = 4608.0) {
  // aim for ~4600 workgroups (6 rounds of 3 workgroups per CU) so the dispatcher can balance the triangular sweep;
  // chunks are multiples of 64 columns (one rotation subtile), at least 128 (64 for small problems)
  double cc = pairs / ((double)BR * target_wgs);
  int chunk = (int)((cc + 63.0) / 64.0) * 64;
  const int min_chunk = big ? 128 : 64;
  if (chunk < min_chunk) chunk = min_chunk;
  if (chunk > 8192) chunk = 8192;
  return chunk;
}

inline int chunks_of(const TilePlan &p, int64_t N, bool sym, int b) {
  const long long cbase = sym ? (long long)b * p.BR : 0;
  const int cb = rpgp_internal::taper_chunk(b, p.chunk_cols, p.taper.tb1, p.taper.tb2, p.taper.tb3);
  return (int)((N - cbase + cb - 1) / cb);
}

// `world`-way split: the chunk size is chosen for the per-rank share of the pairs so that every rank still launches
// a few thousand workgroups; rank r gets workgroups [total*r/world, total*(r+1)/world).
inline TilePlan make_plan(int64_t M, int64_t N, bool sym, int T, int world = 1, int rank = 0, bool r1 = false,
                          int br_override = 0, double target_wgs = 4608.0, bool taper = false, int chunk_override = 0) {
  TilePlan p;
  // two rows per lane halve the LDS traffic per pair (measured: one row per lane is 20 % slower even at T = 11)
  // measured (tools/time_small.py): with T > 4 right-hand sides two rows per lane win from N ~ 4k up (362 vs 429 us at
  // N = 14939, T = 11); with T <= 4 only once there are enough 512-row workgroups to fill the chip (N >~ 10k)
  const long long r2_min = T > 4 ? 4096 : 10240;
  p.R = (M >= r2_min && !r1) ? 2 : 1;   // r1: the family policies are instantiated with one row per lane only
  p.BR = 256 * p.R;
  if (br_override > 0) {                // matrix-core kernels (rpgp_mfma.hip): 4 waves x one 32-row MFMA tile
    p.R = 0;
    p.BR = br_override;
  }
  p.nrb = (int)((M + p.BR - 1) / p.BR);
  const double pairs = (sym ? 0.5 * (double)M * (double)N : (double)M * (double)N) / (double)(world > 0 ? world : 1);
  p.chunk_cols = chunk_override > 0 ? chunk_override : plan_chunk(pairs, p.BR, M >= 16384, target_wgs);
  p.taper = rpgp_internal::Taper{p.nrb, p.nrb, p.nrb};
  if (taper && sym && world <= 1 && M == N && p.chunk_cols >= 256) {
    // remaining share of the pairs after row block b is ((N - b BR) / N)^2: half-size chunks for the last 20 % of the work,
    // quarter-size for the last 6 %, eighth-size for the last 1.5 % (single GPU only: pair-sharding splits by workgroup COUNT)
    const double fr[3] = {0.20, 0.06, 0.015};
    int tb[3];
    for (int l = 0; l < 3; ++l) {
      const double cols = std::sqrt(fr[l]) * (double)N;
      int b = (int)(((double)N - cols) / (double)p.BR);
      if (b < 0) b = 0;
      if (b > p.nrb) b = p.nrb;
      tb[l] = b;
    }
    p.taper = rpgp_internal::Taper{tb[0], tb[1], tb[2]};
  }
  long long total = 0;
  for (int b = 0; b < p.nrb; ++b) total += chunks_of(p, N, sym, b);
  p.total_wg = (int)total;
  if (world <= 1) {
    p.w0 = 0;
    p.w1 = p.total_wg;
  } else {
    p.w0 = (int)(total * rank / world);
    p.w1 = (int)(total * (rank + 1) / world);
  }
  p.partial = !(p.w0 == 0 && p.w1 == p.total_wg);
  // row blocks touched by [w0, w1)
  p.rb0 = 0;
  p.rb1 = 0;
  long long acc = 0;
  bool started = false;
  for (int b = 0; b < p.nrb; ++b) {
    const int cb = chunks_of(p, N, sym, b);
    if (!started && p.w0 < acc + cb) { p.rb0 = b; started = true; }
    if (p.w1 > acc) p.rb1 = b + 1;
    acc += cb;
  }
  if (p.w1 <= p.w0) { p.rb0 = 0; p.rb1 = 0; }
  p.row0 = p.rb0 * p.BR;
  const long long rend = (long long)p.rb1 * p.BR < M ? (long long)p.rb1 * p.BR : M;
  p.rows = (int)(rend - p.row0 > 0 ? rend - p.row0 : 0);
  p.maxchunks = 1;
  for (int b = p.rb0; b < p.rb1; ++b) {
    const int cb = chunks_of(p, N, sym, sym ? b : 0);
    if (cb > p.maxchunks) p.maxchunks = cb;
  }
  return p;
}

template <int JT, int TT, bool SYM>
int launch_mvm_tile(const TilePlan &p, const float *Z1, const float *Z2, const float *V, float *slabR, float *slabT,
                    int M, int N, int ldz1, int ldz2, int ldv, int j0, int t0, int tcnt, int accumulate,
                    hipStream_t st) {
  dim3 grid(p.w1 - p.w0), block(256);
  if (p.R == 2)
    hipLaunchKernelGGL((mvm_tile_kernel<JT, TT, 2, SYM>), grid, block, 0, st, Z1, Z2, V, slabR, slabT, M, N, ldz1,
                       ldz2, ldv, j0, t0, tcnt, p.chunk_cols, g_rotdir, accumulate, p.w0, p.rb0, p.row0, p.rows,
                       (const float *)nullptr);
  else
    hipLaunchKernelGGL((mvm_tile_kernel<JT, TT, 1, SYM>), grid, block, 0, st, Z1, Z2, V, slabR, slabT, M, N, ldz1,
                       ldz2, ldv, j0, t0, tcnt, p.chunk_cols, g_rotdir, accumulate, p.w0, p.rb0, p.row0, p.rows,
                       (const float *)nullptr);
  return launch_status();
}

template <int JT, bool SYM>
int dispatch_t(int tt, const TilePlan &p, const float *Z1, const float *Z2, const float *V, float *slabR,
               float *slabT, int M, int N, int ldz1, int ldz2, int ldv, int j0, int t0, int tcnt, int accumulate,
               hipStream_t st) {
  switch (tt) {
    case 1: return launch_mvm_tile<JT, 1, SYM>(p, Z1, Z2, V, slabR, slabT, M, N, ldz1, ldz2, ldv, j0, t0, tcnt, accumulate, st);
    case 4: return launch_mvm_tile<JT, 4, SYM>(p, Z1, Z2, V, slabR, slabT, M, N, ldz1, ldz2, ldv, j0, t0, tcnt, accumulate, st);
    default: return launch_mvm_tile<JT, 12, SYM>(p, Z1, Z2, V, slabR, slabT, M, N, ldz1, ldz2, ldv, j0, t0, tcnt, accumulate, st);
  }
}

template <bool SYM>
int dispatch_jt(int jt, int tt, const TilePlan &p, const float *Z1, const float *Z2, const float *V, float *slabR,
                float *slabT, int M, int N, int ldz1, int ldz2, int ldv, int j0, int t0, int tcnt, int accumulate,
                hipStream_t st) {
  switch (jt) {
    case 20: return dispatch_t<20, SYM>(tt, p, Z1, Z2, V, slabR, slabT, M, N, ldz1, ldz2, ldv, j0, t0, tcnt, accumulate, st);
    case 10: return dispatch_t<10, SYM>(tt, p, Z1, Z2, V, slabR, slabT, M, N, ldz1, ldz2, ldv, j0, t0, tcnt, accumulate, st);
    case 8: return dispatch_t<8, SYM>(tt, p, Z1, Z2, V, slabR, slabT, M, N, ldz1, ldz2, ldv, j0, t0, tcnt, accumulate, st);
    case 5: return dispatch_t<5, SYM>(tt, p, Z1, Z2, V, slabR, slabT, M, N, ldz1, ldz2, ldv, j0, t0, tcnt, accumulate, st);
    case 3: return dispatch_t<3, SYM>(tt, p, Z1, Z2, V, slabR, slabT, M, N, ldz1, ldz2, ldv, j0, t0, tcnt, accumulate, st);
    case 4: return dispatch_t<4, SYM>(tt, p, Z1, Z2, V, slabR, slabT, M, N, ldz1, ldz2, ldv, j0, t0, tcnt, accumulate, st);
    case 2: return dispatch_t<2, SYM>(tt, p, Z1, Z2, V, slabR, slabT, M, N, ldz1, ldz2, ldv, j0, t0, tcnt, accumulate, st);
    default: return dispatch_t<1, SYM>(tt, p, Z1, Z2, V, slabR, slabT, M, N, ldz1, ldz2, ldv, j0, t0, tcnt, accumulate, st);
  }
}

inline size_t plan_workspace_floats(const TilePlan &p, int64_t N, int T, bool sym) {
  size_t f = (size_t)p.maxchunks * p.rows * T;
  if (sym) f += (size_t)(p.rb1 - p.rb0) * N * T;
  return f;
}

inline size_t mvm_workspace_floats(int64_t M, int64_t N, int T, bool sym, int world = 1, int rank = 0, bool r1 = false) {
  size_t f = plan_workspace_floats(make_plan(M, N, sym, T, world, rank, r1), N, T, sym);
  if (sym && !r1 && M == N && world <= 1) {       // the tapered plan of the single-GPU T = 1 prepared path (more chunk slabs)
    const size_t g = plan_workspace_floats(make_plan(M, N, sym, T, world, rank, false, 0, 4608.0, true), N, T, sym);
    if (g > f) f = g;
  }
  return f;
}

template <bool SYM>
int mvm_common(const float *Z1, const float *Z2, const float *V, float *out, int64_t M, int64_t N, int ldz1,
               int ldz2, int T, int j0, int j1, float scale, float noise, void *ws, size_t ws_bytes, void *stream,
               int world = 1, int rank = 0) {
  if (!Z1 || !Z2 || !V || !out || M <= 0 || N <= 0 || T <= 0 || j0 < 0 || j1 <= j0 || ldz1 < j1 || ldz2 < j1)
    return RPGP_EINVAL;
  if (M > 0x7fffffffLL || N > 0x7fffffffLL) return RPGP_EINVAL;
  int rc = rpgp_init();
  if (rc) return rc;
  if (world < 1 || rank < 0 || rank >= world) return RPGP_EINVAL;
  const size_t need = mvm_workspace_floats(M, N, T, SYM, world, rank) * sizeof(float);
  if (!ws || ws_bytes < need) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const TilePlan p = make_plan(M, N, SYM, T, world, rank);
  float *slabR = reinterpret_cast<float *>(ws);
  float *slabT = slabR + (size_t)p.maxchunks * p.rows * T;
  if (p.partial && need) RPGP_CHECK(hipMemsetAsync(ws, 0, need, st));   // row blocks shared with a neighbour rank
  int first = 1;
  const bool prof = g_prof_on && g_prof_n < kProfMax;
  if (prof) RPGP_CHECK(hipEventRecord(g_prof_ev[2 * g_prof_n], st));
  for (int j = j0; j < j1 && p.w1 > p.w0;) {
    const int jt = next_j_piece(j1 - j);
    for (int t0 = 0; t0 < T;) {
      const int tt = next_t_piece(T - t0);
      const int tcnt = (T - t0 < tt) ? T - t0 : tt;
      rc = dispatch_jt<SYM>(jt, tt, p, Z1, Z2, V, slabR, slabT, (int)M, (int)N, ldz1, ldz2, T, j, t0, tcnt,
                            first ? 0 : 1, st);
      if (rc) return rc;
      t0 += tcnt;
    }
    first = 0;
    j += jt;
  }
  if (prof) {
    RPGP_CHECK(hipEventRecord(g_prof_ev[2 * g_prof_n + 1], st));
    ++g_prof_n;
  }
  const size_t total = (size_t)M * T;
  RPGP_LAUNCH_MVM_REDUCE(total, st, slabR, slabT, V,
                     out, (int)M, (int)N, T, p.BR, p.chunk_cols, SYM ? 1 : 0, scale, noise, (const int *)nullptr, p.rb0,
                     p.rb1, p.row0, p.rows);
  return launch_status();
}

// Does the hand-scheduled kernel (rpgp_fact_asm.hip) serve this call?  The whole J = 20 operator, ONE right-hand side, two
// rows per lane, and the rotation direction its LDS image assumes; RPGP_FACT_ASM=0 keeps the compiler-scheduled kernel
// (read per launch: A/B pairs alternate inside one process).
inline bool fact_asm_enabled() {
  const char *env_asm = getenv("RPGP_FACT_ASM");
  return !(env_asm && atoi(env_asm) == 0);
}
inline bool fact_asm_applies(const TilePlan &p, int T, int J, int j0, int j1) {
  return fact_asm_enabled() && p.R == 2 && T == 1 && J == 20 && j0 == 0 && j1 == 20 && g_rotdir == 1;
}

// ---- prepared (factorised) path -----------------------------------------------------------------
template <int JT, int TT>
int launch_mvm_fact(const TilePlan &p, const float2v *rowdat, const float2v *coldat, const float *V, float *slabR,
                    float *slabT, int N, int J, int ldv, int j0, int t0, int tcnt, int accumulate, hipStream_t st) {
  dim3 grid(p.w1 - p.w0), block(256);
  if constexpr (TT == 1 && JT == 20) {
    if (tcnt == 1 && fact_asm_applies(p, 1, J, j0, j0 + 20))
      return rpgp_internal::launch_mvm_fact_asm(rowdat, coldat, V, slabR, slabT, N, ldv, t0, p.chunk_cols, p.taper, accumulate,
                                                p.w0, p.w1 - p.w0, p.rb0, p.row0, p.rows, st);
  }
  if constexpr (TT == 1 && JT >= 2 && JT <= 10) {
    // J-slices (the ranks of north_star's J-split; J = d models): the same generated loop with fewer quads per record
    if (tcnt == 1 && fact_asm_enabled() && p.R == 2 && g_rotdir == 1 && rpgp_internal::fact_asm_thin_supported(JT))
      return rpgp_internal::launch_mvm_fact_asm_thin(JT, rowdat, coldat, V, slabR, slabT, N, J, j0, ldv, t0, p.chunk_cols,
                                                     p.taper, accumulate, p.w0, p.w1 - p.w0, p.rb0, p.row0, p.rows, st);
  }
  if (p.R == 2)
    hipLaunchKernelGGL((mvm_fact_kernel<JT, TT, 2>), grid, block, 0, st, rowdat, coldat, V, slabR, slabT, N, J, ldv,
                       j0, t0, tcnt, p.chunk_cols, g_rotdir, accumulate, p.w0, p.rb0, p.row0, p.rows);
  else
    hipLaunchKernelGGL((mvm_fact_kernel<JT, TT, 1>), grid, block, 0, st, rowdat, coldat, V, slabR, slabT, N, J, ldv,
                       j0, t0, tcnt, p.chunk_cols, g_rotdir, accumulate, p.w0, p.rb0, p.row0, p.rows);
  return launch_status();
}

template <int JT>
int dispatch_fact_t(int tt, const TilePlan &p, const float2v *rowdat, const float2v *coldat, const float *V,
                    float *slabR, float *slabT, int N, int J, int ldv, int j0, int t0, int tcnt, int accumulate,
                    hipStream_t st) {
  switch (tt) {
    case 1: return launch_mvm_fact<JT, 1>(p, rowdat, coldat, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, accumulate, st);
    case 4: return launch_mvm_fact<JT, 4>(p, rowdat, coldat, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, accumulate, st);
    default: return launch_mvm_fact<JT, 12>(p, rowdat, coldat, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, accumulate, st);
  }
}

inline int dispatch_fact_jt(int jt, int tt, const TilePlan &p, const float2v *rowdat, const float2v *coldat,
                            const float *V, float *slabR, float *slabT, int N, int J, int ldv, int j0, int t0,
                            int tcnt, int accumulate, hipStream_t st) {
  switch (jt) {
    case 20: return dispatch_fact_t<20>(tt, p, rowdat, coldat, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, accumulate, st);
    case 10: return dispatch_fact_t<10>(tt, p, rowdat, coldat, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, accumulate, st);
    case 8: return dispatch_fact_t<8>(tt, p, rowdat, coldat, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, accumulate, st);
    case 5: return dispatch_fact_t<5>(tt, p, rowdat, coldat, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, accumulate, st);
    case 3: return dispatch_fact_t<3>(tt, p, rowdat, coldat, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, accumulate, st);
    case 4: return dispatch_fact_t<4>(tt, p, rowdat, coldat, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, accumulate, st);
    case 2: return dispatch_fact_t<2>(tt, p, rowdat, coldat, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, accumulate, st);
    default: return dispatch_fact_t<1>(tt, p, rowdat, coldat, V, slabR, slabT, N, J, ldv, j0, t0, tcnt, accumulate, st);
  }
}

template <int JT>
int launch_dense(const float *Z1, const float *Z2, float *out, int M, int N, int ldz1, int ldz2, long long ldo,
                 int j0, float scale, int accumulate, hipStream_t st) {
  dim3 grid((N + 255) / 256, (M + 15) / 16);
  hipLaunchKernelGGL((dense_kernel<JT>), grid, dim3(256), 0, st, Z1, Z2, out, M, N, ldz1, ldz2, ldo, j0, scale,
                     accumulate, (const float *)nullptr);
  return launch_status();
}

template <int JT>
int launch_bilinear(int tt, const float *Z, const float *L, const float *R, float *slabG, float *slabS, int N,
                    int ldz, int T, int j0, int cols_per_split, int nsplit, hipStream_t st) {
  dim3 grid(nsplit, (N + 255) / 256);
  if (tt <= 1)
    hipLaunchKernelGGL((bilinear_kernel<JT, 1>), grid, dim3(256), 0, st, Z, L, R, slabG, slabS, N, ldz, T, j0, cols_per_split);
  else if (tt <= 4)
    hipLaunchKernelGGL((bilinear_kernel<JT, 4>), grid, dim3(256), 0, st, Z, L, R, slabG, slabS, N, ldz, T, j0, cols_per_split);
  else
    hipLaunchKernelGGL((bilinear_kernel<JT, 12>), grid, dim3(256), 0, st, Z, L, R, slabG, slabS, N, ldz, T, j0, cols_per_split);
  return launch_status();
}

template <int JT>
int launch_bilinear_dense(const float *Z, const float *S, float *slabG, float *slabS, int N, int ldz, long long lds_,
                          int j0, int cols_per_split, int nsplit, hipStream_t st) {
  dim3 grid(nsplit, (N + 255) / 256);
  hipLaunchKernelGGL((bilinear_dense_kernel<JT>), grid, dim3(256), 0, st, Z, S, slabG, slabS, N, ldz, lds_, j0,
                     cols_per_split);
  return launch_status();
}

// symmetric-sweep plan of the bilinear derivative (BR = 512): slabR [maxchunks][N][21] + slabT [row blocks][N][21] + rowS
inline bool bilinear_use_sym(int64_t N) {
  if (N < 2048) return false;
  const double slabT_bytes = (double)((N + 511) / 512) * (double)N * 21.0 * 4.0;
  return slabT_bytes <= 6.0e9;                     // beyond ~190k rows the full sweep's 46 MB workspace is kept
}
// (Chunk sizes from 448 to 896 columns at C4, 64 to 192 at C3 / C2 are within 2 % of each other for this sweep: its nine
//  rounds of unequal workgroups leave the dispatcher enough to smooth — profiles/r5b_bilinear_chunk_sweep.jsonl.)
inline TilePlan bilinear_plan(int64_t N) { return make_plan(N, N, true, 12, 1, 0, false, 512); }
inline size_t bilinear_sym_floats(int64_t N) {
  const TilePlan p = bilinear_plan(N);
  return ((size_t)p.maxchunks * N + (size_t)p.nrb * N) * 21 + (size_t)N + (size_t)N * 20;     // slabs, rowS, scaled copy of Z
}

template <int JT>
int launch_bilinear_sym(int tt, const TilePlan &p, const float *Z, const float *L, const float *R, float *slabR,
                               float *slabT, int N, int ldz, int T, int j0, hipStream_t st, float *Zs) {
  dim3 grid(p.total_wg), block(256);
  // (JT + 8 or JT + 24 a multiple of 8 would pad the LDS records: JT = 8 keeps the register staging)
  constexpr bool dma4 = !((JT + 8) % 4 == 0 && ((JT + 8) / 4) % 2 == 0), dma12 = !((JT + 24) % 4 == 0 && ((JT + 24) / 4) % 2 == 0);
  if constexpr (dma4 && dma12) {
    // the column records are staged by LDS-DMA from a pre-scaled copy of Z (20-column piece at C4: 6.26 -> 5.87 ms alternating
    // in one process, bit-identical results)
    if (Zs) {
      const long long total = (long long)N * JT;
      hipLaunchKernelGGL(scale_columns_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256),
                         0, st, Z, Zs, (long long)N, ldz, j0, JT);
      if constexpr (JT == 20) {
        // the hand-scheduled loop (rpgp_bil_asm.hip) for the training block (5 .. 12 right-hand-side slots) of the whole
        // J = 20 operator; RPGP_BIL_ASM=0 keeps the compiler-scheduled kernel (A/B runs, tests)
        const char *ea = getenv("RPGP_BIL_ASM");
        if (tt > 4 && T <= 12 && g_rotdir == 1 && !(ea && ea[0] == '0'))
          return rpgp_internal::launch_bilinear_sym_asm(Zs, L, R, slabR, slabT, N, T, p.chunk_cols, p.total_wg, st);
      }
      if (tt <= 4)
        hipLaunchKernelGGL((bilinear_sym_kernel<JT, 4, true>), grid, block, 0, st, Zs, L, R, slabR, slabT, N, JT, T, 0,
                           p.chunk_cols, g_rotdir, 0, 0, 0, N);
      else
        hipLaunchKernelGGL((bilinear_sym_kernel<JT, 12, true>), grid, block, 0, st, Zs, L, R, slabR, slabT, N, JT, T, 0,
                           p.chunk_cols, g_rotdir, 0, 0, 0, N);
      return launch_status();
    }
  }
  if (tt <= 4)
    hipLaunchKernelGGL((bilinear_sym_kernel<JT, 4>), grid, block, 0, st, Z, L, R, slabR, slabT, N, ldz, T, j0,
                       p.chunk_cols, g_rotdir, 0, 0, 0, N);
  else
    hipLaunchKernelGGL((bilinear_sym_kernel<JT, 12>), grid, block, 0, st, Z, L, R, slabR, slabT, N, ldz, T, j0,
                       p.chunk_cols, g_rotdir, 0, 0, 0, N);
  return launch_status();
}

// ---- generalised family dispatch ---------------------------------------------------------------
#include <type_traits>
template <int V> using IntC = std::integral_constant<int, V>;

// calls f(policy object) for the (kind, group) pair; RPGP_EINVAL for combinations that are not instantiated
template <class F>
int with_family_policy(int kind, int group, F &&f) {
  switch (kind) {
    case RPGP_KIND_RBF:
      switch (group) {
        case 1: return f(KfGroupRbf<1>{});
        case 2: return f(KfGroupRbf<2>{});
        case 3: return f(KfGroupRbf<3>{});
        case 4: return f(KfGroupRbf<4>{});
        case 5: return f(KfGroupRbf<5>{});
        case 8: return f(KfGroupRbf<8>{});
        case 10: return f(KfGroupRbf<10>{});
        case 20: return f(KfGroupRbf<20>{});
        default: return RPGP_EINVAL;
      }
    case RPGP_KIND_MATERN15: return group == 1 ? f(KfPhi<RPGP_KIND_MATERN15>{}) : RPGP_EINVAL;
    case RPGP_KIND_IMQ: return group == 1 ? f(KfPhi<RPGP_KIND_IMQ>{}) : RPGP_EINVAL;
    case RPGP_KIND_COSINE: return group == 1 ? f(KfPhi<RPGP_KIND_COSINE>{}) : RPGP_EINVAL;
    default: return RPGP_EINVAL;
  }
}

// column pieces: {10, 4, 2, 1} columns for 1-D sub-kernels; for G-dimensional groups as many whole groups as fit in
// ~10 columns (fewer passes over the pair space), then single groups
template <int G> struct GroupBig { static constexpr int v = (G <= 5) ? G * (10 / G) : G; };
template <class KF>
inline int family_piece(int remaining) {
  if (KF::group > 1) return remaining >= GroupBig<KF::group>::v ? GroupBig<KF::group>::v : KF::group;
  return remaining >= 10 ? 10 : (remaining >= 4 ? 4 : (remaining >= 2 ? 2 : 1));
}
template <class KF, class F>
int with_family_piece(int jt, F &&f) {
  if constexpr (KF::group > 1) {
    if (jt == GroupBig<KF::group>::v) return f(IntC<GroupBig<KF::group>::v>{});
    return f(IntC<KF::group>{});
  } else {
    switch (jt) {
      case 10: return f(IntC<10>{});
      case 4: return f(IntC<4>{});
      case 2: return f(IntC<2>{});
      default: return f(IntC<1>{});
    }
  }
}

inline int family_check(const rpgp_family *fam, int ld_a, int ld_b) {
  if (!fam || !fam->weights || fam->ncomp <= 0 || fam->group <= 0) return RPGP_EINVAL;
  const long long cols = (long long)fam->ncomp * fam->group;
  if (cols > ld_a || cols > ld_b) return RPGP_EINVAL;
  return with_family_policy(fam->kind, fam->group, [](auto) { return 0; });
}

template <class KF, int JT, bool SYM>
int launch_family_tile(int tt, const TilePlan &p, const float *Z1, const float *Z2, const float *V, float *slabR,
                       float *slabT, int M, int N, int ldz1, int ldz2, int ldv, int j0, int t0, int tcnt,
                       int accumulate, const float *wts, hipStream_t st) {
  dim3 grid(p.w1 - p.w0), block(256);
  switch (tt) {
    case 1:
      hipLaunchKernelGGL((mvm_tile_kernel<JT, 1, 1, SYM, KF>), grid, block, 0, st, Z1, Z2, V, slabR, slabT, M, N, ldz1,
                         ldz2, ldv, j0, t0, tcnt, p.chunk_cols, g_rotdir, accumulate, p.w0, p.rb0, p.row0, p.rows, wts);
      break;
    case 4:
      hipLaunchKernelGGL((mvm_tile_kernel<JT, 4, 1, SYM, KF>), grid, block, 0, st, Z1, Z2, V, slabR, slabT, M, N, ldz1,
                         ldz2, ldv, j0, t0, tcnt, p.chunk_cols, g_rotdir, accumulate, p.w0, p.rb0, p.row0, p.rows, wts);
      break;
    default:
      hipLaunchKernelGGL((mvm_tile_kernel<JT, 12, 1, SYM, KF>), grid, block, 0, st, Z1, Z2, V, slabR, slabT, M, N, ldz1,
                         ldz2, ldv, j0, t0, tcnt, p.chunk_cols, g_rotdir, accumulate, p.w0, p.rb0, p.row0, p.rows, wts);
      break;
  }
  return launch_status();
}

template <bool SYM>
int family_mvm_common(const rpgp_family *fam, const float *Z1, const float *Z2, const float *V, float *out, int64_t M,
                      int64_t N, int ldz1, int ldz2, int T, float scale, float noise, void *ws, size_t ws_bytes,
                      void *stream) {
  if (!Z1 || !Z2 || !V || !out || M <= 0 || N <= 0 || T <= 0) return RPGP_EINVAL;
  if (M > 0x7fffffffLL || N > 0x7fffffffLL) return RPGP_EINVAL;
  int rc = family_check(fam, ldz1, ldz2);
  if (rc) return rc;
  rc = rpgp_init();
  if (rc) return rc;
  const size_t need = mvm_workspace_floats(M, N, T, SYM, 1, 0, true) * sizeof(float);
  if (!ws || ws_bytes < need) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const TilePlan p = make_plan(M, N, SYM, T, 1, 0, true);
  float *slabR = reinterpret_cast<float *>(ws);
  float *slabT = slabR + (size_t)p.maxchunks * p.rows * T;
  const int J = fam->ncomp * fam->group;
  const float *wts = fam->weights;
  rc = with_family_policy(fam->kind, fam->group, [&](auto kf) -> int {
    using KF = decltype(kf);
    int first = 1;
    for (int j = 0; j < J;) {
      const int jt = family_piece<KF>(J - j);
      for (int t0 = 0; t0 < T;) {
        const int tt = next_t_piece(T - t0);
        const int tcnt = (T - t0 < tt) ? T - t0 : tt;
        const int r2 = with_family_piece<KF>(jt, [&](auto jc) -> int {
          return launch_family_tile<KF, decltype(jc)::value, SYM>(tt, p, Z1, Z2, V, slabR, slabT, (int)M, (int)N, ldz1,
                                                                  ldz2, T, j, t0, tcnt, first ? 0 : 1, wts, st);
        });
        if (r2) return r2;
        t0 += tcnt;
      }
      first = 0;
      j += jt;
    }
    return 0;
  });
  if (rc) return rc;
  const size_t total = (size_t)M * T;
  RPGP_LAUNCH_MVM_REDUCE(total, st, slabR, slabT, V, out, (int)M, (int)N, T, p.BR, p.chunk_cols, SYM ? 1 : 0, scale, noise,
                     (const int *)nullptr, p.rb0, p.rb1, p.row0, p.rows);
  return launch_status();
}

inline int bilinear_nsplit(int64_t N) {
  const int64_t nrb = (N + 255) / 256;
  int64_t ns = (2048 + nrb - 1) / nrb;
  const int64_t maxs = (N + 255) / 256;
  if (ns > maxs) ns = maxs;
  if (ns < 1) ns = 1;
  return (int)ns;
}

}  // namespace

// ------------------------------------ C ABI ---------------------------------------------------

namespace {
// ---- packed symmetric cache: host side ------------------------------------------------------------------------------
struct SymkPlan {
  TilePlan p;
  long long sub0, sub1;        // subtile range [sub0, sub1) of this rank's workgroups
};
inline long long symk_host_first_subtile(int rb, int64_t N, int BR) {
  const long long nsN = (N + 63) / 64, q = BR / 64;
  return (long long)rb * nsN - q * ((long long)rb * (rb - 1) / 2);
}
// first subtile of workgroup `lin` of the plan's numbering (lin == total_wg: one past the last subtile)
inline long long symk_wg_subtile(const TilePlan &p, int64_t N, int lin) {
  long long acc = 0;
  for (int b = 0; b < p.nrb; ++b) {
    const int cb = chunks_of(p, N, true, b);
    if (lin < acc + cb) return symk_host_first_subtile(b, N, p.BR) + (lin - acc) * (long long)(p.chunk_cols / 64);
    acc += cb;
  }
  return symk_host_first_subtile(p.nrb, N, p.BR);
}
// One row-tile set per wave (R = 1: 256-row blocks) for a whole cache up to this size — squarer workgroup tiles and twice the
// workgroups to spread over the chip (tools/experiments/r5_symk_r1_sweep.py, profiles/r5b_symk_r1_sweep.jsonl: wide T = 11
// N = 4k 22.6 -> 20.5 us, 5.5k 25.6 -> 22.0, 7.4k 35.0 -> 32.2, 11k 56.8 -> 53.8, 15k 107.6 -> 100.6; thin T = 1 15k 86.1 -> 77.1).
// The LAYOUT of the cache follows R, so every entry point asks this one function.
constexpr int64_t kSymkR1Max = 16384;
inline bool symk_r1(int64_t N, int world) { return world <= 1 && N <= kSymkR1Max; }

// Smallest chunk (multiple of 64 columns) whose workgroups all fit ONE resident round of `slots`: with every workgroup
// running at once the product's time is the longest workgroup's, and a handful of workgroups beyond the slots wait for a
// whole workgroup time (N = 14939: 523 workgroups of 896 columns 123 us, 465 of 1024 columns 101 us).
inline int symk_one_round_chunk(int64_t N, int BR, int slots) {
  const int nrb = (int)((N + BR - 1) / BR);
  for (int c = 64; c <= 8192; c += 64) {
    long long total = 0;
    for (int b = 0; b < nrb; ++b) total += (N - (long long)b * BR + c - 1) / c;
    if (total <= slots) return c;
  }
  return 8192;
}

inline SymkPlan symk_plan(int64_t N, int world, int rank, bool wide = false) {
  SymkPlan sp;
  // A cached workgroup has no exponentials to hide its prologue (row block of V, ring start-up) and epilogue (slabs)
  // behind, so it wants longer column chunks than the fused sweep: measured optimum ~N / 14 workgroups per rank
  // (N = 15k: 1000, T = 11 product 0.152 ms against 0.199 ms with the fused sweep's 4600; N = 50k: 3600, 1.31 ms).
  double wgs = (double)N / 14.0;
  wgs = wgs < 512.0 ? 512.0 : (wgs > 4608.0 ? 4608.0 : wgs);
  const bool r1 = symk_r1(N, world);
  int chunk = 0;
  if (wide && world <= 1) {
    // The matrix-core (wide) product keeps two workgroups per CU resident and a workgroup's prologue / slabs cost more than
    // in the thin form: few, long workgroups.  The subtile layout of the cache depends on (N, R) only, so the build and the
    // product of a whole (unsharded) cache may be chunked differently; sharded caches keep one rule for both (their byte
    // ranges follow it).
    // EXACTLY filled rounds of the 2-per-CU resident workgroups: one round up to N = 36 000, three above
    // (tools/experiments/r5_symk_rounds.py: 28 001: 342 -> 307 us, 33 000: 426 -> 397, 40 000: 613 -> 592, 50 000: 957 -> 934
    // against the N / 28 rule, whose workgroup count lands between two rounds)
    // (two per CU also for the 256-row kernel, whose 154 registers would admit three: sized for three it is 0 - 9 % slower)
    chunk = symk_one_round_chunk(N, r1 ? 256 : 512, (N <= 36000 ? 1 : 3) * 2 * (g_num_cus > 0 ? g_num_cus : 256));
  }
  sp.p = make_plan(N, N, true, 12, world, rank, r1, 0, wgs, false, chunk);      // R = 2 above kSymkR1Max, whatever T is later
  sp.sub0 = symk_wg_subtile(sp.p, N, sp.p.w0);
  sp.sub1 = symk_wg_subtile(sp.p, N, sp.p.w1);
  return sp;
}
// (+ one subtile of padding: the wide product's requests run a few tiles past a workgroup's last subtile)
inline size_t symk_bytes(const SymkPlan &sp) { return (size_t)(sp.sub1 - sp.sub0 + 1) * sp.p.BR * 64 * sizeof(float); }
// nontemporal loads unless the (rank's share of the) cache fits the Infinity Cache
inline bool symk_nt_loads(const SymkPlan &sp) {
  return symk_bytes(sp) > (size_t)300000000;      // (277 MB: default policy 60.9 us against 64.6; 464 MB: 113 against 106)
}
inline int symk_t_piece(int remaining) {
  if (remaining > 4) return 12;                        // (an exact T = 11 instantiation measured slower than the padded 12)
  if (remaining > 1) return 4;
  return 1;
}
template <int JT>
int symk_launch_build(const SymkPlan &sp, const float *Z, float4v *cache, int N, int ldz, int j0, int accumulate,
                      hipStream_t st) {
  dim3 grid(sp.p.w1 - sp.p.w0), block(256);
  if (sp.p.R == 2)
    hipLaunchKernelGGL((symk_build_kernel<JT, 2>), grid, block, 0, st, Z, cache, N, ldz, j0, sp.p.chunk_cols, g_rotdir,
                       accumulate, sp.p.w0, sp.sub0);
  else
    hipLaunchKernelGGL((symk_build_kernel<JT, 1>), grid, block, 0, st, Z, cache, N, ldz, j0, sp.p.chunk_cols, g_rotdir,
                       accumulate, sp.p.w0, sp.sub0);
  return launch_status();
}
template <int JT>
int symk_launch_build_tile(const SymkPlan &sp, const float *Z, float4v *cache, int N, int ldz, int j0, int accumulate,
                           hipStream_t st) {
  dim3 grid(sp.p.w1 - sp.p.w0), block(256);
  if (sp.p.R == 2)
    hipLaunchKernelGGL((symk_build_tile_kernel<JT, 2>), grid, block, 0, st, Z, cache, N, ldz, j0, sp.p.chunk_cols,
                       accumulate, sp.p.w0, sp.sub0);
  else
    hipLaunchKernelGGL((symk_build_tile_kernel<JT, 1>), grid, block, 0, st, Z, cache, N, ldz, j0, sp.p.chunk_cols,
                       accumulate, sp.p.w0, sp.sub0);
  return launch_status();
}
inline int symk_launch_mvm_tile(const SymkPlan &sp, const float4v *cache, const float *V, float *slabR, float *slabT,
                                int N, int T, int t0, int tcnt, hipStream_t st) {
  dim3 grid(sp.p.w1 - sp.p.w0), block(256);
  const bool nt = symk_nt_loads(sp);
#define RPGP_SYMK_T2(RR, NTF)                                                                                        \
  hipLaunchKernelGGL((symk_mvm_tile2_kernel<RR, NTF>), grid, block, 0, st, cache, V, slabR, slabT, N, T, t0, tcnt,   \
                     sp.p.chunk_cols, sp.p.w0, sp.p.rb0, sp.p.row0, sp.p.rows, sp.sub0)
  if (sp.p.R == 2) {
    if (nt) RPGP_SYMK_T2(2, true); else RPGP_SYMK_T2(2, false);
  } else {
    if (nt) RPGP_SYMK_T2(1, true); else RPGP_SYMK_T2(1, false);
  }
#undef RPGP_SYMK_T2
  return launch_status();
}
template <int TT>
int symk_launch_mvm(const SymkPlan &sp, const float4v *cache, const float *V, float *slabR, float *slabT, int N, int T,
                    int t0, int tcnt, hipStream_t st) {
  dim3 grid(sp.p.w1 - sp.p.w0), block(256);
  const bool nt = symk_nt_loads(sp);
#define RPGP_SYMK_THIN(RR, NTF)                                                                                      \
  hipLaunchKernelGGL((symk_mvm_kernel<TT, RR, NTF>), grid, block, 0, st, cache, V, slabR, slabT, N, T, t0, tcnt,     \
                     sp.p.chunk_cols, g_rotdir, 0, sp.p.w0, sp.p.rb0, sp.p.row0, sp.p.rows, sp.sub0)
  if (sp.p.R == 2) {
    if (nt) RPGP_SYMK_THIN(2, true); else RPGP_SYMK_THIN(2, false);
  } else {
    if (nt) RPGP_SYMK_THIN(1, true); else RPGP_SYMK_THIN(1, false);
  }
#undef RPGP_SYMK_THIN
  return launch_status();
}
}  // namespace


namespace rpgp_internal {
// fixed-order sums of a row vector / of the columns of an N x C block (the by-products of the derivative kernels), for the
// translation units that produce such blocks (rpgp_ski_base.hip)
int sum_vector_launch(const float *x, float *out, int n, float mul, hipStream_t st) {
  hipLaunchKernelGGL(sum_vector_kernel, dim3(1), dim3(1024), 0, st, x, out, n, mul);
  return (int)hipGetLastError();
}
int sum_columns_launch(const float *x, float *out, int n, int C, float mul, hipStream_t st) {
  hipLaunchKernelGGL(sum_columns_kernel, dim3(C), dim3(1024), 0, st, x, out, n, C, mul);
  return (int)hipGetLastError();
}
}  // namespace rpgp_internal

extern "C" {

int rpgp_version(void) { return RPGP_ABI_VERSION; }

const char *rpgp_error_string(int code) {
  switch (code) {
    case 0: return "success";
    case RPGP_EINVAL: return "rpgp: invalid argument";
    case RPGP_EWORKSPACE: return "rpgp: workspace too small";
    case RPGP_ENODEVICE: return "rpgp: no usable gfx950 device";
    case RPGP_ENUMERIC: return "rpgp: NaNs encountered in an iterative solve";
    default: return hipGetErrorString((hipError_t)code);
  }
}

int rpgp_init(void) {
  if (g_rotdir != 0) return 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RPGP_ENODEVICE;
  {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0)
      g_num_cus = ncu;
  }
  int *d = nullptr;
  RPGP_CHECK(hipMalloc(&d, 64 * sizeof(int)));
  hipLaunchKernelGGL(probe_rotate_kernel, dim3(1), dim3(64), 0, 0, d);
  int h[64];
  hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return (int)e;
  if (h[0] == 1 && h[63] == 0) g_rotdir = 1;
  else if (h[0] == 63 && h[1] == 0) g_rotdir = -1;
  else return RPGP_ENODEVICE;
  return 0;
}

int rpgp_prepared_kernel_id(int64_t N, int J, int T) {
  if (N <= 0 || J <= 0 || T <= 0) return -RPGP_EINVAL;
  if (rpgp_init()) return -RPGP_EINVAL;
  const TilePlan p = make_plan(N, N, true, T);
  if (fact_asm_applies(p, T, J, 0, J)) return 1;
  // a sweep over J projections that is one compiled piece (a rank's J-slice, or a model with that many projections)
  if (T == 1 && fact_asm_enabled() && p.R == 2 && g_rotdir == 1 && rpgp_internal::fact_asm_thin_supported(J)) return 2;
  return 0;
}

int rpgp_profile_begin(void) {
  if (!g_prof_on) {
    for (int i = 0; i < 2 * kProfMax; ++i)
      if (!g_prof_ev[i]) RPGP_CHECK(hipEventCreate(&g_prof_ev[i]));
  }
  g_prof_on = true;
  g_prof_n = 0;
  return 0;
}

int rpgp_profile_end(float *avg_ms_host, int *count_host) {
  g_prof_on = false;
  double sum = 0.0;
  for (int i = 0; i < g_prof_n; ++i) {
    RPGP_CHECK(hipEventSynchronize(g_prof_ev[2 * i + 1]));
    float ms = 0.f;
    RPGP_CHECK(hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]));
    sum += ms;
  }
  if (avg_ms_host) *avg_ms_host = g_prof_n ? (float)(sum / g_prof_n) : 0.f;
  if (count_host) *count_host = g_prof_n;
  g_prof_n = 0;
  return 0;
}

int rpgp_space_equally(float *P, int J, int d, float lr, int niter, float *final_loss, void *stream) {
  if (!P || J <= 0 || d <= 0 || J > 64 || d > 64 || niter < 0) return RPGP_EINVAL;
  hipLaunchKernelGGL(space_equally_kernel, dim3(1), dim3(256), 0, as_stream(stream), P, J, d, lr, niter, final_loss);
  return launch_status();
}

int rpgp_project(const float *X, const float *Peff, float *Z, int64_t N, int d, int J, void *stream) {
  if (!X || !Peff || !Z || N <= 0 || d <= 0 || J <= 0 || (size_t)(d + 3) * (J + 15) * 4 > 64 * 1024) return RPGP_EINVAL;
  const long long stripes = (N + 15) / 16;
  int blocks = (int)((stripes + 3) / 4 < 4096 ? (stripes + 3) / 4 : 4096);
  const size_t lds = (size_t)((d + 3) & ~3) * ((J + 15) & ~15) * sizeof(float);
  hipLaunchKernelGGL(project_kernel, dim3(blocks), dim3(256), lds, as_stream(stream), X, Peff, Z, (long long)N, d, J);
  return launch_status();
}

int rpgp_project_grad(const float *X, const float *G, float *dPeff, int64_t N, int d, int J, void *stream) {
  if (!X || !G || !dPeff || N <= 0 || d <= 0 || J <= 0) return RPGP_EINVAL;
  // two-pass deterministic reduction; partials live in a small static-size device buffer per call
  // (allocated by the caller-visible workspace would be cleaner, but d*J*nblk is tiny) -> use hipMallocAsync.
  hipStream_t st = as_stream(stream);
  if (d <= 64 && J <= 64) {
    // thin X^T G on the matrix cores with float64 accumulation (rpgp_precond.hip; the slab kernel below walked 512 rows
    // serially per thread on N / 512 workgroups: 88 us at N = 7372, d = 8, J = 20)
    double *gpart = nullptr;
    RPGP_CHECK(hipMallocAsync((void **)&gpart, rpgp_internal::gram_part_bytes(d, J), st));
    const int grc = rpgp_internal::gram_launch(X, d, G, J, (long long)N, d, J, nullptr, dPeff, gpart, st);
    (void)hipFreeAsync(gpart, st);
    return grc;
  }
  const int nblk = (int)((N + 511) / 512 < 1024 ? (N + 511) / 512 : 1024);
  const long long rows_per_block = (N + nblk - 1) / nblk;
  float *part = nullptr;
  RPGP_CHECK(hipMallocAsync((void **)&part, (size_t)nblk * d * J * sizeof(float), st));
  hipLaunchKernelGGL(project_grad_partial_kernel, dim3(nblk), dim3(256), 0, st, X, G, part, (long long)N, d, J,
                     rows_per_block);
  hipLaunchKernelGGL(sum_partials_kernel, dim3((d * J + 255) / 256), dim3(256), 0, st, part, dPeff, d * J, nblk, 1.0f);
  int rc = launch_status();
  (void)hipFreeAsync(part, st);
  return rc;
}

size_t rpgp_mvm_sym_workspace_bytes(int64_t N, int T) {
  if (N <= 0 || T <= 0) return 0;
  return mvm_workspace_floats(N, N, T, true) * sizeof(float);
}

int rpgp_mvm_sym(const float *Z, const float *V, float *out, int64_t N, int ldz, int T, int j0, int j1, float scale,
                 float noise, void *workspace, size_t workspace_bytes, void *stream) {
  return mvm_common<true>(Z, Z, V, out, N, N, ldz, ldz, T, j0, j1, scale, noise, workspace, workspace_bytes, stream);
}

// workgroups of the range pass: 256 rows each (a thread walks ~256 J / 256 coordinates; the round-5 form gave a thread of
// its 25 workgroups 2 000 serial rows and took 60 us at C4 — VERDICT r5 #6a), at most 1 024 partial records
inline int prep_minmax_blocks(int64_t N) { return (int)((N + 255) / 256 < 1024 ? (N + 255) / 256 : 1024); }

size_t rpgp_prepare_bytes(int64_t N, int J) {
  if (N <= 0 || J <= 0 || J > kPrepMidFloats) return 0;
  // header + mid + rowdat + coldat + min/max partials (tail)
  const size_t nblk = (size_t)prep_minmax_blocks(N);
  return (kPrepHeaderFloats + kPrepMidFloats) * sizeof(float) + 2 * (size_t)N * J * sizeof(float2v) +
         nblk * J * 2 * sizeof(float);
}

int rpgp_prepare(const float *Z, int64_t N, int ldz, int J, void *prep, size_t prep_bytes, void *stream) {
  if (!Z || !prep || N <= 0 || J <= 0 || J > kPrepMidFloats || ldz < J) return RPGP_EINVAL;
  if (prep_bytes < rpgp_prepare_bytes(N, J)) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  PrepLayout L = prep_layout(prep, N, J);
  float *part = reinterpret_cast<float *>(L.coldat + (size_t)N * J);
  const int nblk = prep_minmax_blocks(N);
  const long long rows_per_block = (N + nblk - 1) / nblk;
  hipLaunchKernelGGL(prep_minmax_kernel, dim3(nblk), dim3(256), 0, st, Z, part, (long long)N, ldz, J, rows_per_block);
  hipLaunchKernelGGL(prep_finish_kernel, dim3(1), dim3(256), 0, st, part, nblk, J, L.header, L.mid);
  const long long total = N * J;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(prep_build_kernel, dim3(blocks), dim3(256), 0, st, Z, L.mid, L.rowdat, L.coldat, (long long)N, ldz, J);
  return launch_status();
}

int rpgp_prepare_status(const void *prep, int *fast_ok_host, float *max_abs_host, void *stream) {
  if (!prep) return RPGP_EINVAL;
  float h[2];
  hipStream_t st = as_stream(stream);
  RPGP_CHECK(hipMemcpyAsync(h, prep, sizeof(h), hipMemcpyDeviceToHost, st));
  RPGP_CHECK(hipStreamSynchronize(st));
  if (fast_ok_host) *fast_ok_host = *reinterpret_cast<int *>(&h[0]);
  if (max_abs_host) *max_abs_host = h[1];
  return 0;
}

int rpgp_mvm_sym_prepared(const void *prep, const float *V, float *out, int64_t N, int J, int T, int j0, int j1,
                          float scale, float noise, void *workspace, size_t workspace_bytes, void *stream) {
  return rpgp_mvm_sym_prepared_range(prep, V, out, N, J, T, j0, j1, 1, 0, scale, noise, workspace, workspace_bytes,
                                     stream);
}

size_t rpgp_mvm_sym_range_workspace_bytes(int64_t N, int T, int world, int rank) {
  if (N <= 0 || T <= 0 || world < 1 || rank < 0 || rank >= world) return 0;
  return mvm_workspace_floats(N, N, T, true, world, rank) * sizeof(float);
}

int rpgp_mvm_sym_range(const float *Z, const float *V, float *out, int64_t N, int ldz, int T, int j0, int j1, int world,
                       int rank, float scale, float noise, void *workspace, size_t workspace_bytes, void *stream) {
  return mvm_common<true>(Z, Z, V, out, N, N, ldz, ldz, T, j0, j1, scale, noise, workspace, workspace_bytes, stream,
                          world, rank);
}

int rpgp_mvm_sym_prepared_range(const void *prep, const float *V, float *out, int64_t N, int J, int T, int j0, int j1,
                                int world, int rank, float scale, float noise, void *workspace, size_t workspace_bytes,
                                void *stream) {
  if (!prep || !V || !out || N <= 0 || T <= 0 || J <= 0 || J > kPrepMidFloats || j0 < 0 || j1 <= j0 || j1 > J)
    return RPGP_EINVAL;
  if (N > 0x7fffffffLL) return RPGP_EINVAL;
  int rc = rpgp_init();
  if (rc) return rc;
  if (world < 1 || rank < 0 || rank >= world) return RPGP_EINVAL;
  const size_t need = mvm_workspace_floats(N, N, T, true, world, rank) * sizeof(float);
  if (!workspace || workspace_bytes < need) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  TilePlan p = make_plan(N, N, true, T, world, rank);
  // the hand-scheduled kernel (rpgp_fact_asm.hip) understands tapered chunks: smaller workgroups at the end of the sweep
  if (fact_asm_applies(p, T, J, j0, j1)) p = make_plan(N, N, true, T, world, rank, false, 0, 4608.0, true);
  PrepLayout L = prep_layout(const_cast<void *>(prep), N, J);
  float *slabR = reinterpret_cast<float *>(workspace);
  float *slabT = slabR + (size_t)p.maxchunks * p.rows * T;
  if (p.partial && need) RPGP_CHECK(hipMemsetAsync(workspace, 0, need, st));
  int first = 1;
  const bool prof = g_prof_on && g_prof_n < kProfMax;
  if (prof) RPGP_CHECK(hipEventRecord(g_prof_ev[2 * g_prof_n], st));
  for (int j = j0; j < j1 && p.w1 > p.w0;) {
    const int jt = next_j_piece(j1 - j);
    for (int t0 = 0; t0 < T;) {
      const int tt = next_t_piece(T - t0);
      const int tcnt = (T - t0 < tt) ? T - t0 : tt;
      rc = dispatch_fact_jt(jt, tt, p, L.rowdat, L.coldat, V, slabR, slabT, (int)N, J, T, j, t0, tcnt, first ? 0 : 1, st);
      if (rc) return rc;
      t0 += tcnt;
    }
    first = 0;
    j += jt;
  }
  if (prof) {
    RPGP_CHECK(hipEventRecord(g_prof_ev[2 * g_prof_n + 1], st));
    ++g_prof_n;
  }
  const size_t total = (size_t)N * T;
  RPGP_LAUNCH_MVM_REDUCE(total, st, slabR, slabT, V, out,
                     (int)N, (int)N, T, p.BR, p.chunk_cols, 1, scale, noise,
                     reinterpret_cast<const int *>(L.header), p.rb0, p.rb1, p.row0, p.rows, p.taper);
  return launch_status();
}

size_t rpgp_mvm_rect_workspace_bytes(int64_t M, int64_t N, int T) {
  if (M <= 0 || N <= 0 || T <= 0) return 0;
  return mvm_workspace_floats(M, N, T, false) * sizeof(float);
}

int rpgp_mvm_rect(const float *Z1, const float *Z2, const float *V, float *out, int64_t M, int64_t N, int ldz1,
                  int ldz2, int T, int j0, int j1, float scale, void *workspace, size_t workspace_bytes,
                  void *stream) {
  return mvm_common<false>(Z1, Z2, V, out, M, N, ldz1, ldz2, T, j0, j1, scale, 0.f, workspace, workspace_bytes,
                           stream);
}

int rpgp_dense(const float *Z1, const float *Z2, float *out, int64_t M, int64_t N, int ldz1, int ldz2, int64_t ldo,
               int j0, int j1, float scale, void *stream) {
  if (!Z1 || !Z2 || !out || M <= 0 || N <= 0 || j0 < 0 || j1 <= j0 || ldz1 < j1 || ldz2 < j1 || ldo < N)
    return RPGP_EINVAL;
  if (M > 0x7fffffffLL || N > 0x7fffffffLL) return RPGP_EINVAL;
  hipStream_t st = as_stream(stream);
  int first = 1;
  for (int j = j0; j < j1;) {
    const int jt = next_j_piece(j1 - j);
    int rc;
    switch (jt) {
      case 20: rc = launch_dense<20>(Z1, Z2, out, (int)M, (int)N, ldz1, ldz2, ldo, j, scale, !first, st); break;
      case 10: rc = launch_dense<10>(Z1, Z2, out, (int)M, (int)N, ldz1, ldz2, ldo, j, scale, !first, st); break;
      case 8: rc = launch_dense<8>(Z1, Z2, out, (int)M, (int)N, ldz1, ldz2, ldo, j, scale, !first, st); break;
      case 5: rc = launch_dense<5>(Z1, Z2, out, (int)M, (int)N, ldz1, ldz2, ldo, j, scale, !first, st); break;
      case 3: rc = launch_dense<3>(Z1, Z2, out, (int)M, (int)N, ldz1, ldz2, ldo, j, scale, !first, st); break;
      case 4: rc = launch_dense<4>(Z1, Z2, out, (int)M, (int)N, ldz1, ldz2, ldo, j, scale, !first, st); break;
      case 2: rc = launch_dense<2>(Z1, Z2, out, (int)M, (int)N, ldz1, ldz2, ldo, j, scale, !first, st); break;
      default: rc = launch_dense<1>(Z1, Z2, out, (int)M, (int)N, ldz1, ldz2, ldo, j, scale, !first, st); break;
    }
    if (rc) return rc;
    first = 0;
    j += jt;
  }
  return 0;
}

size_t rpgp_bilinear_grad_workspace_bytes(int64_t N, int J) {
  if (N <= 0 || J <= 0) return 0;
  const int ns = bilinear_nsplit(N);
  // slabG [ns][N][<=20] + slabS [ns][N] + rowS [N]
  size_t f = (size_t)ns * N * 21 + (size_t)N;
  if (bilinear_use_sym(N)) {
    const size_t g = bilinear_sym_floats(N);
    if (g > f) f = g;
  }
  return f * sizeof(float);
}

int rpgp_bilinear_grad(const float *Z, const float *L, const float *R, float *gZ, float *gscale, int64_t N, int ldz,
                       int ldg, int T, int j0, int j1, float scale, void *workspace, size_t workspace_bytes,
                       void *stream) {
  if (!Z || !L || !R || !gZ || !gscale || N <= 0 || T <= 0 || T > 12 || j0 < 0 || j1 <= j0 || ldz < j1 || ldg < j1)
    return RPGP_EINVAL;
  if (N > 0x7fffffffLL) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_bilinear_grad_workspace_bytes(N, j1 - j0)) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  if (bilinear_use_sym(N)) {
    // symmetric sweep: every unordered pair once (half the exponentials), tile plan of the fused MVM with BR = 512
    int rc = rpgp_init();
    if (rc) return rc;
    const TilePlan p = bilinear_plan(N);
    float *slabR = reinterpret_cast<float *>(workspace);
    float *slabT = slabR + (size_t)p.maxchunks * N * 21;
    float *rowS = slabT + (size_t)p.nrb * N * 21;
    float *Zs = rowS + (size_t)N;
    int first = 1;
    for (int j = j0; j < j1;) {
      const int jt = next_j_piece(j1 - j);
      switch (jt) {
        case 20: rc = launch_bilinear_sym<20>(T, p, Z, L, R, slabR, slabT, (int)N, ldz, T, j, st, Zs); break;
        case 10: rc = launch_bilinear_sym<10>(T, p, Z, L, R, slabR, slabT, (int)N, ldz, T, j, st, Zs); break;
        case 8: rc = launch_bilinear_sym<8>(T, p, Z, L, R, slabR, slabT, (int)N, ldz, T, j, st, Zs); break;
        case 5: rc = launch_bilinear_sym<5>(T, p, Z, L, R, slabR, slabT, (int)N, ldz, T, j, st, Zs); break;
        case 3: rc = launch_bilinear_sym<3>(T, p, Z, L, R, slabR, slabT, (int)N, ldz, T, j, st, Zs); break;
        case 4: rc = launch_bilinear_sym<4>(T, p, Z, L, R, slabR, slabT, (int)N, ldz, T, j, st, Zs); break;
        case 2: rc = launch_bilinear_sym<2>(T, p, Z, L, R, slabR, slabT, (int)N, ldz, T, j, st, Zs); break;
        default: rc = launch_bilinear_sym<1>(T, p, Z, L, R, slabR, slabT, (int)N, ldz, T, j, st, Zs); break;
      }
      if (rc) return rc;
      const size_t total = (size_t)N * (jt + 1);
      hipLaunchKernelGGL(bilinear_sym_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, slabR,
                         slabT, gZ, rowS, (int)N, jt, ldg, j, p.BR, p.chunk_cols, (int)N, -scale / kExp2Scale,
                         first ? 0 : 1);
      rc = launch_status();
      if (rc) return rc;
      first = 0;
      j += jt;
    }
    hipLaunchKernelGGL(sum_vector_kernel, dim3(1), dim3(1024), 0, st, rowS, gscale, (int)N, 0.5f);
    return launch_status();
  }
  const int ns_max = bilinear_nsplit(N);          // the slab layout (and the workspace size) uses the upper bound
  int cps = (int)((N + ns_max - 1) / ns_max);
  cps = (cps + 63) / 64 * 64;
  // splits that actually own columns: rounding cps up to 64 can leave trailing splits empty, and an empty split's
  // workgroups return without writing their slab — the reduce must not read them
  const int ns = (int)((N + cps - 1) / cps);
  float *slabG = reinterpret_cast<float *>(workspace);
  float *slabS = slabG + (size_t)ns_max * N * 20;
  float *rowS = slabS + (size_t)ns_max * N;
  int first = 1;
  for (int j = j0; j < j1;) {
    const int jt = next_j_piece(j1 - j);
    int rc;
    switch (jt) {
      case 20: rc = launch_bilinear<20>(T, Z, L, R, slabG, slabS, (int)N, ldz, T, j, cps, ns, st); break;
      case 10: rc = launch_bilinear<10>(T, Z, L, R, slabG, slabS, (int)N, ldz, T, j, cps, ns, st); break;
      case 8: rc = launch_bilinear<8>(T, Z, L, R, slabG, slabS, (int)N, ldz, T, j, cps, ns, st); break;
      case 5: rc = launch_bilinear<5>(T, Z, L, R, slabG, slabS, (int)N, ldz, T, j, cps, ns, st); break;
      case 3: rc = launch_bilinear<3>(T, Z, L, R, slabG, slabS, (int)N, ldz, T, j, cps, ns, st); break;
      case 4: rc = launch_bilinear<4>(T, Z, L, R, slabG, slabS, (int)N, ldz, T, j, cps, ns, st); break;
      case 2: rc = launch_bilinear<2>(T, Z, L, R, slabG, slabS, (int)N, ldz, T, j, cps, ns, st); break;
      default: rc = launch_bilinear<1>(T, Z, L, R, slabG, slabS, (int)N, ldz, T, j, cps, ns, st); break;
    }
    if (rc) return rc;
    // d/dZ = -scale * S e (z_i - z_i') ; kernel accumulated S e (c z_i - c z_i')  ->  multiply by -scale / c
    const size_t total = (size_t)N * (jt + 1);
    hipLaunchKernelGGL(bilinear_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, slabG, slabS,
                       gZ, rowS, (int)N, jt, ldg, j, ns, -scale / kExp2Scale, first ? 0 : 1);
    rc = launch_status();
    if (rc) return rc;
    first = 0;
    j += jt;
  }
  hipLaunchKernelGGL(sum_vector_kernel, dim3(1), dim3(1024), 0, st, rowS, gscale, (int)N, 0.5f);
  return launch_status();
}

int rpgp_bilinear_grad_dense(const float *Z, const float *S, float *gZ, float *gscale, int64_t N, int ldz, int ldg,
                             int64_t lds_, int j0, int j1, float scale, void *workspace, size_t workspace_bytes,
                             void *stream) {
  if (!Z || !S || !gZ || !gscale || N <= 0 || j0 < 0 || j1 <= j0 || ldz < j1 || ldg < j1 || lds_ < N)
    return RPGP_EINVAL;
  if (N > 0x7fffffffLL) return RPGP_EINVAL;
  if (!workspace || workspace_bytes < rpgp_bilinear_grad_workspace_bytes(N, j1 - j0)) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const int ns_max = bilinear_nsplit(N);          // the slab layout (and the workspace size) uses the upper bound
  int cps = (int)((N + ns_max - 1) / ns_max);
  cps = (cps + 63) / 64 * 64;
  // splits that actually own columns: rounding cps up to 64 can leave trailing splits empty, and an empty split's
  // workgroups return without writing their slab — the reduce must not read them
  const int ns = (int)((N + cps - 1) / cps);
  float *slabG = reinterpret_cast<float *>(workspace);
  float *slabS = slabG + (size_t)ns_max * N * 20;
  float *rowS = slabS + (size_t)ns_max * N;
  int first = 1;
  for (int j = j0; j < j1;) {
    const int jt = next_j_piece(j1 - j);
    int rc;
    switch (jt) {
      case 20: rc = launch_bilinear_dense<20>(Z, S, slabG, slabS, (int)N, ldz, lds_, j, cps, ns, st); break;
      case 10: rc = launch_bilinear_dense<10>(Z, S, slabG, slabS, (int)N, ldz, lds_, j, cps, ns, st); break;
      case 8: rc = launch_bilinear_dense<8>(Z, S, slabG, slabS, (int)N, ldz, lds_, j, cps, ns, st); break;
      case 5: rc = launch_bilinear_dense<5>(Z, S, slabG, slabS, (int)N, ldz, lds_, j, cps, ns, st); break;
      case 3: rc = launch_bilinear_dense<3>(Z, S, slabG, slabS, (int)N, ldz, lds_, j, cps, ns, st); break;
      case 4: rc = launch_bilinear_dense<4>(Z, S, slabG, slabS, (int)N, ldz, lds_, j, cps, ns, st); break;
      case 2: rc = launch_bilinear_dense<2>(Z, S, slabG, slabS, (int)N, ldz, lds_, j, cps, ns, st); break;
      default: rc = launch_bilinear_dense<1>(Z, S, slabG, slabS, (int)N, ldz, lds_, j, cps, ns, st); break;
    }
    if (rc) return rc;
    const size_t total = (size_t)N * (jt + 1);
    hipLaunchKernelGGL(bilinear_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, slabG, slabS,
                       gZ, rowS, (int)N, jt, ldg, j, ns, -scale / kExp2Scale, first ? 0 : 1);
    rc = launch_status();
    if (rc) return rc;
    first = 0;
    j += jt;
  }
  hipLaunchKernelGGL(sum_vector_kernel, dim3(1), dim3(1024), 0, st, rowS, gscale, (int)N, 0.5f);
  return launch_status();
}

// ---- generalised family: C-ABI -------------------------------------------------------------------
size_t rpgp_family_mvm_workspace_bytes(int64_t M, int64_t N, int T, int sym) {
  if (M <= 0 || N <= 0 || T <= 0) return 0;
  return mvm_workspace_floats(M, N, T, sym != 0, 1, 0, true) * sizeof(float);
}

int rpgp_family_mvm_sym(const rpgp_family *fam, const float *Z, const float *V, float *out, int64_t N, int ldz, int T,
                        float scale, float noise, void *workspace, size_t workspace_bytes, void *stream) {
  return family_mvm_common<true>(fam, Z, Z, V, out, N, N, ldz, ldz, T, scale, noise, workspace, workspace_bytes,
                                 stream);
}

int rpgp_family_mvm_rect(const rpgp_family *fam, const float *Z1, const float *Z2, const float *V, float *out,
                         int64_t M, int64_t N, int ldz1, int ldz2, int T, float scale, void *workspace,
                         size_t workspace_bytes, void *stream) {
  return family_mvm_common<false>(fam, Z1, Z2, V, out, M, N, ldz1, ldz2, T, scale, 0.f, workspace, workspace_bytes,
                                  stream);
}

int rpgp_family_dense(const rpgp_family *fam, const float *Z1, const float *Z2, float *out, int64_t M, int64_t N,
                      int ldz1, int ldz2, int64_t ldo, float scale, void *stream) {
  if (!Z1 || !Z2 || !out || M <= 0 || N <= 0 || ldo < N || M > 0x7fffffffLL || N > 0x7fffffffLL) return RPGP_EINVAL;
  int rc = family_check(fam, ldz1, ldz2);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const int J = fam->ncomp * fam->group;
  const float *wts = fam->weights;
  return with_family_policy(fam->kind, fam->group, [&](auto kf) -> int {
    using KF = decltype(kf);
    int first = 1;
    for (int j = 0; j < J;) {
      const int jt = family_piece<KF>(J - j);
      const int r2 = with_family_piece<KF>(jt, [&](auto jc) -> int {
        constexpr int JT = decltype(jc)::value;
        dim3 grid((unsigned)((N + 255) / 256), (unsigned)((M + 15) / 16));
        hipLaunchKernelGGL((dense_kernel<JT, KF>), grid, dim3(256), 0, st, Z1, Z2, out, (int)M, (int)N, ldz1, ldz2,
                           (long long)ldo, j, scale, first ? 0 : 1, wts);
        return launch_status();
      });
      if (r2) return r2;
      first = 0;
      j += jt;
    }
    return 0;
  });
}

size_t rpgp_family_bilinear_grad_workspace_bytes(int64_t N, int ncols, int ncomp) {
  if (N <= 0 || ncols <= 0 || ncomp <= 0) return 0;
  const int ns = bilinear_nsplit(N);
  // slabG [ns][N][<=20] + slabC [ns][N][<=10] + rowC [N][ncomp]
  return ((size_t)ns * N * 30 + (size_t)N * ncomp) * sizeof(float);
}

// mode 0: S = L R^T + R L^T from N x T factors; mode 1: explicit symmetric S (N x N, row stride lds)
static int family_bilinear_common(const rpgp_family *fam, const float *Z, const float *L, const float *R,
                                  const float *S, int64_t lds_, float *gZ, float *gcomp, int64_t N, int ldz, int ldg,
                                  int T, float scale, void *workspace, size_t workspace_bytes, void *stream) {
  if (!Z || !gZ || !gcomp || N <= 0 || N > 0x7fffffffLL) return RPGP_EINVAL;
  int rc = family_check(fam, ldz, ldg);
  if (rc) return rc;
  const int J = fam->ncomp * fam->group, C = fam->ncomp;
  if (!workspace || workspace_bytes < rpgp_family_bilinear_grad_workspace_bytes(N, J, C)) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const int ns_max = bilinear_nsplit(N);          // the slab layout (and the workspace size) uses the upper bound
  int cps = (int)((N + ns_max - 1) / ns_max);
  cps = (cps + 63) / 64 * 64;
  // splits that actually own columns: rounding cps up to 64 can leave trailing splits empty, and an empty split's
  // workgroups return without writing their slab — the reduce must not read them
  const int ns = (int)((N + cps - 1) / cps);
  float *slabG = reinterpret_cast<float *>(workspace);
  float *slabC = slabG + (size_t)ns_max * N * 20;
  float *rowC = slabC + (size_t)ns_max * N * 10;
  const float *wts = fam->weights;
  rc = with_family_policy(fam->kind, fam->group, [&](auto kf) -> int {
    using KF = decltype(kf);
    for (int j = 0; j < J;) {
      const int jt = family_piece<KF>(J - j);
      const int r2 = with_family_piece<KF>(jt, [&](auto jc) -> int {
        constexpr int JT = decltype(jc)::value;
        constexpr int NC = JT / KF::group;
        dim3 grid(ns, (unsigned)((N + 255) / 256));
        if (S) {
          hipLaunchKernelGGL((family_bilinear_dense_kernel<JT, KF>), grid, dim3(256), 0, st, Z, S, slabG, slabC, (int)N,
                             ldz, (long long)lds_, j, cps, wts);
        } else if (T <= 1) {
          hipLaunchKernelGGL((family_bilinear_kernel<JT, 1, KF>), grid, dim3(256), 0, st, Z, L, R, slabG, slabC, (int)N,
                             ldz, T, j, cps, wts);
        } else if (T <= 4) {
          hipLaunchKernelGGL((family_bilinear_kernel<JT, 4, KF>), grid, dim3(256), 0, st, Z, L, R, slabG, slabC, (int)N,
                             ldz, T, j, cps, wts);
        } else {
          hipLaunchKernelGGL((family_bilinear_kernel<JT, 12, KF>), grid, dim3(256), 0, st, Z, L, R, slabG, slabC, (int)N,
                             ldz, T, j, cps, wts);
        }
        int r3 = launch_status();
        if (r3) return r3;
        const size_t total = (size_t)N * (JT + NC);
        hipLaunchKernelGGL(family_bilinear_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, slabG,
                           slabC, gZ, rowC, (int)N, JT, NC, ldg, j, j / KF::group, C, ns, scale * KF::mulG);
        return launch_status();
      });
      if (r2) return r2;
      j += jt;
    }
    return 0;
  });
  if (rc) return rc;
  hipLaunchKernelGGL(sum_columns_kernel, dim3(C), dim3(1024), 0, st, rowC, gcomp, (int)N, C, 0.5f);
  return launch_status();
}

int rpgp_family_bilinear_grad(const rpgp_family *fam, const float *Z, const float *L, const float *R, float *gZ,
                              float *gcomp, int64_t N, int ldz, int ldg, int T, float scale, void *workspace,
                              size_t workspace_bytes, void *stream) {
  if (!L || !R || T <= 0 || T > 12) return RPGP_EINVAL;
  return family_bilinear_common(fam, Z, L, R, nullptr, 0, gZ, gcomp, N, ldz, ldg, T, scale, workspace, workspace_bytes,
                                stream);
}

int rpgp_family_bilinear_grad_dense(const rpgp_family *fam, const float *Z, const float *S, float *gZ, float *gcomp,
                                    int64_t N, int ldz, int ldg, int64_t lds_, float scale, void *workspace,
                                    size_t workspace_bytes, void *stream) {
  if (!S || lds_ < N) return RPGP_EINVAL;
  return family_bilinear_common(fam, Z, nullptr, nullptr, S, lds_, gZ, gcomp, N, ldz, ldg, 0, scale, workspace,
                                workspace_bytes, stream);
}

// diag_work: N + kPivcholScratch floats (residual diagonal + argmax partials of the multi-workgroup form)
static int pivchol_common(const float *Z, float *L, float *diag_work, int64_t N, int ldz, int ncols, int rank,
                          float scale, float d0, int kind, int group, int ncomp, const float *wts, void *stream,
                          const float *gp = nullptr, int G = 0, float *Lt = nullptr) {
  hipStream_t st = as_stream(stream);
  if (N <= 2048 && kind == RPGP_KIND_RBF && group == 1 && !wts && !gp) {      // launch-latency regime: one workgroup
    hipLaunchKernelGGL(pivchol_kernel, dim3(1), dim3(1024), 0, st, Z, L, diag_work, (int)N, ldz, ncols, rank, scale);
    return launch_status();
  }
  // (all greedy steps as ONE cooperative launch with grid barriers was built in round 5 and measured no faster than the
  //  per-step launches — a grid barrier ~8 us against a 1.5 - 2 us launch boundary: tools/experiments/r6_removed_forms.patch)
  int nb = (int)((N + 255) / 256);
  // up to 2 048 workgroups: one row per thread up to N = 524 288.  (A cap of 512 made a thread of the C5 operator walk three
  // rows one after the other at two waves per SIMD — a chain of dependent global round trips per row, 14 us per step with
  // nothing to read from the factor yet.)
  constexpr int kMaxParts = RPGP_PIVCHOL_SCRATCH / 4;
  if (nb > kMaxParts) nb = kMaxParts;
  float *pval[2] = {diag_work + N, diag_work + N + kMaxParts};
  int *pidx[2] = {reinterpret_cast<int *>(diag_work + N + 2 * kMaxParts), reinterpret_cast<int *>(diag_work + N + 3 * kMaxParts)};
  if (gp) {    // SKI: the residual diagonal starts at diag(K_ski), which depends on the point
    const int drc = rpgp_ski_diag(Z, gp, diag_work, N, ldz, ncols, G, scale, stream);       // (rpgp_ski_base.hip)
    if (drc) return drc;
    hipLaunchKernelGGL(pivchol_init_from_diag_kernel, dim3(nb), dim3(256), 0, st, diag_work, pval[0], pidx[0], (int)N);
  }
  // (a stationary kernel's residual diagonal starts at d0 everywhere: step 0 knows that itself — no initialisation launch)
  // the flagship operator's fast per-step kernel (own-row operands requested before the pivot is known); every other operator,
  // wide coordinate rows and ranks beyond 16 take the general per-step kernel
  const bool fast = kind == RPGP_KIND_RBF && group == 1 && !wts && !gp && ncols <= 32 &&
                    rank <= 16 && N <= 131072;
  for (int m = 0; fast && m < rank; ++m)
    hipLaunchKernelGGL(pivchol_step_fast_kernel, dim3(nb), dim3(256), 0, st, Z, L, diag_work, pval[m & 1], pidx[m & 1],
                       pval[(m + 1) & 1], pidx[(m + 1) & 1], nb, (int)N, ldz, ncols, rank, m, scale, d0, m == 0 ? 1 : 0);
  if (fast) return launch_status();
  // grid interpolation with few projections, one row per thread, column-major factor: own-row operands requested up front
  const bool ski_fast = gp && Lt && ncols <= 4 && rank <= 16 && (long long)nb * 256 >= N && !wts;
  for (int m = 0; ski_fast && m < rank; ++m)
    hipLaunchKernelGGL(pivchol_step_ski_fast_kernel, dim3(nb), dim3(256), 0, st, Z, diag_work, pval[m & 1], pidx[m & 1],
                       pval[(m + 1) & 1], pidx[(m + 1) & 1], nb, (int)N, ldz, ncols, m, scale, d0, gp, G, Lt);
  for (int m = ski_fast ? rank : 0; m < rank; ++m) {
    hipLaunchKernelGGL(pivchol_step_kernel, dim3(nb), dim3(256), 0, st, Z, L, diag_work, pval[m & 1], pidx[m & 1],
                       pval[(m + 1) & 1], pidx[(m + 1) & 1], nb, (int)N, ldz, ncols, rank, m, scale, d0, kind, group,
                       ncomp, wts, gp, G, (m == 0 && !gp) ? 1 : 0, Lt);
  }
  if (Lt)
    hipLaunchKernelGGL(pivchol_untranspose_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, Lt, L, (int)N, rank);
  return launch_status();
}

int rpgp_pivoted_cholesky(const float *Z, float *L, float *diag_work, int64_t N, int ldz, int J, int rank, float scale,
                          void *stream) {
  if (!Z || !L || !diag_work || N <= 0 || J <= 0 || J > 64 || rank <= 0 || rank > 64 || ldz < J || N > 0x7fffffffLL)
    return RPGP_EINVAL;
  return pivchol_common(Z, L, diag_work, N, ldz, J, rank, scale, scale * (float)J, RPGP_KIND_RBF, 1, J, nullptr, stream);
}

// the SKI factor is built column-major from this size on (below it the factor sits in L2 and the row-major form is as fast)
static const int64_t kPivcholColumnMajorMin = 65536;

size_t rpgp_ski_pivoted_cholesky_work_floats(int64_t N, int rank) {
  if (N <= 0 || rank <= 0) return 0;
  return (size_t)N + RPGP_PIVCHOL_SCRATCH + (N >= kPivcholColumnMajorMin ? (size_t)N * (size_t)rank : 0);
}

int rpgp_ski_pivoted_cholesky(const float *Z, const float *grid_params, float *L, float *diag_work, size_t diag_work_floats,
                              int64_t N, int ldz, int J, int G, int rank, float scale, void *stream) {
  if (!Z || !grid_params || !L || !diag_work || N <= 0 || J <= 0 || J > 64 || G < 8 || rank <= 0 || rank > 64 ||
      ldz < J || N > 0x7fffffffLL)
    return RPGP_EINVAL;
  if (diag_work_floats < rpgp_ski_pivoted_cholesky_work_floats(N, rank)) return RPGP_EWORKSPACE;
  float *Lt = N >= kPivcholColumnMajorMin ? diag_work + N + RPGP_PIVCHOL_SCRATCH : nullptr;
  return pivchol_common(Z, L, diag_work, N, ldz, J, rank, scale, scale * (float)J, RPGP_KIND_RBF, 1, J, nullptr, stream,
                        grid_params, G, Lt);
}

int rpgp_family_pivoted_cholesky(const rpgp_family *fam, const float *Z, float *L, float *diag_work, int64_t N,
                                 int ldz, int rank, float scale, float weight_sum, void *stream) {
  if (!Z || !L || !diag_work || N <= 0 || rank <= 0 || rank > 64 || N > 0x7fffffffLL) return RPGP_EINVAL;
  const int rc = family_check(fam, ldz, ldz);
  if (rc) return rc;
  const int ncols = fam->ncomp * fam->group;
  if (ncols > 64) return RPGP_EINVAL;
  return pivchol_common(Z, L, diag_work, N, ldz, ncols, rank, scale, scale * weight_sum, fam->kind, fam->group,
                        fam->ncomp, fam->weights, stream);
}

size_t rpgp_symcache_bytes(int64_t N, int world, int rank) {
  if (N <= 0 || N > 0x7fffffffLL || world < 1 || rank < 0 || rank >= world) return 0;
  return symk_bytes(symk_plan(N, world, rank));          // (the layout's R depends on (N, world) only: one size for both layouts)
}

size_t rpgp_symcache_workspace_bytes(int64_t N, int T, int world, int rank) {
  if (N <= 0 || N > 0x7fffffffLL || T <= 0 || world < 1 || rank < 0 || rank >= world) return 0;
  (void)rpgp_init();                               // (the wide plan counts the device's CUs; harmless without a device)
  const size_t a = plan_workspace_floats(symk_plan(N, world, rank, false).p, N, T, true);
  const size_t b = plan_workspace_floats(symk_plan(N, world, rank, true).p, N, T, true);        // (chunked differently)
  return (a > b ? a : b) * sizeof(float);
}

int rpgp_symcache_build(const float *Z, void *cache, size_t cache_bytes, int64_t N, int ldz, int j0, int j1, int layout,
                        int world, int rank, void *stream) {
  if (!Z || !cache || N <= 0 || N > 0x7fffffffLL || j0 < 0 || j1 <= j0 || ldz < j1 || world < 1 || rank < 0 ||
      rank >= world || (layout != RPGP_SYMCACHE_THIN && layout != RPGP_SYMCACHE_WIDE))
    return RPGP_EINVAL;
  const int irc = rpgp_init();
  if (irc) return irc;
  SymkPlan sp = symk_plan(N, world, rank);
  if (cache_bytes < symk_bytes(sp)) return RPGP_EWORKSPACE;
  if (world <= 1) {
    // The build of a whole cache is chunked on its own (the layout depends on (N, R) only): a compute-bound sweep of equal
    // workgroups at eight per CU wants MANY of them — ~6000 instead of the product's N / 14 (tools/experiments/
    // r5_symk_build_wgs.py, profiles/r5b_symk_build_wgs.jsonl: N = 7k 93.6 -> 72.8 us, 15k 267 -> 245, 28k 930 -> 834, 50k 2796 -> 2739)
    sp.p = make_plan(N, N, true, 12, world, rank, symk_r1(N, world), 0, 6144.0);
    sp.sub0 = symk_wg_subtile(sp.p, N, sp.p.w0);
    sp.sub1 = symk_wg_subtile(sp.p, N, sp.p.w1);
  }
  if (sp.p.w1 <= sp.p.w0) return 0;
  hipStream_t st = as_stream(stream);
  float4v *c = reinterpret_cast<float4v *>(cache);
  int first = 1;
  for (int j = j0; j < j1;) {
    const int jt = next_j_piece(j1 - j);
    int rc = 0;
    if (layout == RPGP_SYMCACHE_WIDE) {
      switch (jt) {
        case 20: rc = symk_launch_build_tile<20>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
        case 10: rc = symk_launch_build_tile<10>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
        case 8: rc = symk_launch_build_tile<8>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
        case 5: rc = symk_launch_build_tile<5>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
        case 4: rc = symk_launch_build_tile<4>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
        case 3: rc = symk_launch_build_tile<3>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
        case 2: rc = symk_launch_build_tile<2>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
        default: rc = symk_launch_build_tile<1>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
      }
    } else
    switch (jt) {
      case 20: rc = symk_launch_build<20>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
      case 10: rc = symk_launch_build<10>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
      case 8: rc = symk_launch_build<8>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
      case 5: rc = symk_launch_build<5>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
      case 4: rc = symk_launch_build<4>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
      case 3: rc = symk_launch_build<3>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
      case 2: rc = symk_launch_build<2>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
      default: rc = symk_launch_build<1>(sp, Z, c, (int)N, ldz, j, first ? 0 : 1, st); break;
    }
    if (rc) return rc;
    first = 0;
    j += jt;
  }
  return 0;
}

int rpgp_symcache_mvm(const void *cache, size_t cache_bytes, int layout, const float *V, float *out, int64_t N, int T,
                      float scale, float noise, int world, int rank, void *ws, size_t ws_bytes, void *stream) {
  if (!cache || !V || !out || N <= 0 || N > 0x7fffffffLL || T <= 0 || world < 1 || rank < 0 || rank >= world ||
      (layout != RPGP_SYMCACHE_THIN && layout != RPGP_SYMCACHE_WIDE))
    return RPGP_EINVAL;
  const int irc = rpgp_init();
  if (irc) return irc;
  const SymkPlan sp = symk_plan(N, world, rank, layout == RPGP_SYMCACHE_WIDE);
  if (cache_bytes < symk_bytes(sp)) return RPGP_EINVAL;
  const TilePlan &p = sp.p;
  const size_t need = plan_workspace_floats(p, N, T, true) * sizeof(float);
  if (need && (!ws || ws_bytes < need)) return RPGP_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  float *slabR = reinterpret_cast<float *>(ws);
  float *slabT = slabR + (size_t)p.maxchunks * p.rows * T;
  if (p.partial && need) RPGP_CHECK(hipMemsetAsync(ws, 0, need, st));   // row blocks shared with a neighbour rank
  const float4v *c = reinterpret_cast<const float4v *>(cache);
  for (int t0 = 0; layout == RPGP_SYMCACHE_WIDE && t0 < T && p.w1 > p.w0; t0 += 16) {
    const int tcnt = (T - t0 < 16) ? T - t0 : 16;
    const int rc = symk_launch_mvm_tile(sp, c, V, slabR, slabT, (int)N, T, t0, tcnt, st);
    if (rc) return rc;
  }
  for (int t0 = 0; layout == RPGP_SYMCACHE_THIN && t0 < T && p.w1 > p.w0;) {
    const int tt = symk_t_piece(T - t0);
    const int tcnt = (T - t0 < tt) ? T - t0 : tt;
    int rc = 0;
    switch (tt) {
      case 1: rc = symk_launch_mvm<1>(sp, c, V, slabR, slabT, (int)N, T, t0, tcnt, st); break;
      case 4: rc = symk_launch_mvm<4>(sp, c, V, slabR, slabT, (int)N, T, t0, tcnt, st); break;
      default: rc = symk_launch_mvm<12>(sp, c, V, slabR, slabT, (int)N, T, t0, tcnt, st); break;
    }
    if (rc) return rc;
    t0 += tcnt;
  }
  const size_t total = (size_t)N * T;
  RPGP_LAUNCH_MVM_REDUCE(total, st, slabR,
                     slabT, V, out, (int)N, (int)N, T, p.BR, p.chunk_cols, 1, scale, noise, (const int *)nullptr, p.rb0,
                     p.rb1, p.row0, p.rows);
  return launch_status();
}

namespace {
// resident workgroups per CU of dense_gemv_valu_kernel<T>, asked of the runtime once per T (register and LDS budget of
// the instantiation hipcc actually produced), capped at 8
int gemv_blocks_per_cu(int T) {
  static std::atomic<int> cache[13];
  const int tt = T < 1 ? 1 : (T > 12 ? 12 : T);
  int v = cache[tt].load(std::memory_order_relaxed);
  if (v > 0) return v;
  int nb = 0;
  hipError_t e = hipSuccess;
#define RPGP_OCC_CASE(TT_)                                                                                             \
  case TT_:                                                                                                            \
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dense_gemv_valu_kernel<TT_>, 256, 0);                         \
    break
  switch (tt) {
    RPGP_OCC_CASE(1); RPGP_OCC_CASE(2); RPGP_OCC_CASE(3); RPGP_OCC_CASE(4); RPGP_OCC_CASE(5); RPGP_OCC_CASE(6);
    RPGP_OCC_CASE(7); RPGP_OCC_CASE(8); RPGP_OCC_CASE(9); RPGP_OCC_CASE(10); RPGP_OCC_CASE(11);
    default: RPGP_OCC_CASE(12);
  }
#undef RPGP_OCC_CASE
  if (e != hipSuccess || nb < 1) nb = 4;
  if (nb > 8) nb = 8;
  cache[tt].store(nb, std::memory_order_relaxed);
  return nb;
}
}  // namespace

int rpgp_dense_mvm(const float *Kd, const float *V, float *out, int64_t N, int64_t ldk, int T, float noise,
                   void *stream) {
  if (!Kd || !V || !out || N <= 0 || T <= 0 || ldk < N || N > 0x7fffffffLL) return RPGP_EINVAL;
  {
    const int irc = rpgp_init();
    if (irc) return irc;
  }
  hipStream_t st = as_stream(stream);
  float *slab = nullptr;
  int rc = 0;
  if (T <= 12) {
    // VALU form: 256 output columns per workgroup, the rows split over blockIdx.y (and over the four waves inside).
    // Every workgroup streams the same amount of data and they all start together, so the launch must be ONE resident
    // round: a grid a few percent over the resident capacity runs two rounds and doubles the time (N = 14 939, T = 11: 1560
    // workgroups on 1536 slots).  Capacity per CU from the register budget of the instantiation: <= 64 VGPRs (T <= 2) 8
    // workgroups, T <= 8 six, wider five (a conservative count of what hipcc allocates for them).
    const unsigned ncb = (unsigned)((N + 255) / 256);
    const int per_cu = gemv_blocks_per_cu(T);
    const long long resident = (long long)g_num_cus * per_cu;
    long long nsplit = resident / ncb;
    const long long cap_bytes = (long long)(0.15 * (double)N / T);      // slab bytes <= 15 % of the matrix bytes
    if (nsplit > cap_bytes) nsplit = cap_bytes;
    if (nsplit > (N + 63) / 64) nsplit = (N + 63) / 64;
    if (nsplit > 160) nsplit = 160;
    if (nsplit < 1) nsplit = 1;
    int cps = (int)((N + nsplit - 1) / nsplit);
    cps = (cps + 31) / 32 * 32;                        // whole 8-row batches for each of the four waves
    nsplit = (N + cps - 1) / cps;
    const size_t npad = ((size_t)N + 3) & ~(size_t)3;
    RPGP_CHECK(hipMallocAsync((void **)&slab, (size_t)nsplit * npad * T * sizeof(float), st));
    dim3 vgrid(ncb, (unsigned)nsplit), block(256);
#define RPGP_GEMV_CASE(TT_)                                                                                            \
  case TT_:                                                                                                            \
    hipLaunchKernelGGL((dense_gemv_valu_kernel<TT_>), vgrid, block, 0, st, Kd, V, slab, (int)N, (long long)ldk, cps);   \
    break
    switch (T) {
      RPGP_GEMV_CASE(1); RPGP_GEMV_CASE(2); RPGP_GEMV_CASE(3); RPGP_GEMV_CASE(4); RPGP_GEMV_CASE(5); RPGP_GEMV_CASE(6);
      RPGP_GEMV_CASE(7); RPGP_GEMV_CASE(8); RPGP_GEMV_CASE(9); RPGP_GEMV_CASE(10); RPGP_GEMV_CASE(11);
      default: RPGP_GEMV_CASE(12);
    }
#undef RPGP_GEMV_CASE
    hipLaunchKernelGGL(dense_gemv_reduce_kernel, dim3((unsigned)((N + 31) / 32), (unsigned)T), dim3(256), 0, st, slab, V,
                       out, (int)N, T, (int)nsplit, noise);
    rc = launch_status();
    (void)hipFreeAsync(slab, st);
    return rc;
  }
  const int nrb = (int)((N + 255) / 256);
  const int ncb_ = (int)((N + 255) / 256);
  int nsplit = (3072 + ncb_ - 1) / ncb_;          // ~3000 workgroups keep enough loads in flight to stream HBM
  const int max_split = (int)((N + 255) / 256);
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit > 32) nsplit = 32;
  if (nsplit < 1) nsplit = 1;
  int cps = (int)((N + nsplit - 1) / nsplit);
  cps = (cps + 255) / 256 * 256;
  nsplit = (int)((N + cps - 1) / cps);
  RPGP_CHECK(hipMallocAsync((void **)&slab, (size_t)nsplit * N * 16 * sizeof(float), st));
  dim3 grid((unsigned)nrb, (unsigned)nsplit), block(256);
  for (int t0 = 0; t0 < T && rc == 0; t0 += 16) {
    const int tcnt = (T - t0 < 16) ? T - t0 : 16;
    hipLaunchKernelGGL(dense_gemm_kernel, grid, block, 0, st, Kd, V, slab, (int)N, (long long)ldk, T, t0, tcnt, cps);
    const size_t total = (size_t)N * 16;
    hipLaunchKernelGGL(dense_gemm_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, slab, V, out,
                       (int)N, T, t0, tcnt, nsplit, noise, 16);
    rc = launch_status();
  }
  (void)hipFreeAsync(slab, st);
  return rc;
}

}  // extern "C"
