// rpgp_comm.hip — one-shot / two-shot SUM all-reduce over IPC-mapped peer buffers (SURVEY.md §8(f) rank 3, §5
// "Distributed communication backend").  The messages of the sharded solve are small (the N x T partial product of one
// MVM: 200 KB at T = 1, 2.2 MB at T = 11; the J x G x T grid histogram of the row-sharded SKI operator: 270 KB; a few
// hundred bytes of inner products), i.e. latency-bound: a ring all-reduce pays 2 (W - 1) hops, here every rank reads its
// peers' staging buffers directly over xGMI (point-to-point links, one hop) in ONE kernel launch.
//
// Every rank owns one fine-grained device allocation  [flags | stage 0 | stage 1 | result 0 | result 1]  exported with
// hipIpcGetMemHandle and mapped by all peers (hipIpcOpenMemHandle).  Call number s (parity b = s & 1):
//   one-shot  (<= kTwoShotBytes):  copy buf -> my stage[b]; system-scope release; write s into flagsA[me] of every peer;
//                                  wait until my flagsA[p] >= s for all p; buf[i] = sum_p stage_p[b][i] in RANK ORDER
//                                  (every rank adds the same numbers in the same order: bit-identical results, which
//                                  the replicated CG recurrences rely on).
//   two-shot  (larger):            same publish step; rank r reduces ITS 1/W chunk from all peers' stages (rank order) and
//                                  writes the reduced chunk into every peer's result[b]; flagsB; wait; copy result -> buf.
//                                  Each link carries 2/W of the message instead of all of it.
// A stage may be overwritten two calls later: by then every peer has published call s + 1, which it does only after it
// finished reading call s.  Waits are bounded (kTimeoutTicks): a lost peer sets the comm's error word instead of hanging
// the GPU AND the kernel writes NaN into the caller's buffer, so the result can never pass for a sum; the host side reports
// the error word after every sharded solve / MLL evaluation (Reducer.check).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#include "../../include/rpgp.h"

namespace {

constexpr int kMaxWorld = 16;
constexpr size_t kFlagBytes = 4096;
constexpr size_t kTwoShotBytes = 512 * 1024;          // messages above this take the two-shot path
constexpr long long kTimeoutTicks = 20LL * 100000000LL;   // 20 s of the 100 MHz wall clock (s_memrealtime)
constexpr int kMaxBlocks = 128;

struct FlagBlock {                  // at offset 0 of every rank's exported buffer
  uint32_t a[kMaxWorld];            // a[p]: last call number peer p has published its stage for
  uint32_t b[kMaxWorld];            // b[p]: last call number peer p has delivered its reduced chunk for (two-shot)
  uint32_t error;                   // set by a wait that timed out
};

}  // namespace

struct rpgp_comm {
  int world, rank;
  size_t max_bytes;                 // capacity of one stage / result buffer
  char *local;                      // this rank's exported allocation
  char *peer[kMaxWorld];            // mapped base pointers (peer[rank] == local)
  char **peer_dev;                  // the same table in device memory
  unsigned *counters;               // device: two "blocks done" counters
  hipIpcMemHandle_t handle;
  uint32_t seq;
  bool connected;
};

namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__device__ __forceinline__ void publish(char *const *peers, int world, int rank, size_t flag_off, uint32_t seq) {
  for (int p = threadIdx.x; p < world; p += blockDim.x) {
    uint32_t *f = reinterpret_cast<uint32_t *>(peers[p] + flag_off) + rank;
    __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// every thread of the block returns once all peers' flags (in THIS rank's memory) have reached seq; false when a wait ran
// into its bound (the error word is set; the caller poisons its output so that a missed host-side check cannot pass)
__device__ __forceinline__ bool wait_all(char *mine, int world, size_t flag_off, uint32_t seq) {
  __shared__ int timed_out;
  if (threadIdx.x == 0) timed_out = 0;
  __syncthreads();
  if ((int)threadIdx.x < world) {
    uint32_t *f = reinterpret_cast<uint32_t *>(mine + flag_off) + threadIdx.x;
    const long long t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      if (wall_clock64() - t0 > kTimeoutTicks) {
        reinterpret_cast<FlagBlock *>(mine)->error = 1;
        timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");        // system scope: peers' data written before their flag is visible
  const bool ok = timed_out == 0;
  __syncthreads();                                     // (the flag is reused by the next wait of the same launch)
  return ok;
}

template <typename T>
__device__ __forceinline__ void poison(T *buf, size_t count, size_t gtid, size_t gsz) {
  const T nan = (T)__builtin_nanf("");
  for (size_t i = gtid; i < count; i += gsz) buf[i] = nan;
}

// the last workgroup to arrive publishes; `counter` returns to 0 for the next call
__device__ __forceinline__ void arrive_and_publish(unsigned *counter, char *const *peers, int world, int rank,
                                                   size_t flag_off, uint32_t seq) {
  __shared__ int last;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");         // system scope: this thread's stores before the flag
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = prev == gridDim.x - 1;
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (last) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
    publish(peers, world, rank, flag_off, seq);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void allreduce_oneshot_kernel(char *const *peers, int world, int rank, size_t stage_off,
                                                                T *buf, size_t count, uint32_t seq, unsigned *counter) {
  char *mine = peers[rank];
  T *stage = reinterpret_cast<T *>(mine + stage_off);
  const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
  for (size_t i = gtid; i < count; i += gsz) stage[i] = buf[i];
  arrive_and_publish(counter, peers, world, rank, offsetof(FlagBlock, a), seq);
  if (!wait_all(mine, world, offsetof(FlagBlock, a), seq)) {
    poison(buf, count, gtid, gsz);          // a peer never published: NaN, not a sum of stale stages
    return;
  }
  for (size_t i = gtid; i < count; i += gsz) {
    T s = reinterpret_cast<const T *>(peers[0] + stage_off)[i];
    for (int p = 1; p < world; ++p) s += reinterpret_cast<const T *>(peers[p] + stage_off)[i];
    buf[i] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void allreduce_twoshot_kernel(char *const *peers, int world, int rank, size_t stage_off,
                                                                size_t result_off, T *buf, size_t count, uint32_t seq,
                                                                unsigned *counters) {
  char *mine = peers[rank];
  T *stage = reinterpret_cast<T *>(mine + stage_off);
  const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
  for (size_t i = gtid; i < count; i += gsz) stage[i] = buf[i];
  arrive_and_publish(counters, peers, world, rank, offsetof(FlagBlock, a), seq);
  const bool ok_a = wait_all(mine, world, offsetof(FlagBlock, a), seq);
  // (after a timeout this rank still delivers its flag below so that the peers are not held for their own full bound; what
  //  it delivers is marked by NaN)
  // my chunk: a contiguous 1/world of the elements
  const size_t per = (count + world - 1) / world;
  const size_t c0 = per * rank < count ? per * rank : count, c1 = c0 + per < count ? c0 + per : count;
  for (size_t i = c0 + gtid; i < c1; i += gsz) {
    T s = reinterpret_cast<const T *>(peers[0] + stage_off)[i];
    for (int p = 1; p < world; ++p) s += reinterpret_cast<const T *>(peers[p] + stage_off)[i];
    if (!ok_a) s = (T)__builtin_nanf("");
    for (int p = 0; p < world; ++p) reinterpret_cast<T *>(peers[p] + result_off)[i] = s;
  }
  arrive_and_publish(counters + 1, peers, world, rank, offsetof(FlagBlock, b), seq);
  const bool ok_b = wait_all(mine, world, offsetof(FlagBlock, b), seq);
  if (!ok_a || !ok_b) {
    poison(buf, count, gtid, gsz);
    return;
  }
  const T *res = reinterpret_cast<const T *>(mine + result_off);
  for (size_t i = gtid; i < count; i += gsz) buf[i] = res[i];
}

}  // namespace

extern "C" {

int rpgp_comm_create(int world, int rank, size_t max_bytes, rpgp_comm **out, void *handle_out) {
  if (!out || !handle_out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || max_bytes == 0)
    return RPGP_EINVAL;
  static_assert(sizeof(hipIpcMemHandle_t) <= RPGP_COMM_HANDLE_BYTES, "handle size");
  static_assert(sizeof(FlagBlock) <= kFlagBytes, "flag block");
  rpgp_comm *c = new rpgp_comm();
  c->world = world;
  c->rank = rank;
  c->max_bytes = align256(max_bytes);
  c->seq = 0;
  c->connected = false;
  c->peer_dev = nullptr;
  c->counters = nullptr;
  const size_t total = kFlagBytes + 4 * c->max_bytes;
  void *p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, total, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { delete c; return (int)e; }
  c->local = reinterpret_cast<char *>(p);
  e = hipMemset(p, 0, kFlagBytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle(&c->handle, p);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->peer_dev), kMaxWorld * sizeof(char *));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->counters), 2 * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(c->counters, 0, 2 * sizeof(unsigned));
  if (e != hipSuccess) {
    if (c->peer_dev) (void)hipFree(c->peer_dev);
    if (c->counters) (void)hipFree(c->counters);
    (void)hipFree(p);
    delete c;
    return (int)e;
  }
  for (int i = 0; i < kMaxWorld; ++i) c->peer[i] = nullptr;
  c->peer[rank] = c->local;
  memset(handle_out, 0, RPGP_COMM_HANDLE_BYTES);
  memcpy(handle_out, &c->handle, sizeof(hipIpcMemHandle_t));
  *out = c;
  return 0;
}

int rpgp_comm_connect(rpgp_comm *c, const void *all_handles) {
  if (!c || !all_handles || c->connected) return RPGP_EINVAL;
  const char *hs = reinterpret_cast<const char *>(all_handles);
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, hs + (size_t)p * RPGP_COMM_HANDLE_BYTES, sizeof(h));
    void *ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return (int)e;
    c->peer[p] = reinterpret_cast<char *>(ptr);
  }
  hipError_t e = hipMemcpy(c->peer_dev, c->peer, kMaxWorld * sizeof(char *), hipMemcpyHostToDevice);
  if (e != hipSuccess) return (int)e;
  c->connected = true;
  return 0;
}

size_t rpgp_comm_capacity(const rpgp_comm *c) { return c ? c->max_bytes : 0; }

int rpgp_comm_allreduce(void *ctx, void *buf, size_t count, int dtype, void *stream) {
  rpgp_comm *c = reinterpret_cast<rpgp_comm *>(ctx);
  if (!c || !c->connected || !buf || (dtype != RPGP_F32 && dtype != RPGP_F64)) return RPGP_EINVAL;
  if (count == 0) return 0;
  const size_t esz = dtype == RPGP_F64 ? 8 : 4;
  const size_t bytes = count * esz;
  if (bytes > c->max_bytes) return RPGP_EWORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint32_t seq = ++c->seq;
  const size_t stage_off = kFlagBytes + (size_t)(seq & 1) * c->max_bytes;
  const size_t result_off = kFlagBytes + (size_t)(2 + (seq & 1)) * c->max_bytes;
  long long nb = (long long)((count + 1023) / 1024);
  if (nb > kMaxBlocks) nb = kMaxBlocks;
  if (nb < 1) nb = 1;
  const bool two = bytes > kTwoShotBytes && c->world > 2;
  if (dtype == RPGP_F64) {
    if (two)
      hipLaunchKernelGGL(allreduce_twoshot_kernel<double>, dim3((unsigned)nb), dim3(256), 0, st, c->peer_dev, c->world,
                         c->rank, stage_off, result_off, reinterpret_cast<double *>(buf), count, seq, c->counters);
    else
      hipLaunchKernelGGL(allreduce_oneshot_kernel<double>, dim3((unsigned)nb), dim3(256), 0, st, c->peer_dev, c->world,
                         c->rank, stage_off, reinterpret_cast<double *>(buf), count, seq, c->counters);
  } else {
    if (two)
      hipLaunchKernelGGL(allreduce_twoshot_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, c->peer_dev, c->world,
                         c->rank, stage_off, result_off, reinterpret_cast<float *>(buf), count, seq, c->counters);
    else
      hipLaunchKernelGGL(allreduce_oneshot_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, c->peer_dev, c->world,
                         c->rank, stage_off, reinterpret_cast<float *>(buf), count, seq, c->counters);
  }
  return (int)hipGetLastError();
}

int rpgp_comm_error(rpgp_comm *c, int *error_host) {
  if (!c || !error_host) return RPGP_EINVAL;
  uint32_t v = 0;
  hipError_t e = hipMemcpy(&v, c->local + offsetof(FlagBlock, error), sizeof(v), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return (int)e;
  *error_host = (int)v;
  return 0;
}

int rpgp_comm_destroy(rpgp_comm *c) {
  if (!c) return RPGP_EINVAL;
  (void)hipDeviceSynchronize();
  for (int p = 0; p < c->world; ++p)
    if (p != c->rank && c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
  if (c->peer_dev) (void)hipFree(c->peer_dev);
  if (c->counters) (void)hipFree(c->counters);
  if (c->local) (void)hipFree(c->local);
  delete c;
  return 0;
}

}  // extern "C"
