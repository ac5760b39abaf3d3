// rpgp_trace.hip — roctx phase markers of librpgp.so (rpgp_internal::trace_push / trace_pop, C-ABI rpgp_range_push /
// rpgp_range_pop): the counterpart of the reference's `torch.profiler` / nvtx-free runs being readable by phase
// ("Tracing / profiling"; the loop they annotate is fitting/optimizing.py:65-76).
// The marker library (ROCm's librocprofiler-sdk-roctx, what `rocprofv3 --marker-trace` records) is bound at first use with
// dlopen(RTLD_NOLOAD): ONLY when it is already in the process — rocprofv3 preloads it when marker tracing is requested, an
// application that wants markers under another tool links or preloads it itself.  Otherwise every range call is a null test:
// an unprofiled step pays nothing, and librpgp.so loads on a machine without the library.
#include <dlfcn.h>
#include <atomic>

#include "../../include/rpgp.h"
#include "rpgp_internal.h"

namespace {
typedef int (*push_fn)(const char *);
typedef int (*pop_fn)(void);
struct Roctx {
  push_fn push = nullptr;
  pop_fn pop = nullptr;
  Roctx() {
    const char *names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so"};    // (not libroctx64: rocprofv3 does not record it)
    for (const char *n : names) {
      void *h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (!h) continue;
      push = reinterpret_cast<push_fn>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<pop_fn>(dlsym(h, "roctxRangePop"));
      if (push && pop) return;
      push = nullptr;
      pop = nullptr;
    }
  }
};
inline const Roctx &roctx() {
  static const Roctx r;          // (thread-safe one-time initialisation)
  return r;
}
}  // namespace

namespace rpgp_internal {
void trace_push(const char *name) {
  const Roctx &r = roctx();
  if (r.push && name) (void)r.push(name);
}
void trace_pop() {
  const Roctx &r = roctx();
  if (r.pop) (void)r.pop();
}
}  // namespace rpgp_internal

extern "C" {
int rpgp_range_push(const char *name) {
  if (!name) return RPGP_EINVAL;
  rpgp_internal::trace_push(name);
  return 0;
}
int rpgp_range_pop(void) {
  rpgp_internal::trace_pop();
  return 0;
}
int rpgp_range_available(void) { return roctx().push != nullptr ? 1 : 0; }
}
