// Completion of a stream at the end of a short device-resident call.  Waking up from hipStreamSynchronize costs 15-25 us -- as much as the
// kernels of a small call -- so the entry points poll; but a poll without a bound burns a core for as long as the GPU works, and under a cgroup
// CPU quota the polling threads throttle the whole container (round-5 advisor; profiles/r05_cfg5/host_threads_and_cgroup_quota.txt).  So:
// spin (with a pause between queries) for at most VXBA_SPIN_US microseconds -- 4000 by default: every call the polling was introduced for is done
// long before -- then hand the stream to the blocking wait.  VXBA_SPIN_US=0: never spin (an embedding application that shares its cores).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace vxwait {

inline long spin_budget_us() {
  static const long v = [] {
    const char* e = std::getenv("VXBA_SPIN_US");
    const long x = e ? std::atol(e) : 4000;
    return x < 0 ? 0 : x;
  }();
  return v;
}

inline hipError_t stream_wait(hipStream_t s) {
  const long budget = spin_budget_us();
  if (budget == 0) return hipStreamSynchronize(s);
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned k = 0;; k++) {
    const hipError_t q = hipStreamQuery(s);
    if (q != hipErrorNotReady) return q;
#if defined(__x86_64__)
    _mm_pause();
#endif
    if ((k & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(budget)) return hipStreamSynchronize(s);
  }
}

inline hipError_t event_wait(hipEvent_t e) {
  const long budget = spin_budget_us();
  if (budget == 0) return hipEventSynchronize(e);
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned k = 0;; k++) {
    const hipError_t q = hipEventQuery(e);
    if (q != hipErrorNotReady) return q;
#if defined(__x86_64__)
    _mm_pause();
#endif
    if ((k & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(budget)) return hipEventSynchronize(e);
  }
}

}  // namespace vxwait
