// Shared between the translation units of the C-ABI layer (vxba_capi*.hip): the helpers of namespace vxc that vxba_factor.hpp does not
// already declare.  Definitions: vxba_capi_core.hip.
#pragma once
#include "vxba_factor.hpp"
#include "vxba_wait.hpp"

#include <chrono>

namespace vxc {

using vxk::FactorView;
using vxk::PoseArg;

// one-shot peer all-reduce: workgroups per collective, mailbox = 2 slots of the buffer + 2 x PEER_WGS flags + 1 status word
constexpr int PEER_WGS = 8, PEER_THREADS = 256;

int n_planes(const vxba_factor* f);
vxw::WideView wview(const vxba_factor* f);
int ensure_staging(vxba_factor* f, size_t len);
int ensure_capacity(vxba_factor* f, int n_total);
int check_range(vxba_factor* f, int head, int end);
// the f32 re-centred copy of clusters [v0, ...) is stale after a write to the f64 planes
inline void clusters_written(vxba_factor* f, int v0) { if (v0 < f->cl32_built) f->cl32_built = v0; }

// Completion of the factor's stream at the end of a device-resident LM call: a 3-iteration damping_iter is ~0.2 ms of queued kernels, and
// waking up from hipStreamSynchronize costs ~25 us of that (measured in the map and LI shells), so the call polls -- for a bounded time:
// a stream that is still busy after ~4 ms (a long bench loop, a wide window) is handed to the blocking wait, which does not burn a core.
inline hipError_t stream_wait_spin(hipStream_t s) { return vxwait::stream_wait(s); }   // bounded polling, then the blocking wait: vxba_wait.hpp

// ---- profiling: hipEvents on the factor's stream around a launch, drained by vxba_get_kernel_times / vxba_get_collective_time ----
hipEvent_t get_event(vxba_factor* f);
int drain_events(vxba_factor* f);
struct ScopedKernelTimer {
  vxba_factor* f; int kind; int on = 0; hipEvent_t a = nullptr, b = nullptr;
  ScopedKernelTimer(vxba_factor* f_, int kind_) : f(f_), kind(kind_) {
    on = (f->profiling >> kind_) & 1;
    if (on) { a = get_event(f); b = get_event(f); if (a) hipEventRecord(a, f->stream); }
  }
  ~ScopedKernelTimer() {
    if (on && a && b) { hipEventRecord(b, f->stream); f->pending.push_back({a, b, kind}); }
  }
};

// ---- voxel-sharded factors ----
bool has_peer(const vxba_factor* f);
int shard_allreduce(vxba_factor* f, double* d_buf, size_t count);
bool spec_collective(const vxba_factor* f);
int spec_hess_phase(vxba_factor* f, const double* Rp0, int* c, bool first_of_solve, bool has_pending, bool restart, const double* cache_src, int k2_nparts);
int spec_final_decision(vxba_factor* f, const double* Rp0, int* c, int k2_nparts);

bool fused_solve(const vxba_factor* f);
bool fused_sweeps(const vxba_factor* f);
int sweep_fused_device(vxba_factor* f, vxk::LMState* lm, int* c, unsigned seq);
bool fused_sweeps_spec(const vxba_factor* f);
int spec_fused_phase(vxba_factor* f, const double* Rp0, int* c, unsigned seq, int* k2_nparts);
void options_from_env(vxba_factor* f);
int upload_poses(vxba_factor* f, const double* Rp);
// synchronous sweeps into the pinned host buffers (single-sweep entry points, wide windows)
int sweep_hess_host(vxba_factor* f, const double* Rp, int head, int end);
int sweep_residual_host(vxba_factor* f, const double* Rp, int head, int end, double* residual);
int append_meta(vxba_factor* f, int v0, int n, const double* fix, const double* coe, const double* eig_val, const double* eig_vec, const double* merged);

}  // namespace vxc
