// The factor object behind the C ABI (include/vxba.h) and what the translation units of the ABI share: vxba_capi.hip (storage, sweeps,
// the LiDAR-only LM shell, options, measurement) and vxba_capi_li.hip (the inertial half: IMU_PRE wrappers, the LiDAR-inertial shells).
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vxba.h"
#include "vxba_host.hpp"
#include "vxba_imu.hpp"
#include "vxba_voxelize.h"
#include "vxba_wide.h"
#include "vxba_scratch.hpp"
#include "vxba_internal.h"
#include "vxba_kernels.h"

using vxk::FactorView;
using vxk::PoseArg;

namespace vxc {

constexpr int N_META_PLANES = 10 + 1 + 3 + 9 + 10 + 4;  // fix, coe, eigval, eigvec, merged, aux
constexpr int N_CACHE_PLANES = 3 + 9 + 10 + 4;          // eigval, eigvec, merged, aux (contiguous at the tail)

struct EventPair { hipEvent_t a, b; int kind; };

}  // namespace vxc
using vxc::EventPair;
using vxc::N_META_PLANES;
using vxc::N_CACHE_PLANES;

struct vxba_factor {
  int W = 0, device = 0;
  int V = 0;        // voxels in the factor
  int VS = 0;       // plane stride (capacity, multiple of 64)
  int cus = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;
  double* planes = nullptr;      // [(10W + N_META_PLANES)][VS]
  double* clb = nullptr;         // batch-major copy of the clusters for the Hessian sweep
  float* cl32 = nullptr;         // VXBA_PRECISION_MIXED_F32_CLUSTERS: f32 re-centred copy of the cluster planes for the residual sweep (built on demand)
  int cl32_vs = 0;               // voxel stride it was allocated for
  int cl32_built = 0;            // voxels [0, cl32_built) are converted (the factor is append-only between clears)
  double* snapshot = nullptr;    // [N_CACHE_PLANES][snapshot_vs]
  int snapshot_vs = 0, snapshot_v = 0;
  double* staging = nullptr;     // device scratch for uploads / read-backs
  size_t staging_len = 0;
  double* d_partial3 = nullptr;  // K3 workgroup partials
  size_t partial3_len = 0;
  double* d_partial2 = nullptr;  // K2 wave partials
  double* h_partial2 = nullptr;  // the same in mapped host memory (LI shells: the host adds the partials up itself), zc_partial2 = its device address
  double* zc_partial2 = nullptr;
  size_t partial2_len = 0;
  double* d_packed = nullptr;    // [Hess | JacT | residual] (points at own_packed or a caller buffer)
  double* d_scalar = nullptr;
  double* own_packed = nullptr;
  double* own_scalar = nullptr;
  unsigned long long* d_count = nullptr;
  double* h_packed = nullptr;    // pinned, mapped
  double* zc_packed = nullptr;   // device alias of h_packed: kernels of host-driven loops write their result straight into host memory
  double* h_scalar = nullptr;    // pinned
  vxk::LMState* d_lm = nullptr;  // device-resident LM shell state
  char* d_scratch = nullptr;     // grow-only device scratch of the batch factor construction (staging of points and accepted voxels)
  size_t scratch_cap = 0;
  vxk::LMState* h_lm = nullptr;  // pinned read-back copy
  vxw::DenseSolver* wide_solver = nullptr;   // wide windows: device Cholesky of the (6W)-dimensional LM step (vxba_wide.hip)
  bool wide_solver_tried = false;
  bool li_reduction_in_flight = false;   // a speculative Hessian sweep + reduction of the last LI call may still be writing h_packed
  double li_last_call_us = 0;    // VXBA_STAT_LI_LAST_CALL_US
  double li_wait_us = 0;         // development (VXBA_LI_TIMING): time the LI shells spent waiting for the Hessian sweep
  struct LiScratch {             // host buffers of the LI shells, kept between calls (four (15W)^2 matrices: allocating and zeroing them
    std::vector<double> Hess, HessN, A, JacT, JacTN, D, rhs, dxi, work, cov_invs;   // cost ~15 us of a ~300 us call)
    std::vector<int> perm;
    // information matrices of the previous call, with the covariances they were computed from: consecutive calls of a sliding window
    // see the same preintegrated factors again (in place, or moved down by the frames marginalised in between)
    std::vector<double> cov_seen, cov_inv_seen;
    int n_seen = 0;
    void size(int n, int nfac) {
      Hess.resize((size_t)n * n); HessN.resize((size_t)n * n); JacT.resize(n); JacTN.resize(n); D.resize(n); rhs.resize(n); dxi.resize(n);
      work.resize(n); perm.resize(n); cov_invs.resize((size_t)225 * nfac);
    }
  } li;
  vxh::BandSchurWork li_bs;
  int opt[VXBA_OPT_COUNT] = {1, 1, 1, 0, 64, 0, 1, 1, 1, 0};   // vxba_set_option; initial values may come from the environment (see vxba.h)
  vxw::WideStore wstore;         // wide windows: the clusters, compressed rows over the observed (voxel, frame) entries (no cluster planes)
  vxw::WideIndex wide;           // wide windows: incidence structure (entries, entry pairs per Hessian block), rebuilt after a push
  bool wide_dirty = true;
  double* d_poses = nullptr;     // wide windows: W*12 poses on the device (the MFMA kernels take them by value)
  double* h_poses = nullptr;     // pinned staging for the above: a ring of POSE_SLOTS slots, each guarded by an event
  hipEvent_t pose_ev[8] = {};
  unsigned pose_slot = 0;
  size_t xlen = 0;               // doubles the exchange buffers (own_packed, h_packed) hold
  int precision = 0;             // 0: fp64 throughout; 1: Hessian products in f32 on the matrix cores, f64 accumulation; 2: 1 + f32 re-centred cluster rows in the residual sweep
  unsigned lm_seq = 0;           // sequence numbers of solves published inside residual-sweep launches (never 0)
  hipEvent_t li_ev3 = nullptr;
  hipEvent_t li_ev2 = nullptr;   // ... and the end of a Hessian sweep + reduction queued ahead (queued-sweeps mode: the stream never drains)
  double* h_feed = nullptr;      // mapped host memory through which the LI shells hand the trial poses to a residual sweep that is already queued: [seq | 12 W poses]
  double* zc_feed = nullptr;
  double* h_packed2 = nullptr;   // second mapped buffer for the reduced systems of the LI shell's device-solve mode (consecutive systems alternate: the device never waits
  double* zc_packed2 = nullptr;  // for the host there, so the next reduction must not land in the buffer the host is still reading)
  double* h_lirec = nullptr;     // mapped host memory: the reduced pose system of a LiDAR-inertial step for the in-launch solve [u | poses | e | E] (vxba_solve4.hpp)
  double* zc_lirec = nullptr;
  double* lirec_vram = nullptr;  // the same record in fine-grained device memory, written by the host through the BAR (round 4); null: read from zc_lirec
  double* h_liout = nullptr;     // mapped host memory: that solve's answer [dx 6W | trial poses 12W | seq]
  double* zc_liout = nullptr;
  hipEvent_t li_ev = nullptr;    // marks the end of the residual sweep when a speculative Hessian sweep is queued behind it (LI host shells)
  bool solve_timed_out = false;  // the last damping_iter failed because voxel workgroups gave up waiting for the in-launch solve
  int fused_fallbacks = 0;       // times a call was transparently re-run with the solve as its own launch
  bool reject_heavy = false;     // the last device-resident LM call rejected more than a third of its steps: the next one runs the three-launch
                                 // iteration (a fused launch speculates that its step is accepted and pays a whole Hessian half for a rejected one)
  int li_dev_fallbacks = 0;      // times the LI shell discarded a non-finite / undelivered in-launch pose step and solved on the host (VXBA_STAT_LI_DEVICE_FALLBACKS)
  vxba_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  // direct RCCL path: entry points resolved from the librccl.so the process already uses
  void* rccl_lib = nullptr;
  ncclComm_t rccl_comm = nullptr;
  ncclResult_t (*p_ncclAllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*p_ncclCommDestroy)(ncclComm_t) = nullptr;
  // one-shot peer all-reduce over xGMI (vxba_peer_*): every rank's mailbox mapped into every other rank through hipIpc
  struct Peer {
    int nranks = 0, rank = 0;
    size_t len = 0;                       // doubles per mailbox slot
    double* box = nullptr;                // own mailbox: [2][len] f64 + flags [2][PEER_WGS] u64 + status u64 (fine-grained device memory)
    void* opened[VXBA_PEER_MAX] = {};     // hipIpcOpenMemHandle results (own entry stays null)
    double* boxes[VXBA_PEER_MAX] = {};    // mailbox of rank p as seen from this process
    unsigned long long seq = 0;
  } peer;
  int profiling = 0;             // bit mask of kernel kinds to bracket with events: 1 K3, 2 K2, 4 K3 finalize, 8 K1, 16 all-reduce
  std::vector<EventPair> pending;
  std::vector<hipEvent_t> free_events;
  double ms_sum[6] = {0, 0, 0, 0, 0, 0};   // [4]: the all-reduce of a sharded factor (profiling bit 16); [5]: the fused residual + Hessian launch (bit 32)
  int64_t calls[6] = {0, 0, 0, 0, 0, 0};
  std::string err;
  // The reference calls the two sweeps from several std::threads on one LidarFactor with disjoint [head,end)
  // (voxel_map.hpp:318-332); entry points serialise on this lock so such callers stay correct.
  std::recursive_mutex mtx;
};

#define VX_LOCK(f) std::unique_lock<std::recursive_mutex> lk__; if (f) lk__ = std::unique_lock<std::recursive_mutex>((f)->mtx)

#define VX_HIP(f, call)                                                                              \
  do {                                                                                               \
    hipError_t e__ = (call);                                                                         \
    if (e__ != hipSuccess) {                                                                         \
      (f)->err = std::string(#call) + ": " + hipGetErrorString(e__);                                 \
      return VXBA_ERR_HIP;                                                                           \
    }                                                                                                \
  } while (0)
#define VX_NARROW_ONLY(f, what) \
  do { if (vxc::is_wide(f)) return vxc::fail(f, VXBA_ERR_UNSUPPORTED, what ": only for win_size <= VXBA_MAX_WIN"); } while (0)

namespace vxc {
// helpers defined in vxba_capi.hip
int fail(vxba_factor* f, int code, const char* msg);
bool is_wide(const vxba_factor* f);
bool has_collective(const vxba_factor* f);
vxk::FactorView view(const vxba_factor* f);
// view for a residual sweep: under VXBA_PRECISION_MIXED_F32_CLUSTERS the f32 re-centred cluster copy is brought up to date (on f->stream) and attached
int residual_view(vxba_factor* f, vxk::FactorView& fv);
int ensure_exchange(vxba_factor* f);
int ensure_partials3(vxba_factor* f);
void fill_poses(const vxba_factor* f, const double* Rp, vxk::PoseArg& pa);
// asynchronous sweeps on f->stream, results in device (or mapped host) memory
int sweep_hess_device(vxba_factor* f, const double* Rp, vxk::LMState* lm, int* c, const vxk::LMPending* pend, int head, int end, double* d_out,
                      const double* cache_src = nullptr);
int sweep_residual_device(vxba_factor* f, const double* Rp, vxk::LMState* lm, int c, int head, int end, double* d_out, int* nparts_out = nullptr,
                          unsigned fused_seq = 0, bool partials_to_host = false);
}  // namespace vxc
