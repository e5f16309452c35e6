// C-ABI implementation, inertial half (include/vxba.h): IMU_PRE wrappers and the LiDAR-inertial LM shells (LI_BA_Optimizer,
// LI_BA_OptimizerGravity; host shells between the GPU sweeps, and the driver of the device-resident loop).
#include "vxba_wait.hpp"
#include "vxba_factor.hpp"

#include <atomic>
#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
static inline void vx_store_fence() { _mm_sfence(); }
#else
static inline void vx_store_fence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#endif
#include <limits>

using namespace vxc;

extern "C" {

// ---- inertial half (host) -------------------------------------------------------------------------------------
int vxba_imu_init(double* imu, const double* bg, const double* ba) {
  if (!imu) return VXBA_ERR_ARG;
  vxi::imu_init(imu, bg, ba);
  return VXBA_OK;
}
int vxba_imu_add(double* imu, const double* gyr, const double* acc, double dt, const double* noise_meas, const double* noise_walk) {
  if (!imu || !gyr || !acc || !noise_meas || !noise_walk) return VXBA_ERR_ARG;
  vxi::imu_add(imu, gyr, acc, dt, noise_meas, noise_walk);
  return VXBA_OK;
}
int vxba_imu_evaluate(const double* imu, const double* st1, const double* st2, int jac_enable, double* jtj, double* gg, double* residual) {
  if (!imu || !st1 || !st2 || !residual || (jac_enable && (!jtj || !gg))) return VXBA_ERR_ARG;
  vxi::ImuWork w;
  bool ok = true;
  *residual = vxi::imu_evaluate(imu, st1, st2, jac_enable != 0, jtj, gg, w, &ok);
  return ok ? VXBA_OK : VXBA_ERR_STATE;   // singular covariance (a factor without samples)
}
int vxba_imu_evaluate_g(const double* imu, const double* st1, const double* st2, int jac_enable, double* jtj, double* gg, double* residual) {
  if (!imu || !st1 || !st2 || !residual || (jac_enable && (!jtj || !gg))) return VXBA_ERR_ARG;
  vxi::ImuWork w;
  bool ok = true;
  *residual = vxi::imu_evaluate(imu, st1, st2, jac_enable != 0, jtj, gg, w, &ok, true);
  return ok ? VXBA_OK : VXBA_ERR_STATE;
}
int vxba_imu_update_state(double* imu, const double* dxi15) {
  if (!imu || !dxi15) return VXBA_ERR_ARG;
  vxi::imu_update_state(imu, dxi15);
  return VXBA_OK;
}
int vxba_hess_plus(int W, double* Hess15, double* JacT15, const double* Hess6, const double* JacT6) {
  if (W < 1 || !Hess15 || !JacT15 || !Hess6 || !JacT6) return VXBA_ERR_ARG;
  vxi::li_hess_plus(W, Hess15, JacT15, Hess6, JacT6);
  return VXBA_OK;
}

namespace {
void states_to_poses(int W, const double* states, double* Rp) {
  for (int i = 0; i < W; i++) std::memcpy(Rp + 12 * i, states + vxi::STATE_LEN * i, sizeof(double) * 12);   // [R | p] lead the state
}
// divide_thread: the Hessian sweep is queued first, the IMU blocks are built on the host while it runs
// completion of everything queued on the factor's stream, by polling: the sweeps are tens of microseconds, less than what waking
// up from hipStreamSynchronize costs
int wait_stream(vxba_factor* f) {
  hipError_t q;
  q = vxwait::stream_wait(f->stream);
  VX_HIP(f, q);
  return VXBA_OK;
}
// spec_queued: the Hessian sweep at exactly these states was queued speculatively behind the last residual sweep (li_joint_residual)
// and is running or done -- nothing to launch.  imu_ready: likewise the IMU half (see li_joint_residual); points to its residual.
// while_sweeping (optional): called once the IMU blocks are in Hess / JacT and before the host starts waiting for the sweep -- the
// LiDAR factor only adds to the pose-pose blocks afterwards, so everything else of the system is final at that point.
int li_joint_system(vxba_factor* f, const double* states, const double* imus, double imu_coef, double* Hess, double* JacT, double* residual,
                    bool with_g = false, const double* cov_invs = nullptr, bool spec_queued = false, const std::function<void()>* while_sweeping = nullptr,
                    const double* imu_ready = nullptr) {
  const int W = f->W, n = vxi::DIM * W + (with_g ? 3 : 0), m = 6 * W;
  std::vector<double> Rp(12 * W);
  states_to_poses(W, states, Rp.data());
  // single GPU: the reduction kernel writes the packed system straight into pinned host memory (no copy to enqueue) and the host
  // polls for completion after its own half of the work; with a collective the reduced device buffer is copied as before
  const bool zc = !has_collective(f);
  int rc = VXBA_OK;
  if (!(spec_queued && zc)) {
    rc = sweep_hess_device(f, Rp.data(), nullptr, nullptr, nullptr, 0, f->V, zc ? f->zc_packed : f->d_packed);
    if (rc) return rc;
  }
  if (!zc) VX_HIP(f, hipMemcpyAsync(f->h_packed, f->d_packed, vxba_packed_len(f) * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  vxi::ImuWork w;
  bool ok = true;
  double res;
  if (imu_ready) res = *imu_ready;   // Hess / JacT already hold the IMU blocks at these states (built during the last residual sweep)
  else {
    std::memset(Hess, 0, sizeof(double) * n * n);
    std::memset(JacT, 0, sizeof(double) * n);
    res = vxi::li_add_imu_blocks(W, states, imus, imu_coef, true, Hess, JacT, w, &ok, with_g, cov_invs);
  }
  if (while_sweeping && ok) (*while_sweeping)();
  const auto t_w0 = std::chrono::steady_clock::now();
  rc = wait_stream(f);
  f->li_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_w0).count();
  if (rc) return rc;
  if (!ok) return fail(f, VXBA_ERR_STATE, "li: singular IMU covariance (factor without samples?)");
  vxi::li_hess_plus(W, Hess, JacT, f->h_packed, f->h_packed + (size_t)m * m, n);
  *residual = res + f->h_packed[(size_t)m * m + m];
  return VXBA_OK;
}
// sum_partials_kernel on the host, same order (thread t of 1024 adds the partials t, t + 1024, ...; then the halving tree): bitwise the
// value the device reduction gives
double host_sum_partials(const double* p, int n) {
  double red[1024];
  for (int t = 0; t < 1024; t++) {
    double s = 0.0;
    for (int k = t; k < n; k += 1024) s += p[k];
    red[t] = s;
  }
  for (int off = 512; off > 0; off >>= 1)
    for (int t = 0; t < off; t++) red[t] += red[t + off];
  return red[0];
}
// speculate (single GPU only): the Hessian sweep of the NEXT iteration -- at these trial states, on the cache this residual sweep leaves
// -- is queued right behind it, before anybody knows whether the step will be accepted.  The host only waits for the residual (an
// event), takes the decision, and if the step is accepted finds the next joint system already under way instead of paying a cold
// launch and a round trip for it; a rejected step wastes the sweep (upstream recomputes nothing then either).
// next_Hess / next_JacT (with speculate): while the host would otherwise only poll for the sweep, it builds the IMU half of the NEXT joint
// system at these trial states (the inertial residual falls out of the same evaluation); *imu_residual receives the IMU part alone.
int li_joint_residual(vxba_factor* f, const double* states, const double* imus, double imu_coef, double* residual,
                      const double* cov_invs = nullptr, bool speculate = false, double* next_Hess = nullptr, double* next_JacT = nullptr, bool with_g = false,
                      double* imu_residual = nullptr) {
  const int W = f->W;
  std::vector<double> Rp(12 * W);
  states_to_poses(W, states, Rp.data());
  const bool zc = !has_collective(f);
  const size_t plen = vxba_packed_len(f);
  // single GPU: the sweep writes its block partials straight into mapped host memory and the host adds them up (one launch less)
  int nparts = 0;
  int rc = sweep_residual_device(f, Rp.data(), nullptr, 0, 0, f->V, zc ? f->zc_packed + plen : f->d_scalar, &nparts, 0, zc);
  if (rc) return rc;
  if (!zc) VX_HIP(f, hipMemcpyAsync(f->h_scalar, f->d_scalar, sizeof(double), hipMemcpyDeviceToHost, f->stream));
  const bool spec = speculate && zc;
  if (spec) {
    if (!f->li_ev) VX_HIP(f, hipEventCreateWithFlags(&f->li_ev, hipEventDisableTiming));
    VX_HIP(f, hipEventRecord(f->li_ev, f->stream));
    rc = sweep_hess_device(f, Rp.data(), nullptr, nullptr, nullptr, 0, f->V, f->zc_packed);
    if (rc) return rc;
  }
  vxi::ImuWork w;
  bool ok = true;
  double r1;
  if (spec && next_Hess && next_JacT) {
    const int n = vxi::DIM * W + (with_g ? 3 : 0);
    std::memset(next_Hess, 0, sizeof(double) * n * n);
    std::memset(next_JacT, 0, sizeof(double) * n);
    r1 = vxi::li_add_imu_blocks(W, states, imus, imu_coef, true, next_Hess, next_JacT, w, &ok, with_g, cov_invs);
  } else {
    r1 = vxi::li_add_imu_blocks(W, states, imus, imu_coef, false, nullptr, nullptr, w, &ok, false, cov_invs);
  }
  if (imu_residual) *imu_residual = r1;
  if (spec) {
    hipError_t q;
    q = vxwait::event_wait(f->li_ev);
    VX_HIP(f, q);
  } else {
    rc = wait_stream(f);
  }
  if (rc) return rc;
  if (!ok) return fail(f, VXBA_ERR_STATE, "li: singular IMU covariance (factor without samples?)");
  *residual = r1 + (zc ? host_sum_partials(f->h_partial2, nparts) : f->h_scalar[0]);
  return VXBA_OK;
}
}  // namespace

int vxba_li_evaluate(vxba_factor* f, const double* states, const double* imus, double imu_coef, double* Hess, double* JacT, double* residual) {
  VX_LOCK(f);
  if (!f || !states || (!imus && f->W > 1) || !Hess || !JacT || !residual) return fail(f, VXBA_ERR_ARG, "li_evaluate: null argument");
  if (f->V == 0) return fail(f, VXBA_ERR_STATE, "li_evaluate on an empty factor");
  VX_NARROW_ONLY(f, "li_evaluate");
  hipSetDevice(f->device);
  return li_joint_system(f, states, imus, imu_coef, Hess, JacT, residual);
}
// LI_BA_OptimizerGravity::divide_thread (voxel_map.hpp:673-736): the (15W + 3)-dimensional joint system, gravity at the tail
int vxba_li_evaluate_gravity(vxba_factor* f, const double* states, const double* imus, double imu_coef, double* Hess, double* JacT, double* residual) {
  VX_LOCK(f);
  if (!f || !states || (!imus && f->W > 1) || !Hess || !JacT || !residual) return fail(f, VXBA_ERR_ARG, "li_evaluate_gravity: null argument");
  if (f->V == 0) return fail(f, VXBA_ERR_STATE, "li_evaluate_gravity on an empty factor");
  VX_NARROW_ONLY(f, "li_evaluate_gravity");
  hipSetDevice(f->device);
  return li_joint_system(f, states, imus, imu_coef, Hess, JacT, residual, true);
}
// LI_BA_OptimizerGravity::hess_plus (voxel_map.hpp:663-671): the same scatter into a system with leading dimension 15W + 3
int vxba_hess_plus_gravity(int W, double* Hess, double* JacT, const double* Hess6, const double* JacT6) {
  if (W < 1 || !Hess || !JacT || !Hess6 || !JacT6) return VXBA_ERR_ARG;
  vxi::li_hess_plus(W, Hess, JacT, Hess6, JacT6, vxi::DIM * W + 3);
  return VXBA_OK;
}
int vxba_li_only_residual(vxba_factor* f, const double* states, const double* imus, double imu_coef, double* residual) {
  VX_LOCK(f);
  if (!f || !states || (!imus && f->W > 1) || !residual) return fail(f, VXBA_ERR_ARG, "li_only_residual: null argument");
  if (f->V == 0) return fail(f, VXBA_ERR_STATE, "li_only_residual on an empty factor");
  VX_NARROW_ONLY(f, "li_only_residual");
  hipSetDevice(f->device);
  return li_joint_residual(f, states, imus, imu_coef, residual);
}

// LI_BA_Optimizer::damping_iter (voxel_map.hpp:562-653): the voxel sweeps on the GPU, the 15W-dimensional shell on the host.

// ------------------------------------------------------------------------------------------------------------------------------
// The LiDAR-inertial shells with their sweeps QUEUED AHEAD (VXBA_OPT_LI_QUEUED_SWEEPS, single GPU, structured solve).
// Between "the reduced Hessian has arrived" and "the residual sweep starts" the plain shell has the host alone on the critical path:
// hess_plus, the dense half of the solve, the state update -- and then a launch whose kernel starts ~10 us later and spends its first
// 6-7 us loading cluster rows.  Here the residual sweep of an iteration, and the speculative Hessian sweep + reduction behind it, are
// already in the stream when the host starts solving: the sweep's workgroups load their rows and wait; its first workgroup polls a
// sequence number in mapped host memory, and when the host has written the trial poses there it copies them into the device-side
// control block and releases the others (the mechanism of the in-launch solve of the LiDAR-only loop, with the host as the solver);
// the Hessian sweep behind it reads the same control block (LMPending.pending == 2).  The host's API calls leave the critical path too.
// Everything else -- IMU blocks of the next system under the residual sweep, band half of the solve under the Hessian sweep, the
// reference's damping schedule, bias roll-back -- is the plain shell's.  Returns 1 if the mode does not apply (the caller then runs
// the plain shell), else a VXBA status.
// ------------------------------------------------------------------------------------------------------------------------------

// Information matrices of a window's IMU factors (vxi::li_invert_covariances), re-using those of the previous call: a factor whose 15 x 15
// covariance is bit-identical to one seen then (same slot, or any other: the window slides) gets the inverse computed then -- the same
// bits as a fresh inversion of the same input.  Nine LU inversions cost 18 us at the head of every call.
static bool li_information_matrices(vxba_factor* f, int W, const double* imus, double* cov_invs) {
  using namespace vxi;
  constexpr size_t CL = (size_t)DIM * DIM;
  auto& L = f->li;
  const int nfac = W - 1;
  double lu[DIM * DIM];
  int perm[DIM];
  bool ok = true;
  for (int i = 0; i < nfac; i++) {
    const double* cov = imus + (size_t)IMU_LEN * i + O_COV;
    int hit = -1;
    for (int d = 0; d < L.n_seen && hit < 0; d++) {
      const int j = (i + d) % L.n_seen;     // same slot first, then the ones the window may have slid from
      if (std::memcmp(cov, L.cov_seen.data() + CL * j, CL * sizeof(double)) == 0) hit = j;
    }
    if (hit >= 0) std::memcpy(cov_invs + CL * i, L.cov_inv_seen.data() + CL * hit, CL * sizeof(double));
    else ok = dm_inverse(DIM, cov, cov_invs + CL * i, lu, perm) && ok;
  }
  if (!ok) { L.n_seen = 0; return false; }
  L.cov_seen.resize(CL * nfac); L.cov_inv_seen.resize(CL * nfac);
  for (int i = 0; i < nfac; i++) std::memcpy(L.cov_seen.data() + CL * i, imus + (size_t)IMU_LEN * i + O_COV, CL * sizeof(double));
  std::memcpy(L.cov_inv_seen.data(), cov_invs, CL * nfac * sizeof(double));
  L.n_seen = nfac;
  return true;
}

static int li_damping_iter_queued(vxba_factor* f, double* states, double* imus, double imu_coef, int max_iter, double* hess_out, double* resis_out,
                                  double* trace_out, int* n_trace, bool with_g) {
  const int W = f->W;
  if (!f->opt[VXBA_OPT_LI_QUEUED_SWEEPS] || !f->opt[VXBA_OPT_LI_STRUCTURED_SOLVE] || has_collective(f) || W < 2 || max_iter < 1) return 1;
  const int n = vxi::DIM * W + (with_g ? 3 : 0), SL = vxi::STATE_LEN, m6 = 6 * W;
  const int g0 = with_g ? 6 : vxi::DIM;          // gauge rows at the head: frame 0's pose (gravity variant) or all of frame 0
  const int mr = n - g0;
  const auto t_call0 = std::chrono::steady_clock::now();
  // development aid (VXBA_LI_TIMING=1): where the host time of a call goes, phase by phase, summed over its iterations
  static const bool timing = [] { const char* e = getenv("VXBA_LI_TIMING"); return e && e[0] == '1'; }();
  enum { T_IMU, T_PREP, T_REC, T_LAUNCH, T_WAITH, T_HPLUS, T_WAITDX, T_STEP, T_IMUN, T_WAITR, T_DECIDE, T_SETUP, T_INV, T_N };
  double tph[T_N] = {0};
  auto tick = [&]() { return std::chrono::steady_clock::now(); };
  auto lap = [&](int k, std::chrono::steady_clock::time_point& t) { if (timing) { const auto n2 = tick(); tph[k] += std::chrono::duration<double, std::micro>(n2 - t).count(); t = n2; } };
  if (!f->h_feed) {
    VX_HIP(f, hipHostMalloc((void**)&f->h_feed, sizeof(double) * (12 * VXBA_MAX_WIN + 8), hipHostMallocMapped | hipHostMallocCoherent));   // fine-grained: the GPU must see the host's write WHILE the kernel runs
    VX_HIP(f, hipHostGetDevicePointer((void**)&f->zc_feed, f->h_feed, 0));
    f->h_feed[0] = 0.0;
  }
  // The reduced pose system solved INSIDE the residual-sweep launch (VXBA_OPT_LI_DEVICE_POSE_SOLVE, vxba_solve4.hpp): what eliminating the
  // velocity / bias unknowns adds to the pose block goes up through mapped host memory before the launch, dx and the trial poses come back
  // the same way.  Not for the gravity variant (three more dense unknowns) and not for the re-solve after a rejected step (the device's
  // copy of the LiDAR system has been overwritten by the speculative sweep by then): those keep the host solve and the pose feed below.
  const bool dev_solve = !with_g && f->opt[VXBA_OPT_LI_DEVICE_POSE_SOLVE] != 0 && W <= VXBA_MAX_WIN;
  if (dev_solve && !f->h_lirec) {
    VX_HIP(f, hipHostMalloc((void**)&f->h_lirec, sizeof(double) * vxk::li_rec_doubles(VXBA_MAX_WIN), hipHostMallocMapped | hipHostMallocCoherent));
    VX_HIP(f, hipHostGetDevicePointer((void**)&f->zc_lirec, f->h_lirec, 0));
    // Round 4: the record in fine-grained DEVICE memory, written by the host through the PCIe BAR (one sequential copy of the assembled
    // record, posted writes: ~1 us for 30 KB) instead of read by the solve out of host memory (a round trip per load on the step's
    // critical path: one workgroup needs ~20 us for 30 KB from host memory, ~5 from HBM -- scripts/ubench/host_write_vram.hip).  Falls
    // back to the mapped host buffer where the allocation is refused.  VXBA_LI_REC_VRAM=0: the round-3 path (A/B).
    const char* ev = getenv("VXBA_LI_REC_VRAM");
    if (!(ev && ev[0] == '0')) {
      // Only where the CPU can actually store into device memory: without a large BAR the allocation SUCCEEDS and the first host store
      // into it faults (round-4 advisor) -- ask the device first; any doubt keeps the mapped host buffer.
      int dev = 0, large_bar = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) { large_bar = 0; (void)hipGetLastError(); }
      void* p = nullptr;
      if (large_bar) {
        if (hipExtMallocWithFlags(&p, sizeof(double) * vxk::li_rec_doubles(VXBA_MAX_WIN), hipDeviceMallocFinegrained) == hipSuccess) f->lirec_vram = (double*)p;
        else (void)hipGetLastError();
      }
    }
    VX_HIP(f, hipHostMalloc((void**)&f->h_liout, sizeof(double) * vxk::li_out_doubles(VXBA_MAX_WIN), hipHostMallocMapped | hipHostMallocCoherent));
    VX_HIP(f, hipHostGetDevicePointer((void**)&f->zc_liout, f->h_liout, 0));
    std::memset(f->h_liout, 0, sizeof(double) * vxk::li_out_doubles(VXBA_MAX_WIN));
  }
  if (dev_solve && !f->h_packed2) {
    const size_t plen_max = (size_t)36 * VXBA_MAX_WIN * VXBA_MAX_WIN + 6 * VXBA_MAX_WIN + 2;
    VX_HIP(f, hipHostMalloc((void**)&f->h_packed2, plen_max * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    VX_HIP(f, hipHostGetDevicePointer((void**)&f->zc_packed2, f->h_packed2, 0));
  }
  if (!f->li_ev) VX_HIP(f, hipEventCreateWithFlags(&f->li_ev, hipEventDisableTiming));
  if (!f->li_ev2) VX_HIP(f, hipEventCreateWithFlags(&f->li_ev2, hipEventDisableTiming));
  if (!f->li_ev3) VX_HIP(f, hipEventCreateWithFlags(&f->li_ev3, hipEventDisableTiming));
  // two events for "a joint system's sweep + reduction is done": the one the host waits on now, and the one the speculative sweep queued
  // behind the residual sweep records (re-recording the first would make the host wait for a sweep that waits for the host)
  hipEvent_t ev_sys[2] = {f->li_ev2, f->li_ev3};
  int cur = 0;
  const size_t plen = vxba_packed_len(f);
  FactorView fv;
  int rc = residual_view(f, fv);
  if (rc) return rc;
  rc = ensure_partials3(f);
  if (rc) return rc;
  const int nv = vxk::k3_nv(W);
  const int nblocks3 = vxk::k3_blocks_for((f->V - 1) / nv + 1, vxk::k3_grid_blocks(f->cus));
  PoseArg pa0;
  std::memset(&pa0, 0, sizeof pa0);
  auto wait_event = [&](hipEvent_t ev) -> int {
    hipError_t q;
    q = vxwait::event_wait(ev);
    VX_HIP(f, q);
    return VXBA_OK;
  };
  // "The reduced system has arrived" is read off the mapped host buffer itself: the host fills it with NaNs at a moment when NO
  // reduction can be writing it, and the system is complete when the last NaN is gone (every slot is written exactly once per
  // reduction) -- a few microseconds before the end-of-launch release and the event behind it would be seen.  Safe moments: the start
  // of the call once the stream has drained (a previous call may have left a speculative reduction behind), and right after a system
  // has been consumed -- the next reduction sits behind a residual sweep that is still waiting for its poses.  After a REJECTED step
  // the wasted speculative reduction may still be running when the next poses are fed, so no fill then: `sentinel` is off and the next
  // system is awaited through its event alone.  (A first version filled after every residual sweep, "a Hessian sweep before the next
  // reduction": on small windows that sweep is 10 us, the fill sometimes came after the reduction, and the solve ran on NaNs.)
  // Device-solve mode: consecutive systems alternate between two mapped buffers (index = `cur`, like their events).  The launch whose
  // workgroup 0 solves does not wait for the host, so the speculative reduction behind it may complete at any time: it must neither land
  // in the buffer the host is still reading nor race the sentinel fill -- the sentinel of the NEXT buffer therefore goes in before the
  // launches that will write it (`next_sentinel`), never after.  The host-solve modes keep one buffer (both entries alias it).
  ensure_exchange(f);
  double* hpk[2] = {f->h_packed, dev_solve ? f->h_packed2 : f->h_packed};
  double* zpk[2] = {f->zc_packed, dev_solve ? f->zc_packed2 : f->zc_packed};
  bool sentinel = false, next_sentinel = false;
  auto nan_fill_buf = [&](double* b) { for (size_t k = 0; k < plen; k++) b[k] = std::numeric_limits<double>::quiet_NaN(); };
  auto nan_fill_packed = [&]() { nan_fill_buf(hpk[cur]); sentinel = true; };
  auto wait_packed = [&](hipEvent_t ev) -> int {
    const volatile double* hp = hpk[cur];
    for (;;) {
      if (sentinel) {
        bool all = true;
        for (size_t k = 0; k < plen; k++)
          if (!(hp[k] == hp[k])) { all = false; break; }
        if (all) return VXBA_OK;
      }
      const hipError_t q = hipEventQuery(ev);
      if (q == hipSuccess) return VXBA_OK;
      if (q != hipErrorNotReady) VX_HIP(f, q);
    }
  };
  if (f->li_reduction_in_flight) {     // the previous call ended (converged) with its last speculative reduction unconsumed
    hipError_t q;
    q = vxwait::stream_wait(f->stream);
    VX_HIP(f, q);
    f->li_reduction_in_flight = false;
  }
  // Device-solve mode: the first Hessian sweep goes out before anything else (it reads neither the control block nor the host buffer);
  // the sentinel fill and the control-block reset follow while it runs, ahead of the reduction that writes both.
  if (!dev_solve) {
    nan_fill_packed();
    // the device-side control block the queued sweeps read their poses from: not done, no error
    vxk::launch_lm_reset(f->d_lm, f->stream);
  }
  // first joint system: Hessian sweep at the caller's poses, as in the plain shell, marked by an event (the stream will not drain)
  {
    double Rp0[12 * VXBA_MAX_WIN];
    for (int i = 0; i < W; i++) std::memcpy(Rp0 + 12 * i, states + SL * i, sizeof(double) * 12);
    if (dev_solve) {      // the reduction also leaves the gauge-fixed LiDAR system in the device's LM state, where the in-launch solve reads it
      PoseArg pa;
      fill_poses(f, Rp0, pa);
      vxk::LMPending none;
      std::memset(&none, 0, sizeof none);
      if (vxk::launch_k3_hessian(fv, pa, nullptr, 0, none, nullptr, 0, f->V, f->d_partial3, nblocks3, f->precision, f->stream) < 0)
        return fail(f, VXBA_ERR_STATE, "li: cache planes are not consecutive");
      vxk::launch_lm_reset(f->d_lm, f->stream);
      nan_fill_packed();              // nothing that writes this buffer is in the stream yet (a previous call's reduction was drained above)
      vxk::launch_k3_finalize(f->d_partial3, nblocks3, W, f->d_lm, 0, 1, zpk[cur], f->stream, 1);
      VX_HIP(f, hipGetLastError());
    } else {
      rc = sweep_hess_device(f, Rp0, nullptr, nullptr, nullptr, 0, f->V, zpk[cur]);
      if (rc) return rc;
    }
    VX_HIP(f, hipEventRecord(ev_sys[cur], f->stream));
  }
  f->li.size(n, W - 1);
  std::vector<double>&Hess = f->li.Hess, &A = f->li.A, &JacT = f->li.JacT, &D = f->li.D, &rhs = f->li.rhs, &dxi = f->li.dxi, &work = f->li.work;
  std::vector<double>&HessN = f->li.HessN, &JacTN = f->li.JacTN, &cov_invs = f->li.cov_invs;
  std::vector<int>& perm = f->li.perm;
  std::vector<double> x_temp(states, states + (size_t)SL * W);
  auto tsetup = t_call0;
  lap(T_SETUP, tsetup);
  if (!li_information_matrices(f, W, imus, cov_invs.data())) return fail(f, VXBA_ERR_STATE, "li: singular IMU covariance (factor without samples?)");
  lap(T_INV, tsetup);
  vxh::BandSchurWork& bs = f->li_bs;
  const vxh::LiIndexSets sets = vxh::li_index_sets(W - 1, with_g ? 9 : 0, with_g ? 3 : 0);
  const int ny = (int)sets.Y.size(), nx = (int)sets.X.size();
  double* Aw = nullptr;                            // the block behind the gauge rows (set once Hess is known: the buffers swap)
  double u = 0.01, v = 2, residual1 = 0, residual2 = 0, imu_res = 0, imu_res_next = 0;
  bool is_calc_hess = true, imu_ready = false, sys_queued = true;   // sys_queued: a Hessian sweep for `states` is in the stream (li_ev2 marks its end)
  int nt = 0, nparts = 0;
  unsigned seq = 0;
  bool armed = false;                               // a residual sweep is queued and waits for its poses
  const double* last_hess = nullptr;
  vxi::ImuWork wk;
  bool ok = true;
  // queue iteration's residual sweep (waiting for the feed) and, with_spec, the Hessian sweep + reduction at the same trial poses
  auto queue_sweeps = [&](bool with_spec, bool dev) -> int {
    seq = ++f->lm_seq;
    if (seq == 0) seq = ++f->lm_seq;
    if (dev) {    // nobody will feed this launch: the sentinels of its outputs go in now (the previous launch's have been consumed)
      for (int k = 0; k < 18 * W + 1; k++) f->h_liout[k] = std::numeric_limits<double>::quiet_NaN();
      const int np = vxk::k2_nparts(f->V, f->opt[VXBA_OPT_K2_VOXELS_PER_BLOCK]);   // the launch below returns the same number by construction
      for (int k = 0; k < np; k++) f->h_partial2[k] = std::numeric_limits<double>::quiet_NaN();
      std::atomic_thread_fence(std::memory_order_release);
      // (measured and rejected: the record brought over by one DMA on a side stream + event instead of the solve reading it out of mapped host memory --
      // the three extra API calls cost 10 us per iteration, the solve was no faster: the PCIe reads are not what it waits for)
      nparts = vxk::launch_k2_residual(fv, pa0, f->d_lm, 0, seq, 0, f->V, f->zc_partial2, f->opt[VXBA_OPT_K2_VOXELS_PER_BLOCK], f->stream, nullptr, nullptr, nullptr,
                                       f->lirec_vram ? f->lirec_vram : f->zc_lirec, f->zc_liout);
    } else {
      nparts = vxk::launch_k2_residual(fv, pa0, f->d_lm, 0, seq, 0, f->V, f->zc_partial2, f->opt[VXBA_OPT_K2_VOXELS_PER_BLOCK], f->stream, nullptr, nullptr, f->zc_feed);
    }
    VX_HIP(f, hipGetLastError());
    VX_HIP(f, hipEventRecord(f->li_ev, f->stream));
    armed = !dev;
    if (with_spec) {
      vxk::LMPending pd;
      std::memset(&pd, 0, sizeof pd);
      pd.pending = 2;                               // linearise at the control block's trial poses; no decision inside the sweep
      if (vxk::launch_k3_hessian(fv, pa0, f->d_lm, 0, pd, nullptr, 0, f->V, f->d_partial3, nblocks3, f->precision, f->stream) < 0)
        return fail(f, VXBA_ERR_STATE, "li: cache planes are not consecutive");
      if (dev_solve) vxk::launch_k3_finalize(f->d_partial3, nblocks3, W, f->d_lm, 0, 1, zpk[cur ^ 1], f->stream, 1);
      else vxk::launch_k3_finalize(f->d_partial3, nblocks3, W, nullptr, 0, 0, zpk[cur ^ 1], f->stream);
      VX_HIP(f, hipGetLastError());
      VX_HIP(f, hipEventRecord(ev_sys[cur ^ 1], f->stream));
    }
    return VXBA_OK;
  };
  auto feed = [&](const double* st_trial) {
    for (int k = 0; k < nparts; k++) f->h_partial2[k] = std::numeric_limits<double>::quiet_NaN();   // a sweep that gave up leaves these
    for (int i = 0; i < W; i++) std::memcpy(f->h_feed + 1 + 12 * i, st_trial + SL * i, sizeof(double) * 12);
    std::atomic_thread_fence(std::memory_order_release);
    *(volatile double*)f->h_feed = (double)seq;
    armed = false;
  };
  // never leave a queued sweep waiting: whatever ends the call releases it (poses of the current states: harmless)
  struct Release { decltype(feed)& fd; bool& armed; const double* st; ~Release() { if (armed) fd(st); } } release_guard{feed, armed, states};

  for (int it = 0; it < max_iter; it++) {
    const bool recomputed = is_calc_hess;
    const bool with_spec = it + 1 < max_iter;
    f->li_reduction_in_flight = with_spec;          // cleared again when the next iteration consumes (or supersedes) it
    bool prepared = false;
    bool dev_this = false;                          // this iteration's pose system is solved inside the residual-sweep launch
    auto tp = tick();
    if (is_calc_hess) {
      if (imu_ready) { Hess.swap(HessN); JacT.swap(JacTN); imu_res = imu_res_next; imu_ready = false; }
      else {
        std::memset(Hess.data(), 0, sizeof(double) * n * n);
        std::memset(JacT.data(), 0, sizeof(double) * n);
        imu_res = vxi::li_add_imu_blocks(W, states, imus, imu_coef, true, Hess.data(), JacT.data(), wk, &ok, with_g, cov_invs.data());
        if (!ok) return fail(f, VXBA_ERR_STATE, "li: singular IMU covariance (factor without samples?)");
      }
      Aw = &Hess[(size_t)g0 * n + g0];
      lap(T_IMU, tp);
      if (dev_solve) {
        // band half first (velocities / biases: IMU terms only; the GPU is still sweeping), then what it adds to the pose block goes into the
        // record the in-launch solve reads, and only then the launches: the residual sweep's workgroup 0 solves, nobody waits for the host
        for (int y : sets.Y) { rhs[y] = -JacT[y + g0]; work[y] = u * Hess[(size_t)(y + g0) * n + y + g0]; }
        prepared = vxh::band_schur_prepare(Aw, n, work.data(), rhs.data(), sets.Y.data(), ny, sets.bw, sets.X.data(), nx, sets.xlo.data(), bs);
        dev_this = prepared && nx == m6 - 6;
        lap(T_PREP, tp);
        if (dev_this) {
          double* rec = f->h_lirec;
          rec[0] = u;
          for (int i = 0; i < W; i++) std::memcpy(rec + 1 + 12 * i, states + SL * i, sizeof(double) * 12);
          double* e = rec + 1 + 12 * W;
          double* E = rec + 1 + 18 * W;
          std::memset(e, 0, sizeof(double) * (m6 + (size_t)m6 * m6));
          const int nc = vxh::band_schur_stride(nx);
          for (int p = 0; p < nx; p++) {           // X[p] = pose unknown 6 + p of the device's ordering; Hess / JacT hold the IMU terms only at this point
            const double* Arow = Aw + (size_t)sets.X[p] * n;
            e[6 + p] = -JacT[sets.X[p] + g0] + bs.rx0[p];
            for (int q = 0; q <= p; q++) {
              const double v2 = Arow[sets.X[q]] + bs.S0[(size_t)p * nc + q] + (p == q ? u * Arow[sets.X[p]] : 0.0);
              E[(size_t)(6 + q) * m6 + 6 + p] = v2;
              E[(size_t)(6 + p) * m6 + 6 + q] = v2;
            }
          }
          if (f->lirec_vram) {
            // posted write-combining stores through the BAR; the store fence drains the write-combining buffers BEFORE the launch's doorbell is
            // rung (a C++ release fence emits no instruction on x86 and does not order WC stores: the record could otherwise still sit in a
            // buffer when the solve reads it -- round-4 advisor)
            std::memcpy(f->lirec_vram, rec, sizeof(double) * vxk::li_rec_doubles(W));
            vx_store_fence();
          }
          std::atomic_thread_fence(std::memory_order_release);
          if (with_spec) { nan_fill_buf(hpk[cur ^ 1]); next_sentinel = true; }   // nothing can be writing that buffer: its last system was consumed an iteration ago
        }
        lap(T_REC, tp);
        rc = queue_sweeps(with_spec, dev_this);
        if (rc) return rc;
        lap(T_LAUNCH, tp);
      } else {
        rc = queue_sweeps(with_spec, false);          // behind the system's sweep: starts when that is done, then waits for the poses
        if (rc) return rc;
        // band half of the solve (velocities / biases: IMU terms only) while the GPU is still sweeping
        for (int y : sets.Y) { rhs[y] = -JacT[y + g0]; work[y] = u * Hess[(size_t)(y + g0) * n + y + g0]; }
        prepared = vxh::band_schur_prepare(Aw, n, work.data(), rhs.data(), sets.Y.data(), ny, sets.bw, sets.X.data(), nx, sets.xlo.data(), bs);
      }
      if (!sys_queued) return fail(f, VXBA_ERR_STATE, "li: internal -- no Hessian sweep in flight for the accepted state");
      {
        const auto tw = std::chrono::steady_clock::now();
        rc = wait_packed(ev_sys[cur]);
        f->li_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw).count();
        if (rc) return rc;
      }
      sys_queued = false;
      lap(T_WAITH, tp);
      vxi::li_hess_plus(W, Hess.data(), JacT.data(), hpk[cur], hpk[cur] + (size_t)m6 * m6, n);
      residual1 = imu_res + hpk[cur][(size_t)m6 * m6 + m6];
      if (with_spec && !dev_this) {                // consumed; the speculative reduction sits behind a residual sweep that has no poses yet
        nan_fill_buf(hpk[cur ^ 1]);
        if (dev_solve) next_sentinel = true; else sentinel = true;
      }
      last_hess = Hess.data();
      if (it == 0 && resis_out) resis_out[0] = residual1;
    } else {
      rc = queue_sweeps(with_spec, false);          // rejected step: same system, new damping, new trial poses (host solve, fed below)
      if (rc) return rc;
      sentinel = false;                             // the rejected trial's reduction may still be writing the buffer
      next_sentinel = false;
    }
    // gauge rows: identity with a zero right-hand side (never written into the matrix: the solve works on the block behind them)
    for (int r = 0; r < g0; r++) { JacT[r] = 0.0; dxi[r] = 0.0; }
    for (int r = 0; r < n; r++) D[r] = r < g0 ? 1.0 : Hess[(size_t)r * n + r];
    for (int r = 0; r < mr; r++) { rhs[r] = -JacT[r + g0]; work[r] = u * D[r + g0]; }
    bool solved = prepared || vxh::band_schur_prepare(Aw, n, work.data(), rhs.data(), sets.Y.data(), ny, sets.bw, sets.X.data(), nx, sets.xlo.data(), bs);
    lap(T_HPLUS, tp);
    if (dev_this) {
      // the pose part of the step comes from the device (it has been solving while the host added the LiDAR blocks to its copy);
      // the launch never waits, so its end is the back-stop of this poll
      // Complete when every one of its 18W slots has replaced the NaN the host put there before the launch -- like the residual
      // partials and the reduced system, and for the same reason: writes of a kernel to host memory are not ordered among themselves,
      // so a "done" word written last proves nothing about the words before it (a first version polled the sequence number the
      // solve writes behind its results: one call in ~1000 read a stale entry).
      const volatile double* lo = f->h_liout;
      auto complete = [&]() { for (int k = 0; k < 18 * W; k++) if (!(lo[k] == lo[k])) return false; return true; };
      bool got = false;
      for (;;) {
        if (complete()) { got = true; break; }
        const hipError_t q = hipEventQuery(f->li_ev);
        if (q == hipSuccess) { got = complete(); break; }
        if (q != hipErrorNotReady) VX_HIP(f, q);
      }
      if (f->opt[VXBA_OPT_DEBUG_SOLVE_TIMEOUT] == 2 && it == 0) got = false;   // test hook: treat the first device step as not delivered
      if (!got) {
        // The launch is over and slots still hold their sentinels: the device's unpivoted elimination produced a non-finite step (an
        // ill-conditioned or indefinite reduced system) -- which cannot be told from "not delivered".  The host-solve modes take the
        // pivoted LDL^T for such a system; so does this one now: the launch's residual sweep (it ran at those poses) is discarded, the
        // sweeps are queued again waiting for the host's feed -- the path of a step after a rejection -- and the host solves below.
        f->li_dev_fallbacks++;
        dev_this = false;
        rc = queue_sweeps(with_spec, false);
        if (rc) return rc;
        sentinel = false;
        next_sentinel = false;
      }
    }
    if (dev_this) {
      const volatile double* lo = f->h_liout;
      std::atomic_thread_fence(std::memory_order_acquire);
      lap(T_WAITDX, tp);
      double xs[6 * VXBA_MAX_WIN];
      for (int p = 0; p < nx; p++) xs[p] = lo[6 + p];
      vxh::band_schur_finish_y(sets.Y.data(), ny, sets.bw, sets.X.data(), nx, xs, dxi.data() + g0, bs);
    } else if (solved) vxh::band_schur_finish(Aw, n, work.data(), rhs.data(), sets.Y.data(), ny, sets.bw, sets.X.data(), nx, dxi.data() + g0, bs);
    else {                                          // a band pivot was not positive: the reference's dense pivoted LDL^T on the whole system
      A.resize((size_t)n * n);
      std::memcpy(A.data(), Hess.data(), sizeof(double) * n * n);
      for (int c = 0; c < n; c++)
        for (int r = 0; r < g0; r++) { A[(size_t)c * n + r] = 0.0; A[(size_t)r * n + c] = 0.0; }
      for (int r = 0; r < g0; r++) A[(size_t)r * n + r] = 1.0;
      for (int r = 0; r < n; r++) { A[(size_t)r * n + r] += u * D[r]; rhs[r] = -JacT[r]; }
      vxh::ldlt_solve_inplace(n, A.data(), rhs.data(), dxi.data(), perm.data(), work.data());
    }
    // trial state and the factors' bias deltas (voxel_map.hpp:599-609 / 813-822)
    if (with_g) for (int k = 0; k < 3; k++) x_temp[21 + k] += dxi[n - 3 + k];     // x_stats_temp[0].g += dxi.tail(3): never reset upstream -- kept
    for (int j = 0; j < W; j++) {
      const double* d = &dxi[(size_t)vxi::DIM * j];
      const double* s = states + (size_t)SL * j;
      double* t = &x_temp[(size_t)SL * j];
      vxh::right_multiply_exp(s, d, t);
      for (int k = 0; k < 12; k++) t[9 + k] = s[9 + k] + d[3 + k];
      for (int k = 0; k < 3; k++) t[21 + k] = with_g ? x_temp[21 + k] : s[21 + k];
      if (dev_this) std::memcpy(t, f->h_liout + 6 * W + 12 * j, sizeof(double) * 12);   // the poses the residual sweep is evaluating, bit for bit
    }
    if (!dev_this) feed(x_temp.data());             // the queued residual sweep takes off
    for (int j = 0; j < W - 1; j++) vxi::imu_update_state(imus + (size_t)vxi::IMU_LEN * j, &dxi[(size_t)vxi::DIM * j]);
    double q1 = 0.0;
    for (int r = 0; r < n; r++) q1 += dxi[r] * (u * D[r] * dxi[r] - JacT[r]);
    q1 *= 0.5;
    lap(T_STEP, tp);
    // under the residual sweep: the IMU half of the NEXT system at the trial state (its residual falls out of the same evaluation)
    double r_imu;
    if (with_spec) {
      std::memset(HessN.data(), 0, sizeof(double) * n * n);
      std::memset(JacTN.data(), 0, sizeof(double) * n);
      r_imu = vxi::li_add_imu_blocks(W, x_temp.data(), imus, imu_coef, true, HessN.data(), JacTN.data(), wk, &ok, with_g, cov_invs.data());
    } else {
      r_imu = vxi::li_add_imu_blocks(W, x_temp.data(), imus, imu_coef, false, nullptr, nullptr, wk, &ok, false, cov_invs.data());
    }
    lap(T_IMUN, tp);
    // The residual is complete when every block partial has replaced the NaN the host put there (fine-grained host memory: a partial
    // arrives when its workgroup is done, a few microseconds before the kernel's end-of-launch release and the event behind it would
    // be seen); the event is only the back-stop for a sweep that gave up.
    for (;;) {
      bool all = true;
      const volatile double* hp = f->h_partial2;
      for (int k = 0; k < nparts; k++)
        if (!(hp[k] == hp[k])) { all = false; break; }
      if (all) break;
      const hipError_t q = hipEventQuery(f->li_ev);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) VX_HIP(f, q);
    }
    lap(T_WAITR, tp);
    const double r_lidar = host_sum_partials(f->h_partial2, nparts);
    if (!(r_lidar == r_lidar)) {
      int nan_cnt = 0, first_nan = -1;
      for (int k = 0; k < nparts; k++) if (!(f->h_partial2[k] == f->h_partial2[k])) { nan_cnt++; if (first_nan < 0) first_nan = k; }
      (void)hipStreamSynchronize(f->stream);
      int nan_after = 0;
      for (int k = 0; k < nparts; k++) if (!(f->h_partial2[k] == f->h_partial2[k])) nan_after++;
      vxk::LMState* hl = f->h_lm;
      (void)hipMemcpy(hl, f->d_lm, sizeof(vxk::LMState), hipMemcpyDeviceToHost);
      int nx_nan = 0, nd_nan = 0, nh_nan = 0, nj_nan = 0;
      for (double v2 : x_temp) if (!(v2 == v2)) nx_nan++;
      for (int r = 0; r < n; r++) { if (!(dxi[r] == dxi[r])) nd_nan++; if (!(JacT[r] == JacT[r])) nj_nan++; }
      for (size_t k = 0; k < (size_t)n * n; k++) if (!(Hess[k] == Hess[k])) nh_nan++;
      char msg[480];
      std::snprintf(msg, sizeof msg, "li: a queued residual sweep did not deliver its residual (iteration %d, seq %u: %d of %d block partials missing, first %d; %d after draining the stream; device error flag %d, published seq %u, feed word %.0f; NaNs in trial state %d, step %d, gradient %d, Hessian %d; recomputed %d, u %.3g, residual1 %.6g)",
                    it, seq, nan_cnt, nparts, first_nan, nan_after, hl->error, hl->solve_seq, f->h_feed[0], nx_nan, nd_nan, nj_nan, nh_nan, (int)recomputed, u, residual1);
      return fail(f, VXBA_ERR_STATE, msg);
    }
    residual2 = r_imu + r_lidar;
    const double q = residual1 - residual2;
    const double u_used = u, v_used = v;
    const bool accepted = vxh::lm_update_damping(residual1, residual2, q1, u, v);
    if (accepted) {
      std::memcpy(states, x_temp.data(), sizeof(double) * SL * W);
      is_calc_hess = true;
      if (with_spec) { imu_ready = true; imu_res_next = r_imu; sys_queued = true; cur ^= 1; if (dev_solve) { sentinel = next_sentinel; next_sentinel = false; } }
    } else {
      is_calc_hess = false;
      for (int j = 0; j < W - 1; j++) vxi::imu_rollback(imus + (size_t)vxi::IMU_LEN * j);
    }
    if (trace_out) {
      double* o = trace_out + (size_t)VXBA_TRACE_COLS * nt;
      o[0] = residual1; o[1] = residual2; o[2] = u_used; o[3] = v_used; o[4] = q; o[5] = q1; o[6] = accepted; o[7] = recomputed;
    }
    nt++;
    lap(T_DECIDE, tp);
    if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
  }
  if (resis_out) resis_out[1] = residual2;
  if (n_trace) *n_trace = nt;
  if (hess_out && last_hess) std::memcpy(hess_out, last_hess, sizeof(double) * n * n);
  f->li_last_call_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call0).count();
  if (timing)
    std::fprintf(stderr, "[vxba li queued] %d iterations, %.0f us: imu blocks %.0f | band half %.0f | record %.0f | launches %.0f | wait H %.0f | hess_plus+D %.0f | wait dx %.0f | "
                 "step/trial %.0f | imu blocks(next) %.0f | wait residual %.0f | decide %.0f | setup + first launches %.0f | covariance inverses %.0f\n", nt, f->li_last_call_us, tph[T_IMU], tph[T_PREP], tph[T_REC], tph[T_LAUNCH],
                 tph[T_WAITH], tph[T_HPLUS], tph[T_WAITDX], tph[T_STEP], tph[T_IMUN], tph[T_WAITR], tph[T_DECIDE], tph[T_SETUP], tph[T_INV]);
  return VXBA_OK;
}

int vxba_li_damping_iter(vxba_factor* f, double* states, double* imus, double imu_coef, int max_iter, double* hess_out, double* trace_out,
                         int* n_trace) {
  VX_LOCK(f);
  if (!f || !states || (!imus && f->W > 1) || max_iter < 0) return fail(f, VXBA_ERR_ARG, "li_damping_iter: bad argument");
  if (f->V == 0) return fail(f, VXBA_ERR_STATE, "li_damping_iter on an empty factor");
  VX_NARROW_ONLY(f, "li_damping_iter");
  hipSetDevice(f->device);
  const int W = f->W, n = vxi::DIM * W, SL = vxi::STATE_LEN;
  double u = 0.01, v = 2;
  const auto t_call0 = std::chrono::steady_clock::now();
  {
    const int rq = li_damping_iter_queued(f, states, imus, imu_coef, max_iter, hess_out, nullptr, trace_out, n_trace, false);
    if (rq != 1) return rq;
  }
  // the first Hessian sweep goes out before any host-side preparation (covariance inverses, buffers): it needs the poses only
  bool first_sweep_queued = false;
  if (max_iter > 0 && !has_collective(f)) {
    double Rp0[12 * VXBA_MAX_WIN];
    states_to_poses(W, states, Rp0);
    int rc = sweep_hess_device(f, Rp0, nullptr, nullptr, nullptr, 0, f->V, f->zc_packed);
    if (rc) return rc;
    first_sweep_queued = true;
  }
  f->li.size(n, W > 1 ? W - 1 : 0);
  std::vector<double>&Hess = f->li.Hess, &A = f->li.A, &JacT = f->li.JacT, &D = f->li.D, &rhs = f->li.rhs, &dxi = f->li.dxi, &work = f->li.work;
  std::vector<double>&HessN = f->li.HessN, &JacTN = f->li.JacTN;   // IMU half of the next joint system (speculative, see li_joint_residual)
  std::vector<double>& cov_invs = f->li.cov_invs;                   // cov is constant during the loop: invert once
  std::vector<int>& perm = f->li.perm;
  std::vector<double> x_temp(states, states + (size_t)SL * W);
  if (!li_information_matrices(f, W, imus, cov_invs.data())) return fail(f, VXBA_ERR_STATE, "li: singular IMU covariance (factor without samples?)");
  double residual1 = 0, residual2 = 0;
  bool is_calc_hess = true;
  int nt = 0;
  vxh::LiIndexSets li_sets;
  vxh::BandSchurWork& bs_work = f->li_bs;
  bool spec_queued = false;
  double imu_res_next = 0.0;
  const double* last_hess = nullptr;   // buffer that holds the last complete joint Hessian (*hess = Hess, :588): copied out once, at the end
  // development aid: VXBA_LI_TIMING=1 prints where the host time of one call goes
  static const bool timing = [] { const char* e = getenv("VXBA_LI_TIMING"); return e && e[0] == '1'; }();
  double t_sys = 0, t_solve = 0, t_res = 0;
  const double wait0 = f->li_wait_us;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (int it = 0; it < max_iter; it++) {
    const bool recomputed = is_calc_hess;
    const auto t0 = now();
    bool prepared = false;
    if (is_calc_hess) {
      // the band half of the structured solve needs nothing from the LiDAR factor: it runs while the GPU is still sweeping
      const std::function<void()> band_half = [&]() {
        const int g = vxi::DIM, m = n - g;
        if (m <= 0 || !f->opt[VXBA_OPT_LI_STRUCTURED_SOLVE]) return;
        if (li_sets.Y.empty()) li_sets = vxh::li_index_sets(W - 1, 0, 0);
        for (int y : li_sets.Y) { rhs[y] = -JacT[y + g]; work[y] = u * Hess[(size_t)(y + g) * n + y + g]; }
        prepared = vxh::band_schur_prepare(&Hess[(size_t)g * n + g], n, work.data(), rhs.data(), li_sets.Y.data(), (int)li_sets.Y.size(), li_sets.bw, li_sets.X.data(),
                                           (int)li_sets.X.size(), li_sets.xlo.data(), bs_work);
      };
      int rc = li_joint_system(f, states, imus, imu_coef, Hess.data(), JacT.data(), &residual1, false, cov_invs.data(), spec_queued || first_sweep_queued, &band_half,
                               spec_queued ? &imu_res_next : nullptr);
      spec_queued = false; first_sweep_queued = false;
      if (rc) return rc;
      last_hess = Hess.data();   // *hess = Hess, before the gauge fix (:588) -- this shell never modifies the matrix
    }
    const auto t1 = now();
    // gauge: frame 0's 15 rows / columns become identity rows with a zero right-hand side (:591-594): dxi = 0 there and they couple
    // to nothing, so the solve simply works on the trailing (n - 15) block of Hess (same solution, 27 % fewer flops at W = 10) --
    // in place: neither the gauge rows nor a damped copy of the matrix are written out unless the dense fallback needs one
    for (int r = 0; r < n; r++) D[r] = r < vxi::DIM ? 1.0 : Hess[(size_t)r * n + r];
    for (int r = 0; r < vxi::DIM; r++) JacT[r] = 0.0;
    {
      const int g = vxi::DIM, m = n - g;
      for (int r = 0; r < m; r++) { rhs[r] = -JacT[r + g]; work[r] = u * D[r + g]; }     // work: the damping u D on the diagonal
      for (int r = 0; r < g; r++) dxi[r] = 0.0;
      // band Cholesky of the velocity / bias part + Schur complement onto the poses (vxba_host.hpp); dense pivoted LDL^T if a band
      // pivot is not positive (or the option is off)
      bool solved = false;
      if (m > 0 && f->opt[VXBA_OPT_LI_STRUCTURED_SOLVE]) {
        if (li_sets.Y.empty()) li_sets = vxh::li_index_sets(W - 1, 0, 0);
        // after a rejected step (same system, new damping) both halves run here
        solved = prepared || vxh::band_schur_prepare(&Hess[(size_t)g * n + g], n, work.data(), rhs.data(), li_sets.Y.data(), (int)li_sets.Y.size(), li_sets.bw,
                                                     li_sets.X.data(), (int)li_sets.X.size(), li_sets.xlo.data(), bs_work);
        if (solved)
          vxh::band_schur_finish(&Hess[(size_t)g * n + g], n, work.data(), rhs.data(), li_sets.Y.data(), (int)li_sets.Y.size(), li_sets.bw, li_sets.X.data(),
                                 (int)li_sets.X.size(), dxi.data() + g, bs_work);
      }
      if (m > 0 && !solved) {
        A.resize((size_t)n * n);
        for (int c = 0; c < m; c++) std::memcpy(&A[(size_t)c * m], &Hess[(size_t)(c + g) * n + g], sizeof(double) * m);
        for (int r = 0; r < m; r++) A[(size_t)r * m + r] += u * D[r + g];
        vxh::ldlt_solve_inplace(m, A.data(), rhs.data(), dxi.data() + g, perm.data(), work.data());
      }
    }
    // trial state (:599-606) and the factors' bias deltas (:608-609)
    for (int j = 0; j < W; j++) {
      const double* d = &dxi[(size_t)vxi::DIM * j];
      const double* s = states + (size_t)SL * j;
      double* t = &x_temp[(size_t)SL * j];
      vxh::right_multiply_exp(s, d, t);
      for (int k = 0; k < 12; k++) t[9 + k] = s[9 + k] + d[3 + k];   // p, v, bg, ba
      for (int k = 0; k < 3; k++) t[21 + k] = s[21 + k];             // g is not optimised
    }
    for (int j = 0; j < W - 1; j++) vxi::imu_update_state(imus + (size_t)vxi::IMU_LEN * j, &dxi[(size_t)vxi::DIM * j]);
    double q1 = 0.0;
    for (int r = 0; r < n; r++) q1 += dxi[r] * (u * D[r] * dxi[r] - JacT[r]);
    q1 *= 0.5;
    const auto t2 = now();
    const bool speculate = it + 1 < max_iter && !has_collective(f);
    int rc = li_joint_residual(f, x_temp.data(), imus, imu_coef, &residual2, cov_invs.data(), speculate, HessN.data(), JacTN.data(), false, &imu_res_next);
    if (rc) return rc;
    spec_queued = speculate;     // only meaningful if the step is accepted (states <- x_temp); a rejected step never asks for the system
    const auto t3 = now();
    t_sys += us(t0, t1); t_solve += us(t1, t2); t_res += us(t2, t3);
    const double q = residual1 - residual2;
    const double u_used = u, v_used = v;
    const bool accepted = vxh::lm_update_damping(residual1, residual2, q1, u, v);
    if (accepted) {
      std::memcpy(states, x_temp.data(), sizeof(double) * SL * W);
      is_calc_hess = true;
      if (spec_queued) { Hess.swap(HessN); JacT.swap(JacTN); }   // the IMU half of the next system, built during the residual sweep
    } else {
      is_calc_hess = false;
      for (int j = 0; j < W - 1; j++) vxi::imu_rollback(imus + (size_t)vxi::IMU_LEN * j);
    }
    if (trace_out) {
      double* o = trace_out + (size_t)VXBA_TRACE_COLS * nt;
      o[0] = residual1; o[1] = residual2; o[2] = u_used; o[3] = v_used; o[4] = q; o[5] = q1; o[6] = accepted; o[7] = recomputed;
    }
    nt++;
    if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
  }
  if (n_trace) *n_trace = nt;
  if (hess_out && last_hess) std::memcpy(hess_out, last_hess, sizeof(double) * n * n);
  f->li_reduction_in_flight = f->li_reduction_in_flight || spec_queued;   // an unconsumed speculative sweep may still be writing the host buffer
  f->li_last_call_us = us(t_call0, now());
  if (timing) std::fprintf(stderr, "[vxba li] %d iterations, %.0f us in the call: joint system %.0f us (of which waiting for the sweep %.0f), solve+update %.0f us, joint residual %.0f us\n", nt, us(t_call0, now()), t_sys, f->li_wait_us - wait0, t_solve, t_res);
  return VXBA_OK;
}

// LI_BA_OptimizerGravity::damping_iter (voxel_map.hpp:775-862): three gravity unknowns at the tail, only frame 0's pose
// is gauge-fixed.  The trial state is never reset from the accepted one upstream (x_stats_temp, :813): the gravity of a
// rejected trial stays and the next increment lands on top of it -- kept.
int vxba_li_damping_iter_gravity(vxba_factor* f, double* states, double* imus, double imu_coef, int max_iter, double* hess_out,
                                 double* resis_out, double* trace_out, int* n_trace) {
  VX_LOCK(f);
  if (!f || !states || (!imus && f->W > 1) || max_iter < 0) return fail(f, VXBA_ERR_ARG, "li_damping_iter_gravity: bad argument");
  if (f->V == 0) return fail(f, VXBA_ERR_STATE, "li_damping_iter_gravity on an empty factor");
  VX_NARROW_ONLY(f, "li_damping_iter_gravity");
  hipSetDevice(f->device);
  {
    const int rq = li_damping_iter_queued(f, states, imus, imu_coef, max_iter, hess_out, resis_out, trace_out, n_trace, true);
    if (rq != 1) return rq;
  }
  const int W = f->W, n = vxi::DIM * W + 3, SL = vxi::STATE_LEN;
  double u = 0.01, v = 2;
  const auto t_call0 = std::chrono::steady_clock::now();
  bool first_sweep_queued = false;   // as in vxba_li_damping_iter: the first sweep runs under the host-side preparation
  if (max_iter > 0 && !has_collective(f)) {
    double Rp0[12 * VXBA_MAX_WIN];
    states_to_poses(W, states, Rp0);
    int rc = sweep_hess_device(f, Rp0, nullptr, nullptr, nullptr, 0, f->V, f->zc_packed);
    if (rc) return rc;
    first_sweep_queued = true;
  }
  f->li.size(n, W > 1 ? W - 1 : 0);
  std::vector<double>&Hess = f->li.Hess, &A = f->li.A, &JacT = f->li.JacT, &D = f->li.D, &rhs = f->li.rhs, &dxi = f->li.dxi, &work = f->li.work;
  std::vector<double>&HessN = f->li.HessN, &JacTN = f->li.JacTN;   // IMU half of the next joint system (speculative, see li_joint_residual)
  std::vector<double>& cov_invs = f->li.cov_invs;
  std::vector<int>& perm = f->li.perm;
  std::vector<double> x_temp(states, states + (size_t)SL * W);
  if (!li_information_matrices(f, W, imus, cov_invs.data())) return fail(f, VXBA_ERR_STATE, "li: singular IMU covariance (factor without samples?)");
  double residual1 = 0, residual2 = 0;
  bool is_calc_hess = true;
  int nt = 0;
  vxh::LiIndexSets li_sets;
  vxh::BandSchurWork& bs_work = f->li_bs;
  bool spec_queued = false;
  double imu_res_next = 0.0;
  const double* last_hess = nullptr;   // buffer that holds the last complete joint Hessian: copied out once, at the end
  for (int it = 0; it < max_iter; it++) {
    const bool recomputed = is_calc_hess;
    bool prepared = false;
    const int mr = n - 6;   // without the six gauge rows (identity, dxi = 0): [v, bg, ba of frame 0 | frames 1 .. W-1 | g]
    const bool structured = f->opt[VXBA_OPT_LI_STRUCTURED_SOLVE] && W > 1;
    if (structured && li_sets.Y.empty()) li_sets = vxh::li_index_sets(W - 1, 9, 3);
    if (is_calc_hess) {
      // the band half of the structured solve (velocities / biases: IMU terms only) runs while the GPU is still sweeping
      const std::function<void()> band_half = [&]() {
        if (!structured) return;
        for (int y : li_sets.Y) { rhs[y] = -JacT[y + 6]; work[y] = u * Hess[(size_t)(y + 6) * n + y + 6]; }
        prepared = vxh::band_schur_prepare(&Hess[(size_t)6 * n + 6], n, work.data(), rhs.data(), li_sets.Y.data(), (int)li_sets.Y.size(), li_sets.bw, li_sets.X.data(),
                                           (int)li_sets.X.size(), li_sets.xlo.data(), bs_work);
      };
      int rc = li_joint_system(f, states, imus, imu_coef, Hess.data(), JacT.data(), &residual1, true, cov_invs.data(), spec_queued || first_sweep_queued, &band_half,
                               spec_queued ? &imu_res_next : nullptr);
      spec_queued = false; first_sweep_queued = false;
      if (rc) return rc;
      last_hess = Hess.data();
    }
    if (it == 0 && resis_out) resis_out[0] = residual1;
    // gauge (:801-806): the pose of frame 0 -- identity rows with a zero right-hand side.  The matrix itself is left alone (the
    // structured solve works on the block behind those rows, the dense fallback applies them to its copy): *hess is copied out once
    for (int r = 0; r < 6; r++) JacT[r] = 0.0;
    for (int r = 0; r < n; r++) D[r] = r < 6 ? 1.0 : Hess[(size_t)r * n + r];
    bool solved = false;
    if (structured) {
      // in place on the trailing block of Hess (the gauge rows / columns lie outside it), damping handed over separately
      for (int r = 0; r < mr; r++) { work[r] = u * D[r + 6]; rhs[r] = -JacT[r + 6]; }
      for (int r = 0; r < 6; r++) dxi[r] = 0.0;
      solved = prepared || vxh::band_schur_prepare(&Hess[(size_t)6 * n + 6], n, work.data(), rhs.data(), li_sets.Y.data(), (int)li_sets.Y.size(), li_sets.bw,
                                                   li_sets.X.data(), (int)li_sets.X.size(), li_sets.xlo.data(), bs_work);
      if (solved)
        vxh::band_schur_finish(&Hess[(size_t)6 * n + 6], n, work.data(), rhs.data(), li_sets.Y.data(), (int)li_sets.Y.size(), li_sets.bw, li_sets.X.data(),
                               (int)li_sets.X.size(), dxi.data() + 6, bs_work);
    }
    if (!solved) {
      A = Hess;
      for (int c = 0; c < n; c++)
        for (int r = 0; r < 6; r++) { A[(size_t)c * n + r] = 0.0; A[(size_t)r * n + c] = 0.0; }
      for (int r = 0; r < 6; r++) A[(size_t)r * n + r] = 1.0;
      for (int r = 0; r < n; r++) { A[(size_t)r * n + r] += u * D[r]; rhs[r] = -JacT[r]; }
      vxh::ldlt_solve_inplace(n, A.data(), rhs.data(), dxi.data(), perm.data(), work.data());
    }
    for (int k = 0; k < 3; k++) x_temp[21 + k] += dxi[n - 3 + k];                 // x_stats_temp[0].g += dxi.tail(3)
    for (int j = 0; j < W; j++) {
      const double* d = &dxi[(size_t)vxi::DIM * j];
      const double* s = states + (size_t)SL * j;
      double* t = &x_temp[(size_t)SL * j];
      vxh::right_multiply_exp(s, d, t);
      for (int k = 0; k < 12; k++) t[9 + k] = s[9 + k] + d[3 + k];
      for (int k = 0; k < 3; k++) t[21 + k] = x_temp[21 + k];
    }
    for (int j = 0; j < W - 1; j++) vxi::imu_update_state(imus + (size_t)vxi::IMU_LEN * j, &dxi[(size_t)vxi::DIM * j]);
    double q1 = 0.0;
    for (int r = 0; r < n; r++) q1 += dxi[r] * (u * D[r] * dxi[r] - JacT[r]);
    q1 *= 0.5;
    const bool speculate = it + 1 < max_iter && !has_collective(f);
    int rc = li_joint_residual(f, x_temp.data(), imus, imu_coef, &residual2, cov_invs.data(), speculate, HessN.data(), JacTN.data(), true, &imu_res_next);
    spec_queued = speculate;
    if (rc) return rc;
    const double q = residual1 - residual2;
    const double u_used = u, v_used = v;
    const bool accepted = vxh::lm_update_damping(residual1, residual2, q1, u, v);
    if (accepted) {
      std::memcpy(states, x_temp.data(), sizeof(double) * SL * W);
      is_calc_hess = true;
      if (spec_queued) { Hess.swap(HessN); JacT.swap(JacTN); }   // the IMU half of the next system, built during the residual sweep
    } else {
      is_calc_hess = false;
      for (int j = 0; j < W - 1; j++) vxi::imu_rollback(imus + (size_t)vxi::IMU_LEN * j);
    }
    if (trace_out) {
      double* o = trace_out + (size_t)VXBA_TRACE_COLS * nt;
      o[0] = residual1; o[1] = residual2; o[2] = u_used; o[3] = v_used; o[4] = q; o[5] = q1; o[6] = accepted; o[7] = recomputed;
    }
    nt++;
    if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
  }
  if (resis_out) resis_out[1] = residual2;
  if (n_trace) *n_trace = nt;
  if (hess_out && last_hess) std::memcpy(hess_out, last_hess, sizeof(double) * n * n);
  f->li_reduction_in_flight = f->li_reduction_in_flight || spec_queued;
  f->li_last_call_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call0).count();
  return VXBA_OK;
}

}  // extern "C"
