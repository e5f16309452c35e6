// C-ABI implementation, LM drivers of the LiDAR-only optimiser (include/vxba.h): Lidar_BA_Optimizer::damping_iter as a device-resident
// loop (voxel_map.hpp:367-442), the host driver over caller-supplied sweeps, and the bench loop vxba_lm_steps.
#include <cstddef>
#include "vxba_capi_internal.hpp"

using namespace vxc;

extern "C" {

// Lidar_BA_Optimizer::damping_iter (voxel_map.hpp:367-442).  The whole loop is enqueued on the stream without a host
// round trip: the LM state (poses, damping, accept/reject flags) lives in device memory (vxk::LMState), the solve and
// the accept/reject step are single-workgroup kernels, and the sweeps gate themselves on the state's flags exactly
// where the reference branches (is_calc_hess, the early break).  One D2H copy + one sync at the end.
static int damping_iter_impl(vxba_factor* f, double* Rp, int max_iter, double* hess_out, double* resis_out, double* trace_out, int* n_trace,
                             int* is_converge) {
  VX_LOCK(f);
  if (!f || !Rp || max_iter < 0 || max_iter > vxk::LM_MAX_ITER) return fail(f, VXBA_ERR_ARG, "damping_iter: bad argument (max_iter <= 64)");
  if (f->V == 0 && !(is_wide(f) && has_collective(f))) return fail(f, VXBA_ERR_STATE, "damping_iter on an empty factor");   // an empty SHARD of a wide window sums zeros
  hipSetDevice(f->device);
  if (is_wide(f)) {
    // wide window (voxel_map.hpp:367-442 unchanged in structure): sweeps on the GPU; the damped (6W)-dimensional step by a dense
    // device Cholesky (only dxi, q1 and residual1 come back: ~5 KB) or, if that is unavailable / the system is not positive
    // definite, by the host's pivoted LDL^T on the downloaded system; accept/reject on the host.
    const int W = f->W, n = 6 * W;
    if (!f->wide_solver && !f->wide_solver_tried && f->opt[VXBA_OPT_WIDE_DEVICE_SOLVE] != 0) {
      // the library's own blocked Cholesky (vxba_wide.hip): a few device buffers, no third-party library to load
      f->wide_solver = vxw::wide_solver_create(n, f->stream);
      f->wide_solver_tried = true;
    }
    const bool use_device_solver = f->wide_solver && f->opt[VXBA_OPT_WIDE_DEVICE_SOLVE] != 0;
    double u = 0.01, v = 2;
    std::vector<double> x(Rp, Rp + 12 * W), x_temp(x), dxi(n), Hh, Jh;
    vxh::LMWorkspace ws;
    double residual1 = 0, residual2 = 0, q1 = 0;
    bool is_calc_hess = true, converge = true, host_copy_valid = false;
    int nt = 0;
    for (int i = 0; i < max_iter; i++) {
      const bool recomputed = is_calc_hess;
      if (is_calc_hess) {
        int rc = sweep_hess_device(f, x.data(), nullptr, nullptr, nullptr, 0, f->V, f->d_packed);
        if (rc) return rc;
        host_copy_valid = false;
      }
      bool on_device = false;
      if (use_device_solver) {
        double r1 = 0;
        on_device = vxw::wide_solver_step(f->wide_solver, f->d_packed, u, f->stream, dxi.data(), &q1, &r1, f->opt[VXBA_OPT_DEBUG_SOLVE_TIMEOUT] == 1) == 0;
        if (!on_device) f->fused_fallbacks++;     // counted like the narrow loop's fallback (VXBA_STAT_FUSED_FALLBACKS)
        if (on_device) {
          if (is_calc_hess) residual1 = r1;
          for (int j = 0; j < W; j++) {
            vxh::right_multiply_exp(&x[12 * j], &dxi[6 * j], &x_temp[12 * j]);
            for (int k = 0; k < 3; k++) x_temp[12 * j + 9 + k] = x[12 * j + 9 + k] + dxi[6 * j + 3 + k];
          }
        }
      }
      if (!on_device) {
        if (!host_copy_valid) {
          VX_HIP(f, hipMemcpyAsync(f->h_packed, f->d_packed, vxba_packed_len(f) * sizeof(double), hipMemcpyDeviceToHost, f->stream));
          VX_HIP(f, hipStreamSynchronize(f->stream));
          Hh.assign(f->h_packed, f->h_packed + (size_t)n * n);
          Jh.assign(f->h_packed + (size_t)n * n, f->h_packed + (size_t)n * n + n);
          if (is_calc_hess) residual1 = f->h_packed[(size_t)n * n + n];
          host_copy_valid = true;
        }
        q1 = vxh::lm_damped_step(W, Hh.data(), Jh.data(), u, x.data(), x_temp.data(), ws);
      }
      if (i == 0 && resis_out) resis_out[0] = residual1;
      int rc = sweep_residual_host(f, x_temp.data(), 0, f->V, &residual2);
      if (rc) return rc;
      const double q = residual1 - residual2;
      const double u_used = u, v_used = v;
      const bool accepted = vxh::lm_update_damping(residual1, residual2, q1, u, v);
      if (accepted) { x = x_temp; is_calc_hess = true; }
      else { is_calc_hess = false; converge = false; }
      if (trace_out) {
        double* o = trace_out + (size_t)VXBA_TRACE_COLS * nt;
        o[0] = residual1; o[1] = residual2; o[2] = u_used; o[3] = v_used; o[4] = q; o[5] = q1; o[6] = accepted; o[7] = recomputed;
      }
      nt++;
      if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
    }
    if (hess_out) {   // *hess = the last Hessian that was computed, before the gauge fix (voxel_map.hpp:391)
      VX_HIP(f, hipMemcpyAsync(f->h_packed, f->d_packed, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, f->stream));
      VX_HIP(f, hipStreamSynchronize(f->stream));
      std::memcpy(hess_out, f->h_packed, sizeof(double) * n * n);
    }
    if (resis_out) resis_out[1] = residual2;
    if (n_trace) *n_trace = nt;
    if (is_converge) *is_converge = converge ? 1 : 0;
    std::memcpy(Rp, x.data(), sizeof(double) * 12 * W);
    return VXBA_OK;
  }
  const int W = f->W, n = 6 * W;
  PoseArg x0;
  fill_poses(f, Rp, x0);
  vxk::launch_lm_init(f->d_lm, x0, W, 0, f->stream);
  // The accept/reject step of iteration i is taken in the prologue of iteration i+1's Hessian sweep (every workgroup
  // recomputes it from ctl[c]; workgroup 0 persists it into ctl[c^1]); a stand-alone decision kernel closes the loop.
  int c = 0;
  vxk::LMPending pend;
  std::memset(&pend, 0, sizeof pend);
  const bool spec = spec_collective(f);
  int spec_nparts = 0;
  const bool fuse_spec = spec && fused_sweeps_spec(f);
  bool spec_have_hess = false;   // the next iteration's Hessian sweep ran inside the previous fused launch (and its reduction + all-reduce behind it)
  for (int i = 0; spec && i < max_iter; i++) {
    int rc = VXBA_OK;
    if (!spec_have_hess) {
      rc = spec_hess_phase(f, Rp, &c, i == 0, i > 0, false, nullptr, spec_nparts);
      if (rc) return rc;
    }
    const unsigned seq = fused_solve(f) ? ++f->lm_seq : 0u;
    if (fuse_spec && seq && i + 1 < max_iter) {
      rc = spec_fused_phase(f, Rp, &c, seq, &spec_nparts);
      if (rc) return rc;
      spec_have_hess = true;
      continue;
    }
    spec_have_hess = false;
    if (!seq) vxk::launch_lm_solve(f->d_lm, c, W, f->stream);
    rc = sweep_residual_device(f, nullptr, f->d_lm, c, 0, f->V, nullptr, &spec_nparts, seq);
    if (rc) return rc;
  }
  if (spec && max_iter > 0) { int rc = spec_final_decision(f, Rp, &c, spec_nparts); if (rc) return rc; }
  // VXBA_OPT_FUSED_SWEEPS: wherever another iteration follows, [solve | residual sweep | the NEXT iteration's Hessian sweep] is one launch and
  // the reduction behind it takes this iteration's decision (have_hess: the system of the next iteration is already in the LM state)
  const bool fuse = !spec && fused_sweeps(f);
  bool have_hess = false;
  for (int i = 0; !spec && i < max_iter; i++) {
    int rc = VXBA_OK;
    if (!have_hess) {
      rc = sweep_hess_device(f, Rp, f->d_lm, &c, &pend, 0, f->V, f->d_packed, nullptr);
      if (rc) return rc;
      pend.pending = 0;
    }
    // damped solve + residual sweep at the trial state: one launch (the solve is workgroup 0 of the sweep) unless
    // VXBA_FUSED_SOLVE=0; without a collective the sweep's wave partials are summed by whoever takes the decision
    const unsigned seq = fused_solve(f) ? ++f->lm_seq : 0u;
    if (fuse && seq && i + 1 < max_iter) {
      rc = sweep_fused_device(f, f->d_lm, &c, seq);
      if (rc) return rc;
      have_hess = true;
      continue;
    }
    have_hess = false;
    if (!seq) vxk::launch_lm_solve(f->d_lm, c, W, f->stream);
    int nparts = 0;
    rc = sweep_residual_device(f, nullptr, f->d_lm, c, 0, f->V, has_collective(f) ? f->d_scalar : nullptr, &nparts, seq, false);
    if (rc) return rc;
    pend.pending = 1; pend.restart = 0;
    pend.d_scalar = has_collective(f) ? f->d_scalar : nullptr;
    pend.partial = f->d_partial2; pend.nparts = nparts;
  }
  if (pend.pending) { vxk::launch_lm_update(f->d_lm, c, pend, x0, W, f->stream); c ^= 1; }
  VX_HIP(f, hipGetLastError());
  VX_HIP(f, hipMemcpyAsync(f->h_lm, f->d_lm, sizeof(vxk::LMState), hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, stream_wait_spin(f->stream));
  if (f->h_lm->error) {
    f->solve_timed_out = true;
    return fail(f, VXBA_ERR_STATE, "damping_iter: a residual-sweep workgroup timed out waiting for the in-launch solve");
  }
  const vxk::LMCtl& st = f->h_lm->ctl[c];
  f->reject_heavy = 3 * st.n_reject > st.n_accept + st.n_reject;   // more than a third: see VXBA_OPT_FUSED_SWEEPS
  std::memcpy(Rp, st.x, sizeof(double) * 12 * W);
  if (hess_out) std::memcpy(hess_out, f->h_lm->hess_out, sizeof(double) * n * n);
  if (resis_out) { resis_out[0] = st.resis[0]; resis_out[1] = st.resis[1]; }
  const int nt = std::min(st.iter, vxk::LM_MAX_ITER);
  if (trace_out) std::memcpy(trace_out, f->h_lm->trace, sizeof(double) * VXBA_TRACE_COLS * nt);
  if (n_trace) *n_trace = nt;
  if (is_converge) *is_converge = st.converge;
  return VXBA_OK;
}

// Host-only LM shell over caller-supplied sweeps (same control flow as voxel_map.hpp:367-442).
int vxba_damping_iter_generic(int W, double* Rp, int max_iter, vxba_hess_fn hess_fn, vxba_resid_fn resid_fn, void* ctx, double* hess_out,
                              double* resis_out, double* trace_out, int* n_trace, int* is_converge) {
  if (W < 1 || !Rp || max_iter < 0 || !hess_fn || !resid_fn) return VXBA_ERR_ARG;
  const int n = 6 * W;
  double u = 0.01, v = 2;
  std::vector<double> packed((size_t)n * n + n + 1), Hess((size_t)n * n), JacT(n), x(Rp, Rp + 12 * W), x_temp(x);
  vxh::LMWorkspace ws;
  double residual1 = 0, residual2 = 0;
  bool is_calc_hess = true, converge = true;
  int nt = 0;
  for (int i = 0; i < max_iter; i++) {
    const bool recomputed = is_calc_hess;
    if (is_calc_hess) {
      if (hess_fn(ctx, x.data(), packed.data()) != 0) return VXBA_ERR_STATE;
      std::memcpy(Hess.data(), packed.data(), sizeof(double) * n * n);
      std::memcpy(JacT.data(), packed.data() + (size_t)n * n, sizeof(double) * n);
      residual1 = packed[(size_t)n * n + n];
      if (hess_out) std::memcpy(hess_out, Hess.data(), sizeof(double) * n * n);  // *hess = Hess, before the gauge fix
    }
    if (i == 0 && resis_out) resis_out[0] = residual1;
    const double q1 = vxh::lm_damped_step(W, Hess.data(), JacT.data(), u, x.data(), x_temp.data(), ws);
    if (resid_fn(ctx, x_temp.data(), &residual2) != 0) return VXBA_ERR_STATE;
    const double q = residual1 - residual2;
    const double u_used = u, v_used = v;
    const bool accepted = vxh::lm_update_damping(residual1, residual2, q1, u, v);
    if (accepted) { x = x_temp; is_calc_hess = true; }
    else { is_calc_hess = false; converge = false; }
    if (trace_out) {
      double* o = trace_out + (size_t)VXBA_TRACE_COLS * nt;
      o[0] = residual1; o[1] = residual2; o[2] = u_used; o[3] = v_used; o[4] = q; o[5] = q1; o[6] = accepted; o[7] = recomputed;
    }
    nt++;
    if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
  }
  if (resis_out) resis_out[1] = residual2;
  if (n_trace) *n_trace = nt;
  if (is_converge) *is_converge = converge ? 1 : 0;
  std::memcpy(Rp, x.data(), sizeof(double) * 12 * W);
  return VXBA_OK;
}

static int lm_steps_impl(vxba_factor* f, const double* Rp_init, int n_steps, int steps_per_solve, double* Rp_out, double* last_resis,
                  int64_t* stats_out) {
  VX_LOCK(f);
  if (!f || !Rp_init || n_steps < 0 || steps_per_solve < 1) return fail(f, VXBA_ERR_ARG, "lm_steps: bad argument");
  if (f->V == 0) return fail(f, VXBA_ERR_STATE, "lm_steps on an empty factor");
  VX_NARROW_ONLY(f, "lm_steps");
  hipSetDevice(f->device);
  const int W = f->W;
  PoseArg x0;
  fill_poses(f, Rp_init, x0);
  if (!f->snapshot || f->snapshot_v != f->V || f->snapshot_vs != f->VS) return fail(f, VXBA_ERR_STATE, "lm_steps needs vxba_snapshot_cache first");
  vxk::launch_lm_init(f->d_lm, x0, W, 1, f->stream);
  int c = 0;
  vxk::LMPending pend;
  std::memset(&pend, 0, sizeof pend);
  const bool spec = spec_collective(f);
  int spec_nparts = 0;
  bool prev_last = false;
  const bool fuse_spec = spec && fused_sweeps_spec(f);
  bool spec_have_hess = false;
  for (int s = 0; spec && s < n_steps; s++) {
    const bool first = (s % steps_per_solve) == 0;
    const bool last = ((s + 1) % steps_per_solve) == 0 && s + 1 < n_steps;
    int rc = VXBA_OK;
    if (!spec_have_hess) {
      rc = spec_hess_phase(f, Rp_init, &c, first, s > 0, prev_last, first ? f->snapshot : nullptr, spec_nparts);
      if (rc) return rc;
    }
    const unsigned seq = fused_solve(f) ? ++f->lm_seq : 0u;
    // inside a solve: this step's residual sweep and the next step's Hessian sweep in one launch, the reduction + all-reduce + decision behind it
    if (fuse_spec && seq && ((s + 1) % steps_per_solve) != 0 && s + 1 < n_steps) {
      rc = spec_fused_phase(f, Rp_init, &c, seq, &spec_nparts);
      if (rc) return rc;
      spec_have_hess = true;
      prev_last = last;
      continue;
    }
    spec_have_hess = false;
    if (!seq) vxk::launch_lm_solve(f->d_lm, c, W, f->stream);
    rc = sweep_residual_device(f, nullptr, f->d_lm, c, 0, f->V, nullptr, &spec_nparts, seq);
    if (rc) return rc;
    prev_last = last;
  }
  if (spec && n_steps > 0) { int rc = spec_final_decision(f, Rp_init, &c, spec_nparts); if (rc) return rc; }
  const bool fuse = !spec && fused_sweeps(f);
  bool have_hess = false;
  for (int s = 0; !spec && s < n_steps; s++) {
    // a new window every steps_per_solve steps: its first Hessian sweep reads the SNAPSHOT cache directly (the re-seeded
    // cache of a new window -- no copy) and its prologue resets poses and damping (pend.restart of the previous step);
    // the residual sweeps keep writing the live cache
    const bool first = (s % steps_per_solve) == 0;
    const bool last = ((s + 1) % steps_per_solve) == 0 && s + 1 < n_steps;
    int rc = VXBA_OK;
    if (!have_hess) {
      rc = sweep_hess_device(f, Rp_init, f->d_lm, &c, &pend, 0, f->V, f->d_packed, first ? f->snapshot : nullptr);
      if (rc) return rc;
      pend.pending = 0;
    }
    const unsigned seq = fused_solve(f) ? ++f->lm_seq : 0u;
    // inside a solve: this step's residual sweep and the next step's Hessian sweep in one launch (as vxba_damping_iter does)
    if (fuse && seq && ((s + 1) % steps_per_solve) != 0 && s + 1 < n_steps) {
      rc = sweep_fused_device(f, f->d_lm, &c, seq);
      if (rc) return rc;
      have_hess = true;
      continue;
    }
    have_hess = false;
    if (!seq) vxk::launch_lm_solve(f->d_lm, c, W, f->stream);
    int nparts = 0;
    rc = sweep_residual_device(f, nullptr, f->d_lm, c, 0, f->V, has_collective(f) ? f->d_scalar : nullptr, &nparts, seq, false);
    if (rc) return rc;
    pend.pending = 1; pend.restart = last ? 1 : 0;
    pend.d_scalar = has_collective(f) ? f->d_scalar : nullptr;
    pend.partial = f->d_partial2; pend.nparts = nparts;
  }
  if (pend.pending) { vxk::launch_lm_update(f->d_lm, c, pend, x0, W, f->stream); c ^= 1; }
  VX_HIP(f, hipGetLastError());
  static_assert(offsetof(vxk::LMState, trace) == vxk::LM_HEAD_BYTES, "LMState: the head vxba_lm_steps reads back");
  VX_HIP(f, hipMemcpyAsync(f->h_lm, f->d_lm, vxk::LM_HEAD_BYTES, hipMemcpyDeviceToHost, f->stream));   // poses, residuals, counters, error flag: not the trace / Hessians (66 KB)
  VX_HIP(f, stream_wait_spin(f->stream));   // by polling, as in damping_iter: waking up from hipStreamSynchronize costs 15-25 us -- 1.5 % of a 20-step call
  if (f->h_lm->error) return fail(f, VXBA_ERR_STATE, "lm_steps: a residual-sweep workgroup timed out waiting for the in-launch solve");
  const vxk::LMCtl& st = f->h_lm->ctl[c];
  f->reject_heavy = 3 * st.n_reject > st.n_accept + st.n_reject;   // more than a third: see VXBA_OPT_FUSED_SWEEPS
  if (Rp_out) std::memcpy(Rp_out, st.x, sizeof(double) * 12 * W);
  if (last_resis) { last_resis[0] = st.residual1; last_resis[1] = st.residual2; }
  if (stats_out) { stats_out[0] = st.iter; stats_out[1] = st.n_accept; stats_out[2] = st.n_reject; }
  return VXBA_OK;
}

}  // extern "C"

// ---- entry points that may have summed through the peers' mailboxes: a peer that never arrived must not pass silently ----
static int peer_check(vxba_factor* f, int rc) {
  if (rc != VXBA_OK || !f || !has_peer(f)) return rc;
  int st = 0;
  const int r2 = vxba_peer_status(f, &st);
  if (r2 != VXBA_OK) return r2;
  return st ? fail(f, VXBA_ERR_STATE, "peer all-reduce: a peer did not arrive within the wait bound (results are not a sum)") : VXBA_OK;
}
int vxba_damping_iter(vxba_factor* f, double* Rp, int max_iter, double* hess_out, double* resis_out, double* trace_out, int* n_trace, int* is_converge) {
  int rc = damping_iter_impl(f, Rp, max_iter, hess_out, resis_out, trace_out, n_trace, is_converge);
  if (rc == VXBA_ERR_STATE && f && f->solve_timed_out && has_collective(f)) {
    // Sharded: a timeout is a per-GPU event, and this rank has already issued the call's all-reduces -- a rank-local retry would issue
    // more of them which no other rank matches (RCCL hangs, the mailbox sequence numbers drift apart).  The error goes to the caller,
    // who switches VXBA_OPT_FUSED_SOLVE off on ALL ranks and calls again.
    f->solve_timed_out = false;
    return fail(f, VXBA_ERR_STATE, "in-launch solve timed out on a sharded factor: set VXBA_OPT_FUSED_SOLVE = 0 on every rank and retry");
  }
  if (rc == VXBA_ERR_STATE && f && f->solve_timed_out) {
    // The in-launch solve relies on workgroup 0 of the residual sweep making progress while the others poll (bounded): true for
    // in-order dispatch on an otherwise idle device, not guaranteed under CU masking / a serialising profiler / a co-resident
    // kernel.  A timeout is therefore not an error of the caller's: run the same call again with the solve as its own launch.
    // Rp is untouched on the failure path; the (lambda, U, merged) cache the first Hessian sweep needs is the one of the entry
    // poses, which the failed attempt has overwritten -- rebuild it first.
    f->solve_timed_out = false;
    const int saved = f->opt[VXBA_OPT_FUSED_SOLVE];
    f->opt[VXBA_OPT_FUSED_SOLVE] = 0;
    double r = 0;
    rc = vxba_evaluate_only_residual(f, Rp, 0, f->V, &r);
    if (rc == VXBA_OK) rc = damping_iter_impl(f, Rp, max_iter, hess_out, resis_out, trace_out, n_trace, is_converge);
    f->opt[VXBA_OPT_FUSED_SOLVE] = saved;
    f->fused_fallbacks++;
  }
  return peer_check(f, rc);
}
int vxba_lm_steps(vxba_factor* f, const double* Rp_init, int n_steps, int steps_per_solve, double* Rp_out, double* last_resis, int64_t* stats_out) {
  return peer_check(f, lm_steps_impl(f, Rp_init, n_steps, steps_per_solve, Rp_out, last_resis, stats_out));
}

