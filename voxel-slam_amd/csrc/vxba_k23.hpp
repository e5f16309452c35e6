// K2 + K3 in ONE launch, behind the in-launch solve: the residual sweep at the trial poses (LidarFactor::evaluate_only_residual,
// voxel_map.hpp:243-279) and the Hessian / gradient sweep that linearises at the SAME poses one LM iteration later
// (LidarFactor::acc_evaluate2, voxel_map.hpp:132-241; Lidar_BA_Optimizer::damping_iter, :386-439).  Included by vxba_kernels.hip
// inside namespace vxk, after vxba_k3.hpp and k2_residual_kernel.
//
// Why (round 6).  One LM iteration was three launches -- Hessian sweep | reduction | [solve + residual sweep] -- and the Hessian sweep
// of iteration i + 1 linearises exactly where the residual sweep of iteration i evaluated (the trial poses, speculated as accepted since
// round 1).  At the metric's size a third of the Hessian sweep is fixed cost -- a cold launch, 12 MB of first batches asked for by the
// whole chip at once with nothing to overlap them with -- and a launch boundary sits between two kernels that walk the same voxels.
// Here every sweep workgroup
//   (1) requests the cluster rows of ITS voxels (one lane per voxel: k2_residual_kernel's mapping) while workgroup 0 runs the damped solve,
//   (2) waits for the trial poses (one poller wave per workgroup, as before),
//   (3) runs the residual half on those voxels -- transform, covariance, warm-started eigen-decomposition: k2_residual_kernel's arithmetic
//       (round-off apart: the lane-pair sum, and the covariance / scales through one reciprocal, k23_cov) -- writes the (lambda, U, merged, aux) cache planes with plain stores (they stay in the XCD's L2) and adds
//       sum coe lambda_0 into one partial per workgroup,
//   (4) passes ONE workgroup barrier (stores acknowledged first) and runs the Hessian half over the same voxels: k3_sweep_body<FUSED>,
//       whose first batches come out of the caches and whose plane parameters are the lines this workgroup has just written.
// The voxel range of a workgroup is the one the Hessian sweep gives it (a contiguous run of NV-voxel batches), so the residual half needs
// no hand-over between workgroups.  The accept / reject decision of the step is taken by the SOLVE workgroup, which has nothing else to do
// once the trial poses are out: it waits for the sweep workgroups' residual sums (8 bytes each, agent-scope stores into slots that held
// NaN), adds them up in lm_residual2's order, runs lm_decide and persists the decided control block into ctl[c ^ 1] -- all of it while
// the Hessian half runs.  The reduction kernel behind the launch (k3_finalize_kernel on ctl[c ^ 1]) is gated by that block as ever:
// calc_hess = "accepted" -> the reduced system is adopted, rejected -> it is dropped, like the sharded speculative loop's
// lm_spec_unpack_kernel.  A rejected step therefore costs a whole Hessian half (the reference recomputes nothing then); the first
// iteration of a solve (cache left by push_voxel) and the last one (no Hessian behind it) keep the stand-alone kernels.
//
// Two things the hand-over between the halves no longer waits for (round 6, second pass over the launch's timeline -- the residual half took
// 19.5k cycles from the poses to its barrier against ~10k in the stand-alone kernel, and the Hessian half another 4.5k to its first barrier):
//   * PAIR mode (a workgroup owns <= 256 voxels, i.e. up to ~65k voxels on 255 CUs: the metric's size): a voxel's frames are split over a
//     LANE PAIR (lane 2j: fix cluster + frames [0, H1), lane 2j + 1: frames [H1, W), H1 = ceil(W / 2)).  Half the rows per lane means ALL of
//     them -- and the warm start, coe and the Hessian half's first batch -- are in registers before the trial poses exist (140 + 20
//     registers; the one-lane-per-voxel form could hold four frames of ten beside a second wave per SIMD and fetched the rest while it
//     worked), the transform chain is five frames long instead of ten, and all eight waves of the workgroup work (196 voxels are 392 lanes).
//     The halves' sums meet through one DPP quad permute per value (a + b on one lane, b + a on the other: the same bits), both lanes run
//     the eigen-decomposition (no extra issue slots: the partner lane was idle), each stores 13 of the 26 cache planes.
//   * The plane parameters of the workgroup's FIRST EIGHT batches go from the residual half's registers straight into the Hessian half's
//     staging corners in LDS (the 18-double record k3_unstage_params reads): the first phase A starts behind one LDS barrier -- no wait
//     for the cache stores' acknowledgement, no round trip through the L2.  The stores are waited for in front of the first step's
//     barrier instead (k3_sweep_body), a phase A later, before any wave asks the L2 for the next step's parameters.
#pragma once

// Planes of a factor relative to fv.cl (ONE allocation: vxc::view): cl [10 W] | fix 10 | coe 1 | eigval 3 | eigvec 9 | merged 10 | aux 4.
template <int W>
struct K23Planes {
  static constexpr int FIX = 10 * W, COE = FIX + 10, EIGVAL = COE + 1, EIGVEC = EIGVAL + 3, MERGED = EIGVEC + 9, AUX = MERGED + 10, COUNT = AUX + 4;
};

// What a lane holds while it waits for the trial poses: the fix cluster and the first HF frames' clusters of its voxel.  All 10 + 10 W rows
// at once (k2_residual_kernel: 294 registers, one wave per SIMD) do not fit beside a second wave per SIMD; the other W - HF frames are
// requested as the transform works through the first ones -- frame i + HF into the registers frame i has just left -- and come out of the
// L2 / the Infinity Cache (a window's 40 MB of cluster rows are re-read every iteration).
template <int W>
struct K23Regs {
  static constexpr int HF = (W + 1) / 2 < 4 ? (W + 1) / 2 : 4;   // four frames ahead at W >= 7 (five: two registers short of 256 at W = 9, 10)
  double fx[10], ring[HF][10];
};

template <int W>
__device__ __forceinline__ void k23_load_frame(__amdgpu_buffer_rsrc_t rs, unsigned vs8, unsigned a8, int i, double c[10]) {
#pragma unroll
  for (int k = 0; k < 10; k++) c[k] = k3_ld64<0>(rs, a8, (unsigned)(10 * i + k) * vs8);
}
// the requests that do not depend on the poses: address = descriptor base (SGPRs) + plane * VS * 8 (SGPR) + voxel * 8 (VGPR)
template <int W>
__device__ __forceinline__ void k23_issue(__amdgpu_buffer_rsrc_t rs, unsigned vs8, unsigned a8, K23Regs<W>& r) {
  using P = K23Planes<W>;
#pragma unroll
  for (int k = 0; k < 10; k++) r.fx[k] = k3_ld64<0>(rs, a8, (unsigned)(P::FIX + k) * vs8);
#pragma unroll
  for (int i = 0; i < K23Regs<W>::HF; i++) k23_load_frame<W>(rs, vs8, a8, i, r.ring[i]);
}

__device__ __forceinline__ void k23_st64(__amdgpu_buffer_rsrc_t rs, unsigned a8, unsigned soff, double v) {
#ifndef VXBA_K23_STORE_AUX
#define VXBA_K23_STORE_AUX 16
#endif
  // aux 16 = sc1, written through like k2_residual_kernel's cache stores: no dirty L2 lines for the kernel boundary behind the launch to write back
  // (8.8 MB at the metric's size); the Hessian half reads them past the L1 (sc1 loads).  0 = plain stores (the lines stay in the XCD's L2): measured below.
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), rs, (int)a8, (int)soff, VXBA_K23_STORE_AUX);
}

// The residual half of one voxel, one lane per voxel: k2_residual_kernel's arithmetic, statement for statement (same frame order, same
// unfused / fused operations inside vxm::*).  Returns coe * lambda_0.
// Instrumented build (DBG), stamps of wave gw: 15 transform done, 18 eigen-decomposition done, 21 cache stores issued.
// hook(): called between the eigen-decomposition and the cache stores (the Hessian half's first cluster rows are requested there).
struct K23NoHook { __device__ __forceinline__ void operator()() const {} };
// Covariance of the merged cluster and the derived scales with ONE reciprocal and two reciprocal square roots (estimate + Newton steps,
// vxm::fast_rcp / fast_rsqrt: full fp64 accuracy, 1 ulp from the IEEE quotients) where vxm::cluster_cov / gap_scales spend nine IEEE divisions, two
// more and two square roots -- ~140 of the residual half's ~1 700 fp64 instructions per wave, and the half is bound by their issue.
// The lane-pair form only (the one-lane form has no registers to spare for it: 660 spilled registers when tried); the stand-alone sweeps
// keep the reference's expression (voxel_map.hpp:264-267); the two forms of the loop agree to round-off, as they already do through the
// lane-pair sum.
__device__ __forceinline__ void k23_cov(const double P[6], const double v[3], double N, double C[6], double& invN) {
  invN = vxm::fast_rcp(N);
  const double vb[3] = {v[0] * invN, v[1] * invN, v[2] * invN};
  C[0] = P[0] * invN - vb[0] * vb[0];
  C[1] = P[1] * invN - vb[0] * vb[1];
  C[2] = P[2] * invN - vb[0] * vb[2];
  C[3] = P[3] * invN - vb[1] * vb[1];
  C[4] = P[4] * invN - vb[1] * vb[2];
  C[5] = P[5] * invN - vb[2] * vb[2];
}
__device__ __forceinline__ void k23_gap_scales(const double lam[3], double& s1, double& s2) {   // sqrt(2 / gap) = 1 / sqrt(gap / 2)
  s1 = vxm::fast_rsqrt(0.5 * (lam[1] - lam[0]));
  s2 = vxm::fast_rsqrt(0.5 * (lam[2] - lam[0]));
}
// The record of a voxel as k3_unstage_params reads it (k3_param_plane): u planes 0..8 | s1 s2 | merged first moment | 1/N | sqrt(coe) | lambda_0 | coe
template <bool FAST>
__device__ __forceinline__ void k23_record(double v[18], const double lam[3], const double U[9], const double Sv[3], double invN, double coe) {
#pragma unroll
  for (int col = 0; col < 3; col++)
#pragma unroll
    for (int row = 0; row < 3; row++) v[3 * col + row] = U[3 * row + col];
  if (FAST) k23_gap_scales(lam, v[9], v[10]);
  else vxm::gap_scales(lam, v[9], v[10]);
  v[11] = Sv[0]; v[12] = Sv[1]; v[13] = Sv[2];
  v[14] = invN; v[15] = sqrt(coe); v[16] = lam[0]; v[17] = coe;
}
// rec (LDS, nullable per lane): where the Hessian half's first phase A expects this voxel's record
template <int W, bool DBG = false, class Hook = K23NoHook>
__device__ __forceinline__ double k23_finish(__amdgpu_buffer_rsrc_t rs, unsigned vs8, unsigned a8, bool valid, const double* pose_lds, K23Regs<W>& r, double* rec = nullptr, int gw = 0, Hook hook = Hook()) {
  using P = K23Planes<W>;
  constexpr int HF = K23Regs<W>::HF;
  double SP[6], Sv[3], SN, C[6], lam[3] = {0.0, 0.0, 0.0}, U[9], Up[9], coe = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) SP[k] = r.fx[k];
#pragma unroll
  for (int k = 0; k < 3; k++) Sv[k] = r.fx[6 + k];
  SN = r.fx[9];
#pragma unroll
  for (int i = 0; i < W; i++) {
    double ci[10];
#pragma unroll
    for (int k = 0; k < 10; k++) ci[k] = r.ring[i % HF][k];
    if (i + HF < W) k23_load_frame<W>(rs, vs8, a8, i + HF, r.ring[i % HF]);
    if (i == (W - HF > 0 ? W - HF : 0)) {
      // the warm start of the eigensolver (previous eigenvectors: plane 3 col + row -> row-major) and coe, requested when the ring starts to drain
      coe = k3_ld64<0>(rs, a8, (unsigned)P::COE * vs8);
#pragma unroll
      for (int col = 0; col < 3; col++)
#pragma unroll
        for (int row = 0; row < 3; row++) Up[3 * row + col] = k3_ld64<0>(rs, a8, (unsigned)(P::EIGVEC + 3 * col + row) * vs8);
    }
    __builtin_amdgcn_sched_barrier(0);   // the requests stay in front of the frame's arithmetic
    double R[9], p[3];
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) R[3 * rr + cc] = pose_lds[12 * i + 3 * cc + rr];
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = pose_lds[12 * i + 9 + k];
    // N == 0 <=> frame i did not observe this voxel (voxel_map.hpp:258): contributes nothing
    const bool obs = ci[9] != 0.0;
#pragma unroll
    for (int k = 0; k < 10; k++) ci[k] = obs ? ci[k] : 0.0;
    vxm::transform_accumulate(ci, ci + 6, ci[9], R, p, SP, Sv, SN);
  }
  vxm::cluster_cov(SP, Sv, SN, C);
  if (DBG) { asm volatile("" :: "v"(C[0]), "v"(C[3]), "v"(C[5])); dbg_stamp(true, gw, 15); }
  vxm::eig_sym3_warm(C, Up, lam, U);
  if (DBG) { asm volatile("" :: "v"(lam[0]), "v"(U[0]), "v"(U[8])); dbg_stamp(true, gw, 18); }
  hook();
  if (rec) {
    double v[18];
    k23_record<false>(v, lam, U, Sv, 1.0 / SN, coe);
#pragma unroll
    for (int k = 0; k < 9; k++) *reinterpret_cast<v2d*>(rec + 2 * k) = (v2d){v[2 * k], v[2 * k + 1]};
  }
  if (valid) {
#pragma unroll
    for (int k = 0; k < 3; k++) k23_st64(rs, a8, (unsigned)(P::EIGVAL + k) * vs8, lam[k]);
#pragma unroll
    for (int col = 0; col < 3; col++)
#pragma unroll
      for (int row = 0; row < 3; row++) k23_st64(rs, a8, (unsigned)(P::EIGVEC + 3 * col + row) * vs8, U[3 * row + col]);
#pragma unroll
    for (int k = 0; k < 6; k++) k23_st64(rs, a8, (unsigned)(P::MERGED + k) * vs8, SP[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) k23_st64(rs, a8, (unsigned)(P::MERGED + 6 + k) * vs8, Sv[k]);
    k23_st64(rs, a8, (unsigned)(P::MERGED + 9) * vs8, SN);
    double s1, s2;
    vxm::gap_scales(lam, s1, s2);
    k23_st64(rs, a8, (unsigned)(P::AUX + 0) * vs8, s1);
    k23_st64(rs, a8, (unsigned)(P::AUX + 1) * vs8, s2);
    k23_st64(rs, a8, (unsigned)(P::AUX + 2) * vs8, 1.0 / SN);
    k23_st64(rs, a8, (unsigned)(P::AUX + 3) * vs8, sqrt(coe));
  }
  dbg_stamp(DBG, gw, 21);
  return valid ? coe * lam[0] : 0.0;
}

// ---- PAIR mode: a voxel's frames over a lane pair -------------------------------------------------------------------------------
template <int W>
struct K23PairRegs {
  static constexpr int H1 = (W + 1) / 2;
  double fx[10], fr[H1][10], Up[9], coe;
};
// everything the residual half reads, requested before the poses exist.  hoff = the lane's half: 0, or H1 frames of ten planes further on
// (odd W: the upper lane's last slot then reads the fix planes -- in range, and masked in k23_pair_finish)
template <int W>
__device__ __forceinline__ void k23_pair_issue(__amdgpu_buffer_rsrc_t rs, unsigned vs8, unsigned a8, unsigned hoff, K23PairRegs<W>& r) {
  using P = K23Planes<W>;
#pragma unroll
  for (int k = 0; k < 10; k++) r.fx[k] = k3_ld64<0>(rs, a8, (unsigned)(P::FIX + k) * vs8);
#pragma unroll
  for (int i = 0; i < K23PairRegs<W>::H1; i++)
#pragma unroll
    for (int k = 0; k < 10; k++) r.fr[i][k] = k3_ld64<0>(rs, a8 + hoff, (unsigned)(10 * i + k) * vs8);
  r.coe = k3_ld64<0>(rs, a8, (unsigned)P::COE * vs8);
#pragma unroll
  for (int col = 0; col < 3; col++)
#pragma unroll
    for (int row = 0; row < 3; row++) r.Up[3 * row + col] = k3_ld64<0>(rs, a8, (unsigned)(P::EIGVEC + 3 * col + row) * vs8);
}
// the partner lane's value (lanes 2j <-> 2j + 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ double k23_partner(double v) {
  const v2i x = __builtin_bit_cast(v2i, v);
  v2i y;
  y[0] = __builtin_amdgcn_update_dpp(0, x[0], 0xB1, 0xf, 0xf, false);
  y[1] = __builtin_amdgcn_update_dpp(0, x[1], 0xB1, 0xf, 0xf, false);
  return __builtin_bit_cast(double, y);
}
// The residual half of one voxel on its lane pair.  The merged cluster is (fix + frames [0, H1)) + (frames [H1, W)) -- a different
// association from the one-lane form's running sum: round-off apart (tests/test_gpu_parity.py compares the two forms of the loop to 1e-9).
// Returns coe * lambda_0 on the lower lane, 0 on the upper.
// Instrumented build (DBG), stamps of wave gw: 4 rows landed, 15 transform + exchange done, 18 eigen-decomposition done, 21 record + cache stores issued.
template <int W, bool DBG = false>
__device__ __forceinline__ double k23_pair_finish(__amdgpu_buffer_rsrc_t rs, unsigned vs8, unsigned a8, bool valid, bool upper, const double* pose_lds, K23PairRegs<W>& r,
                                                  double* rec, int gw = 0) {
  using P = K23Planes<W>;
  constexpr int H1 = K23PairRegs<W>::H1;
  double SP[6], Sv[3], SN, C[6], lam[3] = {0.0, 0.0, 0.0}, U[9];
#pragma unroll
  for (int k = 0; k < 6; k++) SP[k] = upper ? 0.0 : r.fx[k];
#pragma unroll
  for (int k = 0; k < 3; k++) Sv[k] = upper ? 0.0 : r.fx[6 + k];
  SN = upper ? 0.0 : r.fx[9];
  if (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_stamp(true, gw, 4); }
#pragma unroll
  for (int i = 0; i < H1; i++) {
    const bool slot = H1 + i < W;                      // compile-time: false only for the upper lane's last slot at odd W
    const double* pp = pose_lds + 12 * ((upper && slot) ? H1 + i : i);
    double R[9], p[3];
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) R[3 * rr + cc] = pp[3 * cc + rr];
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = pp[9 + k];
    double ci[10];
    const bool obs = r.fr[i][9] != 0.0 && (slot || !upper);   // N == 0: frame did not observe the voxel (voxel_map.hpp:258)
#pragma unroll
    for (int k = 0; k < 10; k++) ci[k] = obs ? r.fr[i][k] : 0.0;
    vxm::transform_accumulate(ci, ci + 6, ci[9], R, p, SP, Sv, SN);
  }
#pragma unroll
  for (int k = 0; k < 6; k++) SP[k] += k23_partner(SP[k]);
#pragma unroll
  for (int k = 0; k < 3; k++) Sv[k] += k23_partner(Sv[k]);
  SN += k23_partner(SN);
  double invN;
  k23_cov(SP, Sv, SN, C, invN);
  if (DBG) { asm volatile("" :: "v"(C[0]), "v"(C[3]), "v"(C[5])); dbg_stamp(true, gw, 15); }
  vxm::eig_sym3_warm(C, r.Up, lam, U);
  if (DBG) { asm volatile("" :: "v"(lam[0]), "v"(U[0]), "v"(U[8])); dbg_stamp(true, gw, 18); }
  const double coe = r.coe;
  double v[18];
  k23_record<true>(v, lam, U, Sv, invN, coe);
  if (rec) {   // the lower lane writes the first nine doubles of the record, the upper lane the other nine
#pragma unroll
    for (int k = 0; k < 9; k++) rec[(upper ? 9 : 0) + k] = upper ? v[9 + k] : v[k];
  }
  if (valid) {
    // the 26 cache planes behind eigval in plane order; the lower lane stores planes 0..12, the upper lane 13..25 (its offset carries the 13 planes)
    double c26[26];
#pragma unroll
    for (int k = 0; k < 3; k++) c26[k] = lam[k];
#pragma unroll
    for (int k = 0; k < 9; k++) c26[3 + k] = v[k];
#pragma unroll
    for (int k = 0; k < 6; k++) c26[12 + k] = SP[k];
#pragma unroll
    for (int k = 0; k < 3; k++) c26[18 + k] = Sv[k];
    c26[21] = SN; c26[22] = v[9]; c26[23] = v[10]; c26[24] = v[14]; c26[25] = v[15];
    const unsigned so = a8 + (upper ? 13u * vs8 : 0u);
#pragma unroll
    for (int j = 0; j < 13; j++) k23_st64(rs, so, (unsigned)(P::EIGVAL + j) * vs8, upper ? c26[13 + j] : c26[j]);
  }
  dbg_stamp(DBG, gw, 21);
  return (valid && !upper) ? coe * lam[0] : 0.0;
}

// Argument order: the first 14 dwords are preloaded into SGPRs at wave launch (see k3_hessian_kernel) -- what the solve workgroup and a
// sweep workgroup's first requests need.  flags: bit 0 = test hook (the sweep workgroups give up waiting for the solve at once).
// Grid: 1 + nwg workgroups of K3_BLOCK threads (workgroup 0 = the solve on four of its waves); LDS: k23_lds_bytes<W>().
template <int W, bool DBG = false, bool MIXED = false, bool PAIR = false>
__global__ __launch_bounds__(K3_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k23_fused_kernel(LMState* __restrict__ st, int c, unsigned seq, const double* li_rec, double* li_out,
                                                             const double* host_feed, int head, int end, int nwg, int flags_hs,
                                                             double* __restrict__ planes, const double* __restrict__ clb, int VS, double* __restrict__ partial2,
                                                             double* __restrict__ partial3) {
  using C = K3Cfg<W>;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int head_start = flags_hs >> 8, flags = flags_hs & 0xff;

  if (blockIdx.x == 0) {
    // ---- the damped solve of this iteration (or the host's trial poses: LiDAR-inertial shells), exactly as in k2_residual_kernel ----
    if (wave >= S4_WAVES) return;
    if (host_feed) {
      if (st->ctl[c].done) return;
      if (wave != 0) return;
      const volatile double* hf = host_feed;
      unsigned spins = 0;
      bool fed = true;
      while ((unsigned)hf[0] != seq) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > K2_SPIN_LIMIT) { fed = false; break; }
      }
      if (!fed) { if (lane == 0) st->error = 2; return; }
      for (int k = lane; k < 12 * W; k += 64) __hip_atomic_store(&st->ctl[c].xt[k], hf[1 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const int left = lm_solve_body4<W, DBG>(st, c, lds, li_rec, li_out, seq);
      if (wave != 0) return;
      if (left) {   // the loop had been left before this launch: the sweep workgroups leave too; the control block moves on unchanged
        if (!(flags & 2)) lm_carry_wave(st, c, W);
        return;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trial poses went out as written-through agent-scope stores: acknowledged, then published
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(&st->solve_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (flags & 2) return;
    // ---- the step's accept / reject decision, while the sweep workgroups work: their residual sums replace the NaN in partial2[0 .. nwg)
    // (nwg <= 256: four slots per lane).  Summed in lm_residual2's order: every reader of these slots gets the same bits.
    LMResidual2Loads L;
#pragma unroll
    for (int k = 0; k < 16; k++) L.v[k] = 0.0;
    unsigned spins = 0;
    for (;;) {
      bool missing = false;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = 64 * k + lane;
        if (i < nwg) {
          L.v[k] = __hip_atomic_load(&partial2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          missing |= L.v[k] != L.v[k];
        }
      }
      if (__builtin_amdgcn_ballot_w64(missing) == 0) break;
      __builtin_amdgcn_s_sleep(16);
      if (++spins > 4u * K2_SPIN_LIMIT || ((flags & 1) && spins > 64u)) {   // a sweep workgroup gave up (or the test hook): the host retries without fusion
        if (lane == 0) st->error = 1;
        return;
      }
    }
    LMPending pd;
    pd.pending = 1; pd.restart = 0; pd.d_scalar = nullptr; pd.partial = partial2; pd.nparts = nwg;
    const double r2 = lm_residual2_finish(pd, L);
    // every sum has been read: the slots go back to NaN for the next fused launch (the reduction kernel behind THIS launch no longer does it:
    // 0.7 us of its 7 at the head of its last workgroup; the one behind a stand-alone Hessian sweep -- first iteration of a solve -- still does)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = 64 * k + lane;
      if (i < nwg) __hip_atomic_store(&partial2[i], __builtin_nan(""), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const LMDecision d = lm_decide(st->ctl[c], r2, 0);
    lm_persist_wave(st, c, d, W);
    return;
  }
  const int g = (int)blockIdx.x - 1;
  const int gw = g * C::WAVES + wave;
  dbg_stamp(DBG, gw, 0);
  if (st->ctl[c].done) return;
  // let the solver's first (dependent) loads through before a thousand waves put 50 MB of requests in front of them
  for (int k = 0; k < head_start; k += 10) __builtin_amdgcn_s_sleep(10);

  // this workgroup's voxels: the batches the Hessian half gives it (k3_sweep_body), cut to [head, end)
  const int b0 = head / C::NV, b1 = (end - 1) / C::NV;
  const int nb_all = b1 - b0 + 1;
  const int q = nb_all / nwg, rem = nb_all % nwg;
  const int cnt = q + (g < rem ? 1 : 0);
  const int bs = b0 + g * q + (g < rem ? g : rem);
  const int v0 = bs * C::NV > head ? bs * C::NV : head;
  const int v1 = cnt > 0 ? ((bs + cnt) * C::NV < end ? (bs + cnt) * C::NV : end) : v0;

  const __amdgpu_buffer_rsrc_t rs = k3_rsrc(planes);
  const unsigned vs8 = (unsigned)VS * 8u;
  double* poseA = lds + 2 * C::BUF;     // where k3_sweep_body expects the poses
  double* lmv = poseA + 24 * W;         // 8 doubles the Hessian half does not use in a fused launch: the waves' residual sums
  __shared__ int k23_gave_up;

  // the Hessian half's staging corners (k3_sweep_body: lmv + 8 + wave * WAVE_DOUBLES): zeros first -- a slot of the first eight batches that
  // no voxel of [head, end) fills must read as a finite record (its lanes are masked by coe = sqrt(coe) = 0, which only works on finite numbers)
  double* const stage_all = lmv + 8;
  constexpr int STAGE_W = K3Stage<W>::WAVE_DOUBLES, REC = K3Stage<W>::REC;
  for (int k = tid; k < C::WAVES * STAGE_W; k += K3_BLOCK) stage_all[k] = 0.0;
  // where voxel a's record goes: batch a / NV is the first batch of wave (a / NV - bs) when that is < 8
  auto record_of = [&](int a, bool valid) __attribute__((always_inline)) -> double* {
    const int bi = a / C::NV - bs;
    return (valid && bi >= 0 && bi < C::WAVES) ? stage_all + bi * STAGE_W + (a - (a / C::NV) * C::NV) * REC : nullptr;
  };

  K3Planes pl;
  pl.cache_ptr = nullptr; pl.coe_ptr = nullptr; pl.clb = clb; pl.vs8 = vs8;
  double c0[10];
#pragma unroll
  for (int k = 0; k < 10; k++) c0[k] = 0.0;
  // The Hessian half's first batch of cluster rows does not depend on anything computed here
  auto load_first = [&]() __attribute__((always_inline)) { if (wave < cnt) k3_load_clusters(pl, bs + wave, lane, c0); };

  // ONE wave per workgroup polls for the trial poses and fetches them for all eight (k2_residual_kernel: relaxed polls, no acquire fence;
  // the poses are read with system-coherent loads issued after the poll that saw `seq`)
  auto wait_for_poses = [&]() __attribute__((always_inline)) -> bool {
    if (wave == 0) {
      unsigned spins = 0;
      const unsigned spin_limit = (flags & 1) ? 1u : 4u * K2_SPIN_LIMIT;
      bool seen = true;
      while (__hip_atomic_load(&st->solve_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > spin_limit) { seen = false; break; }
      }
      if (seen) {
        const volatile double* xt = st->ctl[c].xt;
        if (lane < 12 * W) poseA[lane] = xt[lane];
        if (lane + 64 < 12 * W) poseA[lane + 64] = xt[lane + 64];
      } else if (lane == 0) st->error = 1;
      if (lane == 0) k23_gave_up = seen ? 0 : 1;
    }
    __syncthreads();
    return k23_gave_up == 0;
  };

  double res = 0.0;
  if constexpr (PAIR) {
    // ---- a voxel per lane pair, one pass (the launcher picks this instantiation when no workgroup owns more than 256 voxels); wave w takes
    // voxels [32 w, 32 w + 32) of the workgroup's run
    const bool upper = (lane & 1) != 0;
    const int a = v0 + wave * 32 + (lane >> 1);
    const bool valid = a < v1;
    const bool wave_has = v0 + wave * 32 < v1;   // wave-uniform
    const unsigned a8 = (unsigned)(valid ? a : head) * 8u;
    K23PairRegs<W> rg;
    if (wave_has) k23_pair_issue<W>(rs, vs8, a8, upper ? (unsigned)(10 * K23PairRegs<W>::H1) * vs8 : 0u, rg);
    load_first();
    dbg_stamp(DBG, gw, 30);
    if (!wait_for_poses()) return;
    dbg_stamp(DBG, gw, 5);
    if (wave_has) res = k23_pair_finish<W, DBG>(rs, vs8, a8, valid, upper, poseA, rg, record_of(a, valid), gw);
  } else {
    // wave w of a pass takes voxels [64 w, 64 w + 64) of it; a wave without voxels skips the residual half altogether -- its arithmetic would
    // share a SIMD's fp64 issue with a wave that has voxels
    K23Regs<W> rg;
    int a = v0 + wave * 64 + lane;
    bool valid = a < v1;
    const bool wave_has = v0 + wave * 64 < v1;   // wave-uniform
    if (wave_has) k23_issue<W>(rs, vs8, (unsigned)(valid ? a : head) * 8u, rg);
    // a wave without voxels asks for the Hessian half's first batch now, the others between their eigen-decomposition and their cache stores
    // (one pass) or behind their last pass
    const bool single_pass = v1 - v0 <= K3_BLOCK;
    if (!wave_has) load_first();
    dbg_stamp(DBG, gw, 30);
    if (!wait_for_poses()) return;
    dbg_stamp(DBG, gw, 5);

    // ---- residual half: 512 voxels per pass; later passes request their rows when they start (nothing of a pass is carried across the
    // loop's back edge: the ring would become 100 registers of phi copies)
    if (wave_has) {
      auto hook = [&]() __attribute__((always_inline)) { if (single_pass) load_first(); };
      res = k23_finish<W, false>(rs, vs8, (unsigned)(valid ? a : head) * 8u, valid, poseA, rg, record_of(a, valid), gw, hook);   // (stamps 15 / 18 / 21 inside it: 250 spilled registers in the instrumented build -- off)
    }
    for (int base = v0 + K3_BLOCK; base < v1; base += K3_BLOCK) {
      if (base + wave * 64 >= v1) break;   // wave-uniform: the later waves of the last pass have nothing
      a = base + wave * 64 + lane;
      valid = a < v1;
      K23Regs<W> rn;
      k23_issue<W>(rs, vs8, (unsigned)(valid ? a : head) * 8u, rn);
      res += k23_finish<W>(rs, vs8, (unsigned)(valid ? a : head) * 8u, valid, poseA, rn);
    }
    if (wave_has && !single_pass) load_first();
  }
  k3_clear_pads<W>(lds, tid);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) res += __shfl_down(res, off);
  if (lane == 0) lmv[wave] = res;
  // (the cache stores are NOT waited for here: the first phase A of the Hessian half reads its plane parameters from the staging corners;
  // k3_sweep_body waits in front of the first step's barrier)
  dbg_stamp(DBG, gw, 24);
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < C::WAVES; w++) s += lmv[w];
    __hip_atomic_store(&partial2[g], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the solve workgroup is waiting for it
  }
#ifdef VXBA_K23_DBG_BARRIER2
  __syncthreads();
#endif
  // ---- Hessian half over the same voxels ----
  PoseArg no_poses;      // never read: a fused launch takes no LM decision
  LMPending no_pend;
  const double* cache_planes = planes + (size_t)K23Planes<W>::EIGVAL * VS;
  const double* coe_plane = planes + (size_t)K23Planes<W>::COE * VS;
  k3_sweep_body<W, DBG, MIXED, true>(lds, clb, cache_planes, coe_plane, st, VS, head, end, c, 0, nwg, g, no_poses, no_pend, partial3, c0);
}
