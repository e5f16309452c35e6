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
//   (3) runs the residual half on those voxels -- transform, covariance, warm-started eigen-decomposition, bit for bit the arithmetic of
//       k2_residual_kernel -- writes the (lambda, U, merged, aux) cache planes with plain stores (they stay in the XCD's L2) and adds
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

// The residual half of one voxel: k2_residual_kernel's arithmetic, statement for statement (same frame order, same unfused / fused
// operations inside vxm::*), so that a window evaluates to the same cache bits whichever launch ran it.  Returns coe * lambda_0.
// Instrumented build (DBG), stamps of wave gw: 15 transform done, 18 eigen-decomposition done, 21 cache stores issued.
// hook(): called between the eigen-decomposition and the cache stores (the Hessian half's first cluster rows are requested there).
struct K23NoHook { __device__ __forceinline__ void operator()() const {} };
template <int W, bool DBG = false, class Hook = K23NoHook>
__device__ __forceinline__ double k23_finish(__amdgpu_buffer_rsrc_t rs, unsigned vs8, unsigned a8, bool valid, const double* pose_lds, K23Regs<W>& r, int gw = 0, Hook hook = Hook()) {
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

// Argument order: the first 14 dwords are preloaded into SGPRs at wave launch (see k3_hessian_kernel) -- what the solve workgroup and a
// sweep workgroup's first requests need.  flags: bit 0 = test hook (the sweep workgroups give up waiting for the solve at once).
// Grid: 1 + nwg workgroups of K3_BLOCK threads (workgroup 0 = the solve on four of its waves); LDS: k23_lds_bytes<W>().
template <int W, bool DBG = false, bool MIXED = false>
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

  // wave w of a pass takes voxels [64 w, 64 w + 64) of it; a wave without voxels (waves 4-7 at the metric's size: 196 voxels per workgroup)
  // skips the residual half altogether -- its arithmetic would share a SIMD's fp64 issue with a wave that has voxels
  K23Regs<W> rg;
  int a = v0 + wave * 64 + lane;
  bool valid = a < v1;
  const bool wave_has = v0 + wave * 64 < v1;   // wave-uniform
  if (wave_has) k23_issue<W>(rs, vs8, (unsigned)(valid ? a : head) * 8u, rg);
  // The Hessian half's first batch of cluster rows does not depend on anything computed here: a wave without voxels asks for it now, the
  // others between their eigen-decomposition and their cache stores (one pass) or behind their last pass.
  K3Planes pl;
  pl.cache_ptr = nullptr; pl.coe_ptr = nullptr; pl.clb = clb; pl.vs8 = vs8;
  double c0[10];
#pragma unroll
  for (int k = 0; k < 10; k++) c0[k] = 0.0;
  const bool single_pass = v1 - v0 <= K3_BLOCK;
  auto load_first = [&]() __attribute__((always_inline)) { if (wave < cnt) k3_load_clusters(pl, bs + wave, lane, c0); };
  if (!wave_has) load_first();
  dbg_stamp(DBG, gw, 30);

  // ONE wave per workgroup polls for the trial poses and fetches them for all eight (k2_residual_kernel: relaxed polls, no acquire fence;
  // the poses are read with system-coherent loads issued after the poll that saw `seq`)
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
  if (k23_gave_up) return;
  dbg_stamp(DBG, gw, 5);

  // ---- residual half: 512 voxels per pass (one pass at the metric's size: 196 voxels per workgroup); later passes request their rows
  // when they start (nothing of a pass is carried across the loop's back edge: the ring would become 100 registers of phi copies)
  double res = 0.0;
  if (wave_has) {
    auto hook = [&]() __attribute__((always_inline)) { if (single_pass) load_first(); };
    res = k23_finish<W, false>(rs, vs8, (unsigned)(valid ? a : head) * 8u, valid, poseA, rg, gw, hook);   // (stamps 15 / 18 / 21 inside it: 250 spilled registers in the instrumented build -- off)
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
  k3_clear_pads<W>(lds, tid);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) res += __shfl_down(res, off);
  if (lane == 0) lmv[wave] = res;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's cache stores have reached the L2
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
