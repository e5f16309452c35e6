// Launch interface between the C-ABI layer (vxba_capi.hip) and the gfx950 kernels (vxba_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vxk {

constexpr int MAXW = 10;          // VXBA_MAX_WIN: 6W <= 64 accumulator columns
constexpr int K3_BLOCK = 512;     // 8 waves per workgroup, one workgroup per CU: two waves per SIMD
constexpr int DACC = 28;          // per-frame linear accumulators: g(6) Drr(6) Drt(9) Dtt(6) residual(1)

// Poses travel as a kernel argument (W*96 B <= 960 B): uniform scalar loads in K2, one vector load per lane in K3.
struct PoseArg {
  double Rp[12 * MAXW];  // per frame: R column-major (9) | p (3)   -- the C-ABI pose format
};

// Data layout in HBM (all f64, plane stride VS = voxel capacity rounded up to 64):
//   cl     [W][10][VS]  body-frame clusters, frame-major planes: lane <-> voxel reads are contiguous
//   fix    [10][VS]     world-frame fix clusters
//   coe    [VS]
//   eigval [3][VS], eigvec [9][VS] (plane 3*col+row), merged [10][VS]   -- the (lambda, U, pcr_add) cache
//   aux    [4][VS]      s_1, s_2 (s_k = sqrt(2/(lambda_k - lambda_0))), 1/N_merged, sqrt(coe): derived, device-private
//   clb    [ceil(VS/NV)][5][64][2]  the SAME clusters again in K3's batch-major order: batch b = voxels
//                      [b NV, (b+1) NV), lane = voxel_local * W + frame, component pairs interleaved, so one wave
//                      reads its 60 entries with five fully contiguous 1 KB dwordx4 loads
struct FactorView {
  double* clb;
  double* cl;
  double* fix;
  double* coe;
  double* eigval;
  double* eigvec;
  double* merged;
  double* aux;
  float* cl32;     // f32 re-centred copy of the cluster planes (same frame-major layout), or nullptr: residual sweep of VXBA_OPT_F32_CLUSTERS
  int VS;
  int W;
};

// Device-resident state of the LM shell (Lidar_BA_Optimizer::damping_iter, voxel_map.hpp:367-442): the sweeps read
// their poses and their run/skip gates from here, so a whole damping_iter is enqueued without a host round trip.
// The small control block is double-buffered: the accept/reject decision of iteration i is taken in the PROLOGUE of
// iteration i+1's Hessian sweep (every workgroup recomputes it from ctl[c] and the residual sweep's output while its
// first loads are in flight; workgroup 0 persists the result into ctl[c^1]), which saves a kernel and a boundary per
// iteration without any inter-workgroup synchronisation.
constexpr int LM_MAX_ITER = 64;
struct LMCtl {
  double x[12 * MAXW];        // accepted poses            (x_stats)
  double xt[12 * MAXW];       // trial poses               (x_stats_temp)
  double u, v;                // damping
  double residual1, residual2, q1;
  double resis[2];            // residual before / after   (voxel_map.hpp:394-395,440)
  int calc_hess;              // is_calc_hess: gates the Hessian sweep
  int done;                   // loop left (early break): gates everything
  int iter;                   // iterations executed
  int converge;               // is_converge
  int rejected;               // last step rejected
  int bench_mode;             // 1: never take the early break, so exactly n_steps iterations run (vxba_lm_steps)
  int n_accept, n_reject;     // running totals over all iterations of this init
};
struct LMState {
  LMCtl ctl[2];
  // (behind the control blocks, so that vxba_lm_steps reads back [ctl | solve_seq | error] -- 4 KB -- and not the 66 KB of everything: LM_HEAD_BYTES)
  unsigned solve_seq;                     // sequence number of the last solve published inside a residual-sweep launch
  int error;                              // 1: a voxel workgroup gave up waiting for the solve (never observed)
  double trace[LM_MAX_ITER * 8];
  double Jwork[6 * MAXW];                 // gauge-fixed gradient kept across rejected steps
  double dxi[6 * MAXW];
  double Hwork[36 * MAXW * MAXW];         // gauge-fixed Hessian kept across rejected steps
  double hess_out[36 * MAXW * MAXW];      // *hess, exported before the gauge fix (voxel_map.hpp:391)
};
constexpr size_t LM_HEAD_BYTES = 2 * sizeof(LMCtl) + 8;   // ctl[2] | solve_seq | error
// What a sweep needs to take the pending accept/reject decision in its prologue.
struct LMPending {
  int pending;                // 1: ctl[c] awaits the decision of the step whose residual sweep just ran (taken in the sweep's prologue);
                              // 2 / 3: sharded speculative loop -- no decision in the sweep, linearise at the trial poses (2) or at the
                              // kernel-argument poses (3)
  int restart;                // 1: after the decision start a new window from restart_x0 (bench driver)
  const double* d_scalar;     // all-reduced residual2, or null: sum the nparts wave partials
  const double* partial;
  int nparts;
};

// ---- geometry of the Hessian sweep (vxba_k3.hpp, K3Cfg<W>) shared with the host side ----------------------------------------
// column groups of four (6W rounded up), pairs per wave (the 8 waves of a workgroup split the NG (NG + 1) / 2 group pairs)
__host__ __device__ constexpr int k3_groups(int W) { return (6 * W + 3) / 4; }
__host__ __device__ constexpr int k3_pairs_per_wave(int W) { return (k3_groups(W) * (k3_groups(W) + 1) / 2 + 7) / 8; }
// row stride of the LDS tile in doubles: the smallest value >= 4 NG that is 4 (mod 8) -- conflict-free operand reads
__host__ __device__ constexpr int k3_row_stride(int W) { int rs = 4 * k3_groups(W); while (rs % 8 != 4) rs++; return rs; }
// voxels per wave-batch: even (a step's 24 NV rows are whole 16-row slabs), at most 64 / W lanes' worth and 12, and two tile buffers of
// 8 batches within 140 KB of LDS (the poses, the parameter staging areas and the epilogue share the rest)
__host__ __device__ constexpr int k3_nv(int W) {
  int n = 64 / W < 12 ? 64 / W : 12;
  n &= ~1;
  while (n > 2 && (size_t)2 * 24 * n * k3_row_stride(W) * 8 > (size_t)140 * 1024) n -= 2;
  return n;
}
__host__ __device__ constexpr size_t k3_clb_len(int W, int VS) { return (size_t)((VS + k3_nv(W) - 1) / k3_nv(W)) * 640; }
// doubles per workgroup partial: [8 waves][pairs per wave rounded up to 4][16] sums of S + per-frame linear accumulators
__host__ __device__ constexpr size_t k3_partial_len(int W) { return (size_t)8 * ((k3_pairs_per_wave(W) + 3) & ~3) * 16 + (size_t)W * DACC; }

// K2: residual sweep over voxels [head,end): merge + covariance + eigen-decomposition, writes the cache,
// block partials of sum coe*lambda_0 into d_partial[0..nblocks).  Returns the number of partials.
// st != null: LM mode -- poses are ctl[c].xt and the sweep skips itself on the GPU once the loop is done; else `poses`.
// One lane per voxel, k2_voxels_per_block(...) in [32, 64] voxels per 64-lane workgroup (balanced over `cus` CUs).
int k2_voxels_per_block(int nvox, int cus);
// fused_seq != 0 (LM mode only): workgroup 0 of the launch runs the damped solve of this iteration and publishes `fused_seq`;
// the voxel workgroups wait for it after requesting their cluster rows.  Must be unique per launch and non-zero.
// host_feed (with fused_seq != 0): workgroup 0 does not solve; it waits until the host has written fused_seq to host_feed[0] (mapped host
// memory) and copies the 12W trial poses behind it into ctl[c].xt -- the LiDAR-inertial shells queue the sweep before their own solve is done.
// residual-sweep geometry: voxels per wave for an option value, and the number of wave partials a sweep over nvoxels writes
int k2_voxels_per_wave(int voxels_per_block);
int k2_nparts(int nvoxels, int voxels_per_block);
int launch_k2_residual(const FactorView& fv, const PoseArg& poses, LMState* st, int c, unsigned fused_seq, int head, int end, double* d_partial,
                       int voxels_per_block, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, const double* host_feed = nullptr,
                       const double* li_rec = nullptr, double* li_out = nullptr);
// li_rec / li_out (with fused_seq != 0, host_feed == nullptr): workgroup 0 solves the LiDAR-inertial shells' REDUCED pose system -- li_rec (mapped host memory,
// complete before the launch) = [u | current poses 12W | e 6W | E (6W)^2 column-major], see vxba_solve4.hpp -- and writes [dx 6W | trial poses 12W | seq] to li_out.
inline int li_rec_doubles(int W) { return 1 + 12 * W + 6 * W + 36 * W * W; }
inline int li_out_doubles(int W) { return 6 * W + 12 * W + 1; }
// Deterministic sum of n partials into d_out[0].
void launch_sum_partials(const double* d_partial, int n, double* d_out, hipStream_t s);
// Derive aux (gap scales) from eigval for voxels [head,end) (after a caller-seeded cache).
void launch_seed_aux(const FactorView& fv, int head, int end, hipStream_t s);

// K3: Hessian/gradient sweep over voxels [head,end) into per-workgroup partials; returns #workgroups.
int k3_grid_blocks(int device_cus);
// workgroups for a sweep over nbatches wave-batches: one 8-wave workgroup per CU; small sweeps are spread over as many CUs as they
// have pairs of batches (a workgroup with fewer than 8 batches runs them as one ragged step, whose MFMA work is re-split over its waves)
inline int k3_blocks_for(int nbatches, int device_cus) { const int b = (nbatches + 1) / 2; return b < 1 ? 1 : (b < device_cus ? b : device_cus); }
// cache_src (nullable): read the (lambda, U, merged, aux) cache planes from this base instead of fv's live cache --
// used to start a new window from the snapshot without copying it back first.
// ev_start / ev_stop (nullable): events tied to this dispatch's own begin / end timestamps (hipExtLaunchKernel), i.e.
// the same interval rocprofv3 reports for the kernel -- events recorded around a launch also count the dispatch gap.
// st != null: LM mode -- the sweep first takes the pending accept/reject decision (LMPending) from ctl[c_in] into
// mixed != 0: f32 products on the matrix cores, f64 accumulation (BASELINE configs[2]).
// ctl[c_in ^ 1] (if pend.pending; else it runs on ctl[c_in] as is), linearises at the decided poses and skips itself
// when the decision says so (rejected step / loop done).  `poses` doubles as the restart poses of the bench driver.
int launch_k3_hessian(const FactorView& fv, const PoseArg& poses, LMState* st, int c_in, const LMPending& pend, const double* cache_src,
                      int head, int end, double* d_partial, int nblocks, int mixed, hipStream_t s, hipEvent_t ev_start = nullptr,
                      hipEvent_t ev_stop = nullptr);
// Cross-workgroup reduction + assembly of the packed [Hess (6W)^2 col-major | JacT 6W | residual] buffer.
// force: run even when the state says the sweep was not needed (sharded speculative loop); k2_partial: also leave the sum of the
// residual sweep's wave partials in d_packed[(6W)^2 + 6W + 1].
// reset_slots (nullable): n_reset doubles set to NaN -- the residual slots the solve workgroup of the next fused launch waits on (launch_k23_fused).
void launch_k3_finalize(const double* d_partial, int nblocks, int W, LMState* st, int c, int write_state, double* d_packed, hipStream_t s, int force = 0,
                        const double* k2_partial = nullptr, int k2_nparts = 0, double* reset_slots = nullptr, int n_reset = 0);
// K2 + K3 in one launch behind the in-launch solve (vxba_k23.hpp): workgroup 0 solves ctl[c] and publishes `seq`; nwg sweep workgroups run the
// residual sweep at the trial poses over their run of voxels (one residual sum per workgroup into d_partial2[0..nwg)), then the Hessian sweep
// at the same poses over the same voxels (workgroup partials into d_partial3, k3_partial_len(W) doubles each).  Returns nwg, or -1 when the
// factor cannot take this path (f32 cluster rows, planes beyond 32-bit offsets, foreign plane layout).  flags bit 0: test hook (give up waiting at once).
int k23_sweep_blocks(int nbatches, int device_cus);
bool k23_supported(const FactorView& fv);
int launch_k23_fused(const FactorView& fv, LMState* st, int c, unsigned seq, int head, int end, double* d_partial2, double* d_partial3, int nwg, int mixed, int flags,
                     hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, const double* host_feed = nullptr, const double* li_rec = nullptr,
                     double* li_out = nullptr);
// The solve workgroup then waits for the nwg residual sums (d_partial2 must hold NaN in [0, nwg) when the launch starts: launch_k3_finalize's
// reset_slots), takes the step's accept / reject decision while the Hessian sweep runs and persists the decided control block into ctl[c ^ 1]
// (calc_hess = accepted): launch_k3_finalize on ctl[c ^ 1] behind the launch adopts the system of an accepted step and drops a rejected one's.
// flags bit 1: no decision in the launch (the caller's own shell decides).
constexpr int K23_MAX_SWEEP_BLOCKS = 256;
// Sharded speculative loop: decision for the pending trial from the reduced residual slot (into ctl[c_in ^ 1]) + adoption of the reduced system.
void launch_lm_spec_unpack(LMState* st, int c_in, const double* d_packed, int W, int has_pending, int restart, const PoseArg& x0, hipStream_t s);
// Voxel-sharded LM loop: fill the LM state (Hwork, Jwork, hess_out, residual1) from the ALL-REDUCED packed buffer (write_state
// = 0 above); gated on the state's flags like the sweep itself.
void launch_lm_unpack(LMState* st, int c, const double* d_packed, int W, hipStream_t s);

// LM shell on the device: init (poses, damping, flags into ctl[0]), damped solve + trial state on ctl[c], and the
// stand-alone decision kernel that closes the loop (ctl[c_in] -> ctl[c_in ^ 1]).
void launch_lm_init(LMState* st, const PoseArg& x0, int W, int bench_mode, hipStream_t s);
void launch_lm_reset(LMState* st, hipStream_t s);   // control scalars of both blocks and the error word to zero (see the kernel)
void launch_lm_solve(LMState* st, int c, int W, hipStream_t s);
void launch_lm_update(LMState* st, int c_in, const LMPending& pend, const PoseArg& restart_x0, int W, hipStream_t s);

// K1: clusters of n_voxels*W cells from bucketed points (cell = frame*n_voxels + voxel), written to the
// frame-major planes at voxel offset v0.
void launch_k1_build(const double* d_xyz, const int64_t* d_cell_ptr, int n_voxels, int W, const FactorView& fv, int v0, hipStream_t s);

// K4: plane fit of n packed clusters (AoS n*10) -> eig_val n*3, eig_vec n*9 (col-major); optional plane criteria flags.
struct PlaneCriteria { int min_point; double min_eigen_value, eigen_ratio_thre, factor_ratio_max; };
void launch_k4_plane_fit(const double* d_clusters, int64_t n, double* d_eigval, double* d_eigvec, const PlaneCriteria* crit, unsigned char* d_flags,
                         hipStream_t s);
// K1 stand-alone: n_cells buckets -> packed clusters (AoS n_cells*10).
void launch_k1_build_aos(const double* d_xyz, const int64_t* d_cell_ptr, int64_t n_cells, double* d_clusters, hipStream_t s, hipEvent_t ev_start = nullptr,
                         hipEvent_t ev_stop = nullptr);

// Rebuild the batch-major copy (clb) of voxels [v0, v0+n) from the frame-major planes.
void launch_build_clb(const FactorView& fv, int v0, int n, hipStream_t s);
void launch_build_cl32(const FactorView& fv, int v0, int n, hipStream_t s);

// Layout plumbing between the C-ABI's packed AoS rows and the device planes.
void launch_scatter_clusters(const double* d_src /*[n][W][10]*/, const FactorView& fv, int v0, int n, hipStream_t s);
// CSR clusters (row_ptr n + 1, frame_idx nnz, clusters nnz x 10; frames strictly increasing per voxel) into the frame-major planes at v0;
// *d_bad is set to 1 on a malformed row.
void launch_scatter_clusters_csr(const long long* d_row_ptr, const int* d_frame_idx, const double* d_clusters, const FactorView& fv, int v0, int n, int* d_bad,
                                 hipStream_t s);
void launch_gather_clusters(const FactorView& fv, int head, int n, double* d_dst /*[n][W][10]*/, hipStream_t s);
void launch_scatter_rows(const double* d_src /*[n][K]*/, double* planes, int VS, int v0, int n, int K, hipStream_t s);
void launch_scatter_voxel_records(const double* d_fix, const double* d_coe, const double* d_eigval, const double* d_eigvec, const double* d_merged, const FactorView& fv, int v0, int n,
                                  hipStream_t s);   // fix 10 | coe 1 | eigval 3 | eigvec 9 | merged 10 of n pushed voxels, one launch
void launch_gather_rows(const double* planes, int VS, int head, int n, int K, double* d_dst /*[n][K]*/, hipStream_t s);
void launch_fill(double* p, size_t n, double val, hipStream_t s);
void launch_copy_planes(const double* src, int src_vs, double* dst, int dst_vs, int nplanes, int n, hipStream_t s);
void debug_read_stamps(unsigned long long* host, size_t n);
void debug_clear_stamps();
void launch_mfma_probe(const double* dA, const double* dB, double* dD, hipStream_t s);
void launch_count_nnz(const FactorView& fv, int V, unsigned long long* d_out, hipStream_t s);

}  // namespace vxk
