/* vxba.h -- C ABI of the MI355X-native LiDAR bundle-adjustment factor (libvxba.so).
 *
 * Drop-in boundary for the local-mapping BA hot path of hku-mars/Voxel-SLAM: everything
 * `class LidarFactor` (VoxelSLAM/src/voxel_map.hpp:109-290) does, plus the optimizer shells that
 * own the loop (`Lidar_BA_Optimizer` voxel_map.hpp:293-444, `LI_BA_Optimizer[Gravity]` :446-864 with the
 * `IMU_PRE` factor of preintegration.hpp), plus the batch factor construction of the hierarchical BA
 * (`OctreeGBA`, loop_refine.hpp:273-476), behind plain pointers and sizes.
 * The reference has no FFI; these are the entry points a binding of that class would need
 * (INTEGRATION.md shows the adapter a maintainer would add to voxel_map.hpp).
 *
 * All kernels behind this header are hand-written HIP for gfx950 (CDNA4).  There is NO CPU
 * fallback: every entry point that needs the GPU returns VXBA_ERR_HIP / VXBA_ERR_NODEV if the
 * device or the HIP runtime is unavailable.
 *
 * Packed formats (all f64, caller-owned, host memory unless the name says `_device`):
 *   cluster : 10 f64  [Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz N]        <- PointCluster, tools.hpp:304-365
 *                     (P symmetric; N stored as an exact integer-valued double; N == 0 means
 *                      "frame did not observe this voxel", voxel_map.hpp:178,217,221,258)
 *   pose    : 12 f64  [R column-major (9) | p (3)]                 <- IMUST::R, IMUST::p, tools.hpp:139-140
 *   eig_val :  3 f64  ascending                                     <- LidarFactor::eig_values
 *   eig_vec :  9 f64  column-major, column k = eigenvector k        <- LidarFactor::eig_vectors
 *   Hess    : (6W)x(6W) f64 column-major (Eigen::MatrixXd layout), JacT : 6W f64
 *             per-frame tangent order [dphi(3); dp(3)], R <- R Exp(dphi), p <- p + dp (tools.hpp:154-162)
 */
#ifndef VXBA_H
#define VXBA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vxba_factor vxba_factor; /* opaque; one per `LidarFactor` instance */

enum {
  VXBA_OK = 0,
  VXBA_ERR_ARG = 1,    /* bad argument (null pointer, range, win_size) */
  VXBA_ERR_HIP = 2,    /* a HIP runtime call failed; see vxba_last_error */
  VXBA_ERR_NODEV = 3,  /* no usable gfx950 device */
  VXBA_ERR_STATE = 4,  /* call not legal in the current state (e.g. empty factor) */
  VXBA_ERR_UNSUPPORTED = 5
};

/* Largest window of the MFMA sweeps and the device-resident LM loop (6W <= 64 columns of the accumulator tile set). */
#define VXBA_MAX_WIN 10
/* Wide windows (VXBA_MAX_WIN < win_size <= VXBA_MAX_WIN_WIDE; the top level of the hierarchical BA optimises ~100 submap poses,
 * voxelslam.cpp:2485-2595): same entry points, sparse-incidence sweeps (pair-major Hessian assembly over an incidence index built
 * once per factor content; deterministic), LM shell on the host.  Not available there: vxba_lm_steps, the LiDAR-inertial shells,
 * mixed precision. */
#define VXBA_MAX_WIN_WIDE 128

/* ---- lifetime -------------------------------------------------------------------------------- */
/* LidarFactor(int _w)                                   voxel_map.hpp:120 */
int vxba_create(int win_size, int device, vxba_factor** out);
int vxba_destroy(vxba_factor* f);
/* LidarFactor::clear()                                  voxel_map.hpp:281-286 */
int vxba_clear(vxba_factor* f);
/* `voxhess.win_size = ...` (voxelslam.cpp:623,1609); only legal on an empty factor */
int vxba_set_win_size(vxba_factor* f, int win_size);
int vxba_win_size(const vxba_factor* f);
/* plvec_voxels.size()                                   voxel_map.hpp:314,344 */
int vxba_size(const vxba_factor* f);
/* Run on a caller-owned hipStream_t instead of the factor's own (non-blocking) stream.  NULL selects the factor's own stream again --
 * so the legacy default stream (handle 0, e.g. torch's default current stream) cannot be chosen: work queued there is NOT ordered with the
 * factor's kernels; share a created stream instead (voxel_slam_amd/dist.py does). */
int vxba_set_stream(vxba_factor* f, void* hip_stream);
int vxba_reserve(vxba_factor* f, int n_voxels);
const char* vxba_last_error(const vxba_factor* f);

/* ---- factor construction --------------------------------------------------------------------- */
/* Batched LidarFactor::push_voxel (voxel_map.hpp:122-130): appends n voxels.
 *   clusters n*W*10 (voxel-major, frames chronological, voxel_map.hpp:1317-1319), fix n*10 (world frame),
 *   coe n (>= 0), eig_val n*3, eig_vec n*9, merged n*10 -- the (lambda, U, pcr_add) cache seeded by the
 *   caller (recut's eigen-decomposition, voxel_map.hpp:1161-1163).  eig_val/eig_vec/merged may be NULL:
 *   the cache is then undefined until vxba_evaluate_only_residual has run. */
int vxba_push_voxels(vxba_factor* f, int n, const double* clusters, const double* fix, const double* coe,
                     const double* eig_val, const double* eig_vec, const double* merged);
/* The same for sparse incidence (the hierarchical BA's top level, loop_refine.hpp:358-405, pushes voxels seen from a handful of
 * ~100 poses): voxel a's observed frames are frame_idx[row_ptr[a] .. row_ptr[a + 1]) (strictly increasing, < win_size), their clusters
 * clusters[e * 10 ..]; unlisted frames are unobserved.  row_ptr has n + 1 entries, row_ptr[0] = 0.  Works for any win_size; the planes
 * are filled on the device, so no dense n x W x 10 array exists on the host or crosses PCIe. */
int vxba_push_voxels_csr(vxba_factor* f, int n, const int64_t* row_ptr, const int32_t* frame_idx, const double* clusters, const double* fix, const double* coe,
                         const double* eig_val, const double* eig_vec, const double* merged);

/* K1 -- per-(voxel, frame) cluster accumulation from raw points, PointCluster::push (tools.hpp:326-331;
 * call sites voxel_map.hpp:988, loop_refine.hpp:383-385).  Appends n_voxels voxels whose body-frame
 * clusters are built on the GPU from points bucketed contiguously per cell:
 *   cell index = frame * n_voxels + voxel;  cell_ptr has W*n_voxels + 1 entries;  xyz_body n_points*3.
 *   fix / coe as in vxba_push_voxels (fix may be NULL = no fix clusters, coe NULL = all ones). */
int vxba_push_points(vxba_factor* f, int n_voxels, int64_t n_points, const double* xyz_body, const int64_t* cell_ptr,
                     const double* fix, const double* coe);

/* Read back the body-frame clusters of voxels [head,end) as n*W*10 (parity checks of K1). */
int vxba_read_clusters(vxba_factor* f, int head, int end, double* clusters);

/* ---- the two sweeps -------------------------------------------------------------------------- */
/* LidarFactor::acc_evaluate2 (voxel_map.hpp:132-241): Hessian / gradient / residual of voxels [head,end)
 * under poses Rp (W*12) using the CACHED (lambda, U, merged) of the last evaluate_only_residual / push.
 * Outputs are overwritten (the reference zeroes them, :134) and the lower block triangle is mirrored (:237-239). */
int vxba_acc_evaluate2(vxba_factor* f, const double* Rp, int head, int end, double* Hess, double* JacT, double* residual);

/* LidarFactor::evaluate_only_residual (voxel_map.hpp:243-279): world merge of the clusters under Rp,
 * 3x3 covariance, symmetric eigen-decomposition; WRITES the (lambda, U, merged) cache of [head,end);
 * residual = sum coe * lambda_0. */
int vxba_evaluate_only_residual(vxba_factor* f, const double* Rp, int head, int end, double* residual);

/* Same sweeps leaving the result in DEVICE memory (for an RCCL all-reduce across voxel shards):
 *   d_out of acc_evaluate2: packed [Hess (6W)^2 col-major | JacT 6W | residual 1] = (6W)^2 + 6W + 1 f64
 *   d_out of evaluate_only_residual: 1 f64.  Asynchronous on the factor's stream. */
int vxba_acc_evaluate2_device(vxba_factor* f, const double* Rp, int head, int end, double* d_out);
int vxba_evaluate_only_residual_device(vxba_factor* f, const double* Rp, int head, int end, double* d_out);
size_t vxba_packed_len(const vxba_factor* f); /* (6W)^2 + 6W + 1 */

/* Public members pcr_adds / eig_values / eig_vectors read by OctoTree::margi (voxel_map.hpp:1211-1222)
 * and motion_init (voxelslam.cpp:651-655).  Any output may be NULL. */
int vxba_read_cache(vxba_factor* f, int head, int end, double* eig_val, double* eig_vec, double* merged);
/* Keep / restore a device-side copy of the cache (a new window re-seeds the cache; used by bench.py). */
int vxba_snapshot_cache(vxba_factor* f);
int vxba_restore_cache(vxba_factor* f);

/* ---- K4: batched plane fit ---------------------------------------------------------------------- */
/* eig(pcr.cov()) for n clusters (OctoTree::recut voxel_map.hpp:1161-1163, margi :1242-1244,
 * OctreeGBA::recut loop_refine.hpp:363-366).  Stand-alone; runs on `device`. */
int vxba_plane_fit(int device, int64_t n, const double* clusters, double* eig_val, double* eig_vec);
/* The same with the reference's plane criteria evaluated on the GPU; flags[a] is a bit set:
 *   1 = enough points       N > min_point                                   (voxel_map.hpp:1155, loop_refine.hpp:360)
 *   2 = plane_judge         lambda0 < min_eigen_value && lambda0/lambda2 < eigen_ratio_thre   (voxel_map.hpp:1015-1019)
 *   4 = factor-worthy       lambda0/lambda1 <= factor_ratio_max (0.12 upstream)   (voxel_map.hpp:1314, loop_refine.hpp:378) */
int vxba_plane_fit_judge(int device, int64_t n, const double* clusters, int min_point, double min_eigen_value, double eigen_ratio_thre,
                         double factor_ratio_max, double* eig_val, double* eig_vec, uint8_t* flags);
/* Stand-alone K1: PointCluster::push over n_cells buckets of points (cell_ptr has n_cells + 1 entries) -> n_cells packed
 * clusters.  Same kernel arithmetic as vxba_push_points (bit-exact with the CPU); use it for world-frame fix clusters
 * (OctoTree::push_fix, voxel_map.hpp:996-1013) or to rebuild clusters from stored points (loop_refine.hpp:382-385). */
int vxba_build_clusters(int device, int64_t n_cells, int64_t n_points, const double* xyz, const int64_t* cell_ptr, double* clusters);

/* ---- the LM shell that owns the loop ------------------------------------------------------------ */
/* Collective hook for voxel-sharded multi-GPU BA: called on the packed device buffer after each sweep's
 * on-device reduction, before the host reads it.  Must sum `count` f64 across ranks in place, stream-ordered
 * on `hip_stream`.  NULL = single GPU. */
typedef int (*vxba_allreduce_fn)(void* ctx, double* d_buf, size_t count, void* hip_stream);
int vxba_set_allreduce(vxba_factor* f, vxba_allreduce_fn fn, void* ctx);
/* Direct RCCL collective (no per-sweep host callback): creates an RCCL communicator for this factor's shard group and
 * all-reduces the exchange buffers with ncclAllReduce on the factor's stream after every sweep.
 *   librccl_path : the librccl.so to dlopen -- pass the one the process already uses (e.g. torch/lib/librccl.so) so that
 *                  there is ONE RCCL instance in the process;
 *   unique_id    : 128 bytes from vxba_rccl_unique_id() on rank 0, distributed out of band (e.g. torch.distributed.broadcast).
 * Collective: every rank of the group must call it.  Takes precedence over vxba_set_allreduce. */
int vxba_rccl_unique_id(const char* librccl_path, void* unique_id_out_128);
int vxba_rccl_attach(vxba_factor* f, const char* librccl_path, int nranks, int rank, const void* unique_id_128);
/* One-shot all-reduce over the peers' mailboxes (point-to-point xGMI reads instead of a ring collective): at W <= 10 the exchange
 * buffer is 29 KB and a ring is pure latency.  Every rank: vxba_peer_export -> 64-byte IPC handle of its mailbox; gather all handles
 * (rank order) by any means; vxba_peer_attach.  From then on the sharded LM loop of vxba_damping_iter / vxba_lm_steps / the LI shells
 * sums its exchange buffer through the mailboxes (fixed rank order: bit-identical results on all ranks); RCCL / the hook stay the
 * fallback when this is not attached.  Needs peer access between the GPUs (one node) and HSA_ENABLE_IPC_MODE_LEGACY=0 in the
 * environment.  vxba_peer_status reports a peer that never arrived (the kernel's bounded wait gave up). */
#define VXBA_PEER_HANDLE_BYTES 64
#define VXBA_PEER_MAX 16
int vxba_peer_export(vxba_factor* f, void* handle_out);
int vxba_peer_attach(vxba_factor* f, int nranks, int rank, const void* handles /* nranks x VXBA_PEER_HANDLE_BYTES */);
int vxba_peer_detach(vxba_factor* f);
int vxba_peer_status(vxba_factor* f, int* status);
/* collective: one all-reduce of a known pattern through the mailboxes; *ok = 1 when every element came back as the exact sum */
int vxba_peer_selftest(vxba_factor* f, int* ok);
/* The id exchange through a caller-supplied broadcast (MPI_Bcast, a socket ...): bcast(ctx, buf, nbytes, root) returns 0 after
 * root's bytes are in every rank's buf.  librccl_path NULL: the RCCL the process already uses, else /opt/rocm/lib/librccl.so. */
typedef int (*vxba_bcast_fn)(void* ctx, void* buf, size_t nbytes, int root);
int vxba_rccl_attach_bcast(vxba_factor* f, const char* librccl_path, int nranks, int rank, vxba_bcast_fn bcast, void* ctx);
int vxba_rccl_detach(vxba_factor* f);

/* Let the caller own the exchange buffers the sweeps reduce into (e.g. a torch tensor, so torch.distributed /
 * RCCL can all-reduce it): d_packed holds vxba_packed_len() f64, d_scalar 1 f64.  NULL restores the internal ones.
 * If d_scalar == d_packed + vxba_packed_len() (one allocation of vxba_packed_len() + 1 doubles; the internal buffers are laid
 * out like that) the device-resident LM loop reduces both with ONE collective per iteration of count vxba_packed_len() + 1
 * (speculative form: the next Hessian sweep runs at the trial poses and is discarded if the step is rejected). */
int vxba_use_external_buffers(vxba_factor* f, double* d_packed, double* d_scalar);

/* One LM trace row: [residual1 residual2 u v q q1 accepted recomputed_hess] */
#define VXBA_TRACE_COLS 8

/* Lidar_BA_Optimizer::damping_iter (voxel_map.hpp:367-442).  Rp (W*12) in/out.  hess_out (6W)^2 col-major =
 * `*hess`, exported BEFORE the gauge fix (:391).  resis_out[2] = residual before / after (:394-395,440).
 * trace_out max_iter*VXBA_TRACE_COLS (may be NULL).  *is_converge as the reference's return value. */
int vxba_damping_iter(vxba_factor* f, double* Rp, int max_iter, double* hess_out, double* resis_out, double* trace_out,
                      int* n_trace, int* is_converge);

/* The same LM shell over caller-supplied sweeps (host-only code, needs no GPU): hess_fn must fill `packed`
 * (vxba_packed_len doubles: Hess col-major | JacT | residual) for poses Rp, resid_fn the residual -- e.g. a local voxel
 * shard's sweep followed by an all-reduce.  Used to drive voxel-sharded BA from torch.distributed and to test the
 * N > 1 logic with gloo on CPUs.  Semantics identical to vxba_damping_iter (voxel_map.hpp:367-442). */
typedef int (*vxba_hess_fn)(void* ctx, const double* Rp, double* packed);
typedef int (*vxba_resid_fn)(void* ctx, const double* Rp, double* residual);
int vxba_damping_iter_generic(int win_size, double* Rp, int max_iter, vxba_hess_fn hess_fn, vxba_resid_fn resid_fn, void* ctx,
                              double* hess_out, double* resis_out, double* trace_out, int* n_trace, int* is_converge);

/* Benchmark driver: exactly n_steps LM iterations of damping_iter WITHOUT the early break; every `steps_per_solve`
 * steps a new solve starts from Rp_init with u = 0.01, v = 2 and the snapshot cache restored (a new window).
 * A rejected step behaves like the reference (no Hessian recompute on the next iteration); stats_out[3] (may be NULL)
 * receives {iterations run, accepted steps, rejected steps}.
 * Rp_out (W*12) receives the final poses. */
int vxba_lm_steps(vxba_factor* f, const double* Rp_init, int n_steps, int steps_per_solve, double* Rp_out,
                  double* last_resis, int64_t* stats_out);

/* ---- batch factor construction: points of a window -> voxel hash -> octree subdivision -> plane test -> factor -------- */
/* OctreeGBA::cut_voxel + subdivide + recut (loop_refine.hpp:273-476), the per-round re-voxelisation of the hierarchical BA
 * (voxelslam.cpp:2374-2379), as sorts and segmented sums on the GPU.  A voxel becomes a factor at the coarsest octree layer
 * where it has more than min_points points (10 upstream), passes plane_judge (lambda0 < min_eigen_value and
 * lambda0/lambda2 < eigen_ratio[layer], the already inverted GBA/eigen_value_array), is seen from >= 2 frames and has
 * lambda0/lambda1 <= factor_ratio_max (0.12); non-planes are subdivided while layer < max_layer (<= 3). */
typedef struct vxba_voxelize_params {
  double voxel_size;
  int max_layer;
  int min_points;
  double min_eigen_value;
  double eigen_ratio[4];
  double factor_ratio_max;
  /* OctoTree's variant of the same construction -- the map build of motion_init (cut_voxel for every scan of the window, then one
   * recut + tras_opt, voxelslam.cpp:606-625; OctoTree::recut voxel_map.hpp:1148-1194, tras_opt :1308-1333): the point-count floor
   * depends on the layer (min_point[layer], voxelslam.cpp:812) and a single observing frame is enough. */
  int min_points_layer[4]; /* > 0: overrides min_points for that layer */
  int min_frames;          /* a factor needs at least this many observing frames: 2 for OctreeGBA (loop_refine.hpp:371-376), 0 for OctoTree */
  /* Voxel-sharded windows (one process per GPU; the reference shards the voxel list over threads, voxel_map.hpp:318-321, and builds per-thread
   * factors in OctreeGBA_multi_recut, loop_refine.hpp:483-537).  shard_count > 1: every rank passes the SAME points and keeps the root voxels
   * that hash to its shard_index -- filtered on the device right after the voxel keys are computed, so the sorts and the octree only see the
   * shard.  Whole root voxels stay together: the union of the ranks' factor voxels is the unsharded set, each voxel bit for bit.  0 / 1: off. */
  int shard_index;
  int shard_count;
} vxba_voxelize_params;
/* xyz_local: n_points x 3 body-frame points, frame by frame in cloud order; frame_ptr: win_size + 1 offsets; Rp: win_size*12.
 * Appends the factor voxels to f (coe = 1, no fix cluster, cache seeded with (lambda, U, world cluster) as recut's push_voxel
 * does) and returns their number in *n_pushed.  node_ids (optional, ids_capacity entries) receives one id per pushed voxel,
 * [x:16 | y:16 | z:16 | octant path:9 | 0:4 | layer:3] with the voxel coordinates offset by 32768, in push order (by layer,
 * ascending id inside a layer).  Voxel coordinates must stay within +-32768. */
int vxba_voxelize_push(vxba_factor* f, int64_t n_points, const double* xyz_local, const int64_t* frame_ptr, const double* Rp,
                       const vxba_voxelize_params* params, int64_t* n_pushed, uint64_t* node_ids, int64_t ids_capacity);
/* The same with the points already in device memory (n_points x 3 f64 on the factor's device; frame_ptr, Rp and params stay host
 * arrays): nothing but a few counters crosses PCIe.  The buffer must stay valid and unchanged until the call returns. */
int vxba_voxelize_push_device(vxba_factor* f, int64_t n_points, const double* d_xyz_local, const int64_t* frame_ptr, const double* Rp,
                              const vxba_voxelize_params* params, int64_t* n_pushed, uint64_t* node_ids, int64_t ids_capacity);

/* ---- inertial half of the LiDAR-inertial BA (host code; runs while the GPU sweeps) ------------------- */
/* Flat formats, every matrix column-major:
 *   state (VXBA_STATE_LEN f64) = the IMUST fields the BA touches (tools.hpp:135-199): [R 9 | p 3 | v 3 | bg 3 | ba 3 | g 3]
 *   imu   (VXBA_IMU_LEN f64)   = IMU_PRE (preintegration.hpp:11-30): [R_delta 9 | p_delta 3 | v_delta 3 | bg 3 | ba 3 |
 *       R_bg 9 | p_bg 9 | p_ba 9 | v_bg 9 | v_ba 9 | dtime | dbg 3 | dba 3 | dbg_buf 3 | dba_buf 3 | cov 15x15]
 * Tangent order per frame: [dphi dp dv dbg dba] (15), LiDAR block = the first 6. */
#define VXBA_STATE_LEN 24
#define VXBA_IMU_LEN 304
#define VXBA_LI_DIM 15

/* IMU_PRE::IMU_PRE(bg, ba) (preintegration.hpp:32-48); bg / ba may be NULL (zero). */
int vxba_imu_init(double* imu, const double* bg, const double* ba);
/* IMU_PRE::add_imu (preintegration.hpp:75-135): one mid-point, bias-corrected sample (what push_imu :50-73 feeds it).
 * noise_meas / noise_walk: the reference's 6x6 `noiseMeas` / `noiseWalk` (preintegration.hpp:9, voxelslam.cpp:828-833). */
int vxba_imu_add(double* imu, const double* gyr, const double* acc, double dt, const double* noise_meas, const double* noise_walk);
/* IMU_PRE::give_evaluate (preintegration.hpp:137-212): *residual = r^T cov^-1 r; with jac_enable also jtj (30x30) and gg (30). */
int vxba_imu_evaluate(const double* imu, const double* st1, const double* st2, int jac_enable, double* jtj, double* gg, double* residual);
/* IMU_PRE::give_evaluate_g (preintegration.hpp:214-294): as above with three more Jacobian columns for the gravity
 * vector -- jtj 33x33, gg 33. */
int vxba_imu_evaluate_g(const double* imu, const double* st1, const double* st2, int jac_enable, double* jtj, double* gg, double* residual);
/* IMU_PRE::update_state (preintegration.hpp:296-303). */
int vxba_imu_update_state(double* imu, const double* dxi15);
/* LI_BA_Optimizer::hess_plus (voxel_map.hpp:455-463): scatter-add the (6W) LiDAR system into the (15W) one. */
int vxba_hess_plus(int win_size, double* Hess15, double* JacT15, const double* Hess6, const double* JacT6);

/* LI_BA_Optimizer::divide_thread (voxel_map.hpp:465-523): Hess (15W)^2 col-major and JacT (15W) of the joint system
 * (imu_coef * IMU blocks + scattered LiDAR blocks), *residual = imu_coef/2 * sum r^T cov^-1 r + LiDAR residual.
 * states W*VXBA_STATE_LEN, imus (W-1)*VXBA_IMU_LEN.  The Hessian sweep runs on the GPU, the IMU blocks on the host
 * meanwhile. */
int vxba_li_evaluate(vxba_factor* f, const double* states, const double* imus, double imu_coef, double* Hess, double* JacT,
                     double* residual);
/* LI_BA_Optimizer::only_residual (voxel_map.hpp:525-560). */
int vxba_li_only_residual(vxba_factor* f, const double* states, const double* imus, double imu_coef, double* residual);
/* LI_BA_OptimizerGravity's members of the same names (voxel_map.hpp:663-773): hess_plus into a (15W+3)^2 system, divide_thread (Hess (15W+3)^2
 * column-major, JacT 15W+3: the three gravity unknowns at the tail, IMU factors through give_evaluate_g); only_residual is vxba_li_only_residual
 * (give_evaluate_g without Jacobian is give_evaluate without Jacobian, preintegration.hpp:214-294). */
int vxba_hess_plus_gravity(int win_size, double* Hess15g, double* JacT15g, const double* Hess6, const double* JacT6);
int vxba_li_evaluate_gravity(vxba_factor* f, const double* states, const double* imus, double imu_coef, double* Hess, double* JacT, double* residual);
/* LI_BA_Optimizer::damping_iter (voxel_map.hpp:562-653; three iterations upstream).  states and imus are in/out (the
 * factors' dbg / dba and their _buf copies move with every step and roll back on a rejected one, :608-609, 639-643).
 * hess_out (15W)^2 = `*hess`, exported before the gauge fix (:588).  trace_out max_iter*VXBA_TRACE_COLS, may be NULL. */
int vxba_li_damping_iter(vxba_factor* f, double* states, double* imus, double imu_coef, int max_iter, double* hess_out,
                         double* trace_out, int* n_trace);

/* LI_BA_OptimizerGravity::damping_iter (voxel_map.hpp:775-862; max_iter = 5 at its call site voxelslam.cpp:1644): the
 * gravity vector (states[21..23] of every frame, kept equal) joins the unknowns at the tail of the system; hess_out is
 * (15W+3)^2; only frame 0's pose is gauge-fixed.  resis_out[2] = residual before / after. */
int vxba_li_damping_iter_gravity(vxba_factor* f, double* states, double* imus, double imu_coef, int max_iter, double* hess_out,
                                 double* resis_out, double* trace_out, int* n_trace);

/* ---- the incremental local map, resident on the GPU -------------------------------------------------------
 * `surf_map` / `surf_map_slide` of OctoTree nodes with their sliding windows (voxel_map.hpp:896-1639), driven the way the
 * local-mapping thread drives them (voxelslam.cpp:1592-1700).  Each entry point replaces one call of that thread; the tree, the
 * window clusters, the stored scan points, the fix clusters and the plane records never leave the device, and tras_opt writes the
 * factor's planes in place -- a scan cycle moves the scan up and the poses down.  Parameters are the reference's globals
 * (voxel_map.hpp:83-89, voxelslam.cpp:795-812). */
typedef struct vxba_map vxba_map;
typedef struct vxba_lio vxba_lio;   /* the odometry handle, declared below */
typedef struct vxba_map_params {
  double voxel_size;                 /* voxel_size */
  int max_layer;                     /* max_layer (0..2) */
  double min_point[4];               /* min_point[layer]: a leaf is tested for planarity when N > min_point[layer] */
  double min_eigen_value;            /* plane_judge: lambda0 < min_eigen_value && lambda0 / lambda2 < plane_eigen_value_thre[layer] */
  double plane_eigen_value_thre[4];
  int max_points;                    /* fix cluster cap (voxel_map.hpp:86) */
  int win_size;                      /* <= 16 */
  int thread_num;                    /* only for upstream's `if(g_size < thd_num) return;` early returns (voxel_map.hpp:1603, voxelslam.cpp:1406, 1337) */
} vxba_map_params;
int vxba_map_create(const vxba_map_params* params, int device, vxba_map** out);
int vxba_map_destroy(vxba_map* m);
const char* vxba_map_last_error(const vxba_map* m);
/* cut_voxel_multi(surf_map, pvec, ord, surf_map_slide, win_size, pwld, sws) (voxel_map.hpp:1545-1639, call site voxelslam.cpp:1609):
 * scan `ord` of the window (= win_count - 1): n body-frame points (n x 3), their covariances as pvec_update left them (n x 9
 * column-major), their world coordinates (n x 3).  The _device variant takes device pointers (e.g. the arrays vxba_lio_pvec_update
 * keeps resident). */
int vxba_map_cut_voxel(vxba_map* m, int ord, int64_t n, const double* pnt_body, const double* var_world, const double* pwld);
int vxba_map_cut_voxel_device(vxba_map* m, int ord, int64_t n, const double* d_pnt_body, const double* d_var_world, const double* d_pwld);
/* The same on the scan resident in an odometry handle after vxba_lio_pvec_update (body points, world points and world covariances
 * are taken from the device: nothing crosses PCIe). */
int vxba_map_cut_voxel_lio(vxba_map* m, int ord, vxba_lio* lio);
/* multi_recut (voxelslam.cpp:1396-1453, OctoTree::recut voxel_map.hpp:1148-1194) followed by tras_opt (:1308-1333) straight into
 * `factor` (cleared by the caller, as voxhess.clear(); same win_size).  Rp: win_count poses.  Factor voxels are ordered by node id;
 * *n_pushed (optional) receives their number. */
int vxba_map_recut(vxba_map* m, int win_count, const double* Rp, vxba_factor* factor, int64_t* n_pushed);
/* multi_margi (voxelslam.cpp:1321-1394, OctoTree::margi voxel_map.hpp:1196-1305, mgsize = 1) with the optimised poses; reads the
 * cache the optimiser left in `factor` (pcr_adds / eig_values / eig_vectors) on the device. */
int vxba_map_margi(vxba_map* m, int win_count, const double* Rp, vxba_factor* factor);
/* Brings the odometry's plane map (vxba_lio, what `match` walks, voxel_map.hpp:1335-1392) up to date with the tree, on the device: the
 * plane records of every leaf -- and "no plane" for the empty octants of subdivided nodes -- under all roots that were in the slide map
 * since the last export.  Same voxel_size / max_layer / device.  Call after recut / margi, before the next scan is matched. */
int vxba_map_export_planes(vxba_map* m, vxba_lio* lio, int64_t* n_exported);
/* The ring of window slots moves on by mgsize (voxelslam.cpp:1683-1687). */
int vxba_map_slide(vxba_map* m, int mgsize);
/* out = [roots, roots in the slide map, leaves, mp[0]] */
int vxba_map_counts(vxba_map* m, int64_t out[4]);
/* The pool that holds the marginalised (fix) points of all leaves: out = [cursor, capacity, compactions so far], in points (96 bytes each).
 * The pool is a bump allocator -- a leaf that outgrows its region gets a new one, the old one is abandoned -- that is COMPACTED (live regions
 * moved to the front of a fresh pool, inside vxba_map_recut / vxba_map_margi) whenever the cursor passes max(4M points, 3 x what was live at
 * the last compaction), so a long mapping session holds a bounded multiple of its live points (the reference frees point_fix vectors instead). */
int vxba_map_fix_pool(vxba_map* m, int64_t out[3]);
/* The caller's journey odometer (`jour += spat`, voxelslam.cpp:1677).  The next vxba_map_margi stamps it on every root voxel of the slide
 * map, as multi_margi does (`iter->second->jour = jour`, voxelslam.cpp:1349). */
int vxba_map_set_journey(vxba_map* m, double jour);
/* The release branch of the local-mapping loop (voxelslam.cpp:1503-1523; OctoTree::tras_ptr voxel_map.hpp:1394-1405): every root voxel with
 * int(jour_now - root.jour) >= min_age (700 upstream) leaves the map with its whole subtree.  The node pool is compacted (survivors keep their
 * relative order, so id-ordered steps see the order they would have seen), child / root references and the voxel table are rebuilt, the
 * fix-point pool is compacted behind it, and the arrays are re-sized to what is left: device memory follows the map down.  Roots that are in
 * the slide map are kept whatever their stamp (upstream would delete them under the window's feet).  lio (may be NULL): the odometry handle
 * whose plane map mirrors this map -- the released roots are cleared there too.  Outputs may be NULL. */
int vxba_map_release(vxba_map* m, double jour_now, int min_age, vxba_lio* lio, int64_t* n_roots_released, int64_t* n_nodes_released);
/* Device memory held by the map, bytes: [0] node pool, [1] fix-point pool, [2] resident scans of the window, [3] voxel table + scratch, [4] total. */
int vxba_map_device_bytes(vxba_map* m, int64_t out[5]);
/* Every leaf (octo_state == 0), unordered.  ids: [x:16 | y:16 | z:16 | octant path:9 | 0:4 | layer:3] (as vxba_voxelize_push);
 * ints n x 8 = [layer, isexist, is_plane, has window, opt_state, last_num, stored fix points, root in slide map]; dbl n x (156 + 11 W) =
 * [pcr_add 10 | pcr_fix 10 | eig_value 3 | eig_vector 9 | plane centre 3 | normal 3 | radius | plane_var 36 | cov_add 81 | window
 * clusters W x 10 in window order | stored points per window slot W], matrices column-major.  *n_out = number of leaves (fills up to
 * `capacity`; pass NULL arrays to query the count). */
int vxba_map_leaves(vxba_map* m, int64_t capacity, uint64_t* ids, int32_t* ints, double* dbl, int64_t* n_out);

/* ---- execution options (per handle; none of them changes results beyond rounding) -----------------------------------------
 * Every switch that used to be an environment variable is a setter.  The environment variables of the same name
 * (VXBA_FUSED_SOLVE, VXBA_SPEC_COLLECTIVE, VXBA_WIDE_DEVICE_SOLVE, VXBA_K2_VPB, VXBA_LIO_DEVICE_EKF) still give the
 * INITIAL value of a new handle -- for A/B scripts -- and nothing reads them after vxba_create / vxba_lio_create. */
#define VXBA_OPT_FUSED_SOLVE 0        /* 1 (default): the 6W damped solve runs as workgroup 0 of the residual-sweep launch; 0: own launch */
#define VXBA_OPT_SPEC_COLLECTIVE 1    /* 1 (default): sharded LM loop with ONE all-reduce per iteration (speculative Hessian sweep at the trial poses) */
#define VXBA_OPT_WIDE_DEVICE_SOLVE 2  /* 1 (default): the damped step of a wide window (W > 10) by the library's blocked Cholesky on the device (host pivoted
                                         LDL^T when a pivot is not positive); 0: always the host LDL^T */
#define VXBA_OPT_LI_DEVICE_LOOP 3     /* reserved: the device-resident 15W loop of rounds 1-3 (IMU factor kernels + a one-wave block-Thomas solve, ~265 us per
                                         iteration against ~66 for the default shell) was removed in round 4; only 0 is accepted */
#define VXBA_OPT_K2_VOXELS_PER_BLOCK 4 /* 64 (default) or 32..63: voxels per residual-sweep workgroup (tuning experiment) */
#define VXBA_OPT_DEBUG_SOLVE_TIMEOUT 5 /* test hook, 0 (default): with 1 the voxel workgroups of a fused launch give up waiting for the in-launch solve at
                                         once, which exercises the transparent non-fused retry of vxba_damping_iter; with 2 vxba_li_damping_iter treats the
                                         first in-launch pose step of a call as undelivered, which exercises its host-solve fallback */
#define VXBA_OPT_LI_STRUCTURED_SOLVE 6 /* 1 (default): the host shells of LI_BA_Optimizer[Gravity] solve the damped 15W(+3) system by a band Cholesky of
                                         the velocity/bias part + Schur complement onto the poses (3x fewer flops); 0: dense pivoted LDL^T */
#define VXBA_OPT_LI_QUEUED_SWEEPS 7    /* 1 (default): the host shells of LI_BA_Optimizer[Gravity] queue an iteration's residual sweep (and the speculative Hessian sweep
                                       * behind it) BEFORE the host has finished the damped solve; the sweep's first workgroup waits for the trial poses in mapped host
                                       * memory while the others already hold their cluster rows.  0: every sweep is launched when its poses exist. */
#define VXBA_OPT_LI_DEVICE_POSE_SOLVE 8 /* 1 (default): LI_BA_Optimizer's host shell (queued sweeps, structured solve) eliminates velocities and biases on the host WHILE
                                       * the Hessian sweep runs and lets the device solve the reduced 6W-dimensional pose system inside the residual-sweep launch
                                       * (the LiDAR-only loop's four-wave solve): no kernel ever waits for the host, the LiDAR Hessian never crosses PCIe on the
                                       * critical path.  Steps after a rejection, the gravity variant and a non-positive band pivot take the host solve.  0: host solve. */
#define VXBA_OPT_FUSED_SWEEPS 9         /* 1 (default): inside one solve of the device-resident 6W loop (vxba_damping_iter, vxba_lm_steps; no collective attached)
                                       * the residual sweep at the trial poses and the Hessian sweep that linearises at the same poses one iteration later are
                                       * ONE launch behind the in-launch solve (Lidar_BA_Optimizer::damping_iter, voxel_map.hpp:386-439: only_residual, then
                                       * divide_thread of the next iteration), and the reduction behind it takes the accept / reject decision; a rejected step
                                       * then costs a Hessian sweep the reference does not run -- so a factor whose LAST call rejected more than a third of its steps runs its next
                                       * call as the three-launch iteration (break-even by the round-6 kernel times: 18 % rejected steps at 50k voxels, 32 % at 400k; results do not depend on the form).
                                       * 2: fused whatever the history -- and ALSO on a factor with a collective attached (one process per GPU only: the launch takes whole CUs and waits inside
                                       * itself, which two process ranks sharing one device do not survive; worth 7-9 % of a step from ~100k voxels per rank, nothing at 50k).  0: the three-launch iteration of rounds 1-5.  Needs
                                       * VXBA_OPT_FUSED_SOLVE = 1.  (The number belonged to a round-4 experiment that was removed in round 5.) */
#define VXBA_OPT_COUNT 10
#define VXBA_STAT_FUSED_FALLBACKS 100 /* read-only (vxba_get_option): times vxba_damping_iter re-ran a call with the solve as its own launch after the
                                         voxel workgroups of a fused launch had timed out waiting for it */
#define VXBA_STAT_LI_LAST_CALL_US 101 /* read-only (vxba_get_option): wall time of the last vxba_li_damping_iter[_gravity] call, microseconds,
                                         * measured inside the library (what a C caller sees; the Python mirror adds its array handling) */
#define VXBA_STAT_LI_DEVICE_FALLBACKS 102 /* read-only (vxba_get_option): times vxba_li_damping_iter found the in-launch pose step non-finite / undelivered when its
                                         * launch had ended, discarded that launch's residual sweep and took the host solve (the reference's pivoted LDL^T) instead */
#define VXBA_STAT_REJECT_HEAVY 103     /* read-only (vxba_get_option): 1 when the factor's last device-resident LM call rejected more than a third of its steps
                                       * (VXBA_OPT_FUSED_SWEEPS = 1 then runs the next call as the three-launch iteration) */
int vxba_set_option(vxba_factor* f, int option, int value);
int vxba_get_option(const vxba_factor* f, int option, int* value);

/* Arithmetic of the Hessian sweep (BASELINE.json configs[2], the mixed-precision tolerance study).  F64 (default): fp64
 * throughout, like the reference.  MIXED: the rank-3 rows of every voxel ("Jacobian") are rounded to f32 and multiplied on the
 * f32 matrix cores, summed in f32 inside one wave (<= 72 voxels) and accumulated in f64 across waves, workgroups and GPUs;
 * gradient, block-diagonal terms, residuals, eigen-decompositions and the solve stay fp64.  The gradient is exact, so the LM
 * fixed point is unchanged; only the step direction carries ~1e-7 relative error. */
#define VXBA_PRECISION_F64 0
#define VXBA_PRECISION_MIXED 1
/* MIXED, and the residual sweep reads the clusters from an f32 copy ("clusters also emitted as f32", configs[2]): per (voxel, frame)
 * [C sym6 | c | n] with c the cluster's mean and C its second moments ABOUT that mean -- raw body-frame moments do not survive f32, the
 * re-centred ones do (round-off ~1e-7 m^2 against a plane thickness of ~1e-2 m^2).  Half the bytes of that sweep's dominant stream.
 * The copy is kept beside the f64 rows (the Hessian sweep and every read-back use those) and refreshed when voxels were appended.
 * Eigenvalues / residuals move by ~1e-6 relative, and with them the LM fixed point by tens of nanometres (measured; the per-row errors average out). */
#define VXBA_PRECISION_MIXED_F32_CLUSTERS 2
int vxba_set_precision(vxba_factor* f, int mode);

/* ---- odometry: point-to-plane state estimation against the voxel plane map ---------------- */
/* Replaces `lio_state_estimation` (voxelslam.cpp:855-958): per EKF iteration every scan point is transformed with the
 * current pose, matched to the plane of the octree leaf it falls into (`match`, voxel_map.hpp:1335-1392, 1674-1698) and
 * the matched points' information is summed into HTH (6x6), HTz (6) and nnt (3x3).  The map walk (float-typed voxel index,
 * `wld > voxel_center` descent, float-typed distance tests, the per-point node cache `octos[i]` + `inside` :1471-1480) is
 * reproduced exactly, on a flattened map held in GPU memory.  The 15-dimensional EKF algebra between the sweeps runs on
 * the host.  No CPU fallback.
 *   state : VXBA_STATE_LEN f64 as above [R 9 col-major | p 3 | v 3 | bg 3 | ba 3 | g 3]      <- IMUST, tools.hpp:135-199
 *   cov   : 15x15 f64 col-major, tangent order [dphi dp dv dbg dba]                          <- IMUST::cov
 *   sweep : VXBA_LIO_SWEEP_LEN f64 = [HTH 36 col-major | HTz 6 | nnt 9 col-major | match_num] */
typedef struct vxba_lio vxba_lio; /* opaque: one plane map (`surf_map`) + the current scan (`pptr`) */
#define VXBA_LIO_SWEEP_LEN 52
#define VXBA_LIO_MAX_ITER 4 /* num_max_iter, voxelslam.cpp:859 */

/* voxel_size / max_layer: the reference's globals (voxel_map.hpp:86-88; max_layer <= 3 here). */
int vxba_lio_create(double voxel_size, int max_layer, int device, vxba_lio** out);
/* VXBA_LIO_OPT_DEVICE_EKF: 1 (default) = the 15x15 EKF update between sweeps runs as a kernel, whole state estimation enqueued
 * up front; 0 = the algebra on the host between sweeps. */
#define VXBA_LIO_OPT_DEVICE_EKF 0
int vxba_lio_set_option(vxba_lio* h, int option, int value);
int vxba_lio_destroy(vxba_lio* h);
const char* vxba_lio_last_error(const vxba_lio* h);

/* The plane map, flattened: one entry per octree leaf (OctoTree with octo_state == 0) that the walk can reach.
 *   loc n*3       root voxel, the VOXEL_LOC key of `surf_map` (tools.hpp:24-35); |index| < 2^20
 *   layer n       depth of the leaf, 0 = the root voxel itself
 *   path n        `leafnum` taken at each level on the way down (voxel_map.hpp:1371), 3 bits per level, first level lowest
 *   is_plane n    Plane::is_plane; 0 removes the leaf's plane (NULL = all ones)
 *   center, normal n*3, plane_var n*36 col-major, radius n  <- struct Plane (voxel_map.hpp:66-80; radius is a float there)
 * Upsert: a leaf seen before has its plane replaced.  When the host tree splits a leaf, remove it (is_plane = 0) in an
 * earlier call than the one that adds its children.  Plane records are reclaimed by vxba_lio_map_clear only. */
int vxba_lio_map_update(vxba_lio* h, int64_t n, const int64_t* loc, const int32_t* layer, const int32_t* path, const int32_t* is_plane,
                        const double* center, const double* normal, const double* plane_var, const double* radius);
int vxba_lio_map_clear(vxba_lio* h);
int vxba_lio_map_size(const vxba_lio* h, int64_t* n_roots, int64_t* n_planes);

/* The scan.  _raw = var_init (voxelslam.hpp:187-201): sensor-frame float points -> calcBodyVar (:164-185; dept_err / beam_err are
 * narrowed to float as there) -> extrinsic ext = [R 9 col-major | p 3] (NULL = identity).  _set takes ready pointVar arrays
 * (pnt n*3, var n*9 col-major; voxel_map.hpp:14-19).  _read returns what the sweeps use. */
int vxba_lio_scan_raw(vxba_lio* h, int64_t n, const float* xyz, const double* ext, double dept_err, double beam_err);
int vxba_lio_scan_set(vxba_lio* h, int64_t n, const double* pnt, const double* var);
int64_t vxba_lio_scan_size(const vxba_lio* h);
int vxba_lio_scan_read(vxba_lio* h, double* pnt, double* var);

/* One pass of the loop body voxelslam.cpp:873-919 under (state, cov).  reset_cache != 0 forgets the per-point node cache first
 * (the reference starts every lio_state_estimation call with an empty one).  plane_of_point (index of the matched leaf's plane
 * record = order of first insertion, -1 = no match) and sigma_of_point may both be NULL. */
int vxba_lio_sweep(vxba_lio* h, const double* state, const double* cov, int reset_cache, double* sweep, int32_t* plane_of_point,
                   double* sigma_of_point);
/* lio_state_estimation: state and cov in/out (x_curr).  info[4] = [ok (nnt's smallest eigenvalue >= 14, :951-957), iterations,
 * match_num of the last sweep, that eigenvalue]; sweeps_out VXBA_LIO_MAX_ITER * VXBA_LIO_SWEEP_LEN.  Both may be NULL. */
int vxba_lio_state_estimation(vxba_lio* h, double* state, double* cov, double* info, double* sweeps_out);
/* pvec_update (voxelslam.hpp:203-215): world points (n*3) and world covariances (n*9 col-major) of the scan under (state, cov).
 * Both also stay on the device for vxba_lio_leaf_stats until the scan is replaced; var may be NULL (not downloaded: the host tree
 * needs the world points to bucket them into leaves, the 72 bytes of covariance per point only feed cov_add). */
int vxba_lio_pvec_update(vxba_lio* h, const double* state, const double* cov, double* pwld, double* var);
/* What cut_voxel adds to the leaves this scan touches (OctoTree::push, voxel_map.hpp:969-993: pcr_add.push(pw), cov_add += Bf_var),
 * from the resident result of the last vxba_lio_pvec_update.  The host tree's bucketing comes in as indices: leaf c owns the scan
 * points order[cell_ptr[c] .. cell_ptr[c+1]) in push order.  Out: clusters n_cells x 10 (the PointCluster of the new points,
 * bit-identical to sequential push) and cov_add n_cells x 81 col-major -- increments the caller adds to its nodes.
 * VXBA_ERR_STATE without a preceding pvec_update of the current scan. */
int vxba_lio_leaf_stats(vxba_lio* h, int64_t n_cells, const int64_t* cell_ptr, const int32_t* order, double* clusters, double* cov_add);

/* The map-side producers of those plane records (f2's arithmetic; the host tree decides which leaves exist):
 * cov_add of OctoTree::push (voxel_map.hpp:990-992) = sum over a cell's points of Bf_var (:91-106), n_cells x 81 col-major 9x9, from
 * world points (n*3), their world covariances (n*9 col-major, e.g. vxba_lio_pvec_update's) and bucket offsets (n_cells + 1);
 * OctoTree::plane_update (:1118-1146) batched: world cluster (10), its eigen-decomposition (vxba_plane_fit) and cov_add ->
 * center / normal n*3, plane_var n*36 col-major, radius n (rounded to float as upstream). */
int vxba_cov_add_build(int device, int64_t n_cells, int64_t n_points, const double* xyz_world, const double* var, const int64_t* cell_ptr, double* cov_add);
int vxba_plane_update(int device, int64_t n, const double* clusters, const double* eig_val, const double* eig_vec, const double* cov_add, double* center,
                      double* normal, double* plane_var, double* radius);

/* down_sampling_voxel (tools.hpp:201-238): the voxel-grid filter upstream runs on every raw scan before the odometry
 * (voxelslam.cpp:1236, 1577-1583) and on every merged submap of the hierarchical BA (:2447).  One point per occupied voxel = the
 * running mean of its points in cloud order, in float and unfused like upstream (bit-identical means); voxel index = upstream's
 * float quotient / "-1 if negative" / truncation, |index| < 2^20.  xyz n*3 float, out_xyz capacity n*3; output ordered by ascending
 * (x, y, z) voxel index (upstream: unordered_map iteration order).  voxel_size < 0.001 copies the cloud through, like upstream. */
int vxba_down_sampling_voxel(int device, int64_t n, const float* xyz, double voxel_size, float* out_xyz, int64_t* n_out);

/* ---- hierarchical global BA: one bottom-up pass over a session of keyframes (BASELINE configs[4]) ----------------------------
 * thd_globalmapping's loop (voxelslam.cpp:2485-2595): windows of `wdsize` keyframes with stride `mgsize`, each refined by one round of
 * HBA_add_edge (:2320-2482 -- OctreeGBA::cut_voxel + OctreeGBA_multi_recut, Lidar_BA_Optimizer::damping_iter with 4 iterations, Hessian ->
 * pose-graph edge weights :2405-2427) and merged into a voxel-filtered submap anchored at its first keyframe (:2430-2450); then ONE
 * HBA_add_edge over all S submap poses (S <= VXBA_MAX_WIN_WIDE; vxba_hba_num_windows) with up to `top_max_iter` re-voxelisation
 * rounds.  The keyframe clouds live in device memory from vxba_hba_add_keyframes on; a pass moves only poses, Hessians and counts
 * across PCIe.  What consumes the edges (GTSAM's ISAM2 in the reference, :2231-2317) is outside this library. */
typedef struct vxba_hba vxba_hba;
int vxba_hba_create(int device, vxba_hba** out);
int vxba_hba_destroy(vxba_hba* h);
const char* vxba_hba_last_error(const vxba_hba* h);
/* Appends n_keyframes clouds (PointType xyz as float, keyframe coordinates): cloud_ptr n_keyframes + 1 offsets in points (cloud_ptr[0] = 0). */
int vxba_hba_add_keyframes(vxba_hba* h, int64_t n_keyframes, const int64_t* cloud_ptr, const float* xyz);
int vxba_hba_num_keyframes(const vxba_hba* h);
int vxba_hba_threads_used(const vxba_hba* h); /* host threads / streams the last vxba_hba_pass ran its bottom level on */
int vxba_hba_clear(vxba_hba* h); /* forget the keyframes (the device buffers are kept) */
/* Windows of a bottom-up pass over K keyframes (thd_globalmapping, voxelslam.cpp:2498-2575): S_full = (K - wdsize) / mgsize + 1 full windows of
 * wdsize keyframes at stride mgsize (0 when K < wdsize) and, with tail != 0, the CLOSING window of upstream's total_ba iteration (:2519-2523: it skips
 * the "fewer than wdsize keyframes" test) over the keyframes left behind the last pop, [S_full mgsize, K) -- wdsize - mgsize .. wdsize - 1 of them
 * in a long session, all K of a session shorter than one window.  A closing window of one keyframe is not refined (its cloud is the submap).
 * vxba_hba_window: first keyframe and keyframe count of window w. */
int vxba_hba_num_windows(int n_keyframes, int wdsize, int mgsize, int tail);
int vxba_hba_window(int n_keyframes, int wdsize, int mgsize, int tail, int w, int* first, int* count);
/* One whole pass on one device.  poses: K x 12 ([R column-major 9 | p 3], as everywhere).  n_threads: 1 .. 8 host threads / streams for the
 * bottom level; <= 0: the library picks -- 4 (measured best on one MI355X), fewer under a small cgroup CPU quota (the threads poll while their
 * streams run).  tail: see above (upstream: 1).
 * Out, S = vxba_hba_num_windows(..): submap_poses S x 12 (refined anchors), submap_sizes S (points per voxel-filtered submap; may be NULL); the edges
 * of both levels in order -- bottom level window by window, then the top level: edge_ij 2 ints (keyframe indices i < j), edge_data 18 doubles
 * [R_i^T R_j row-major 9 | R_i^T (p_j - p_i) 3 | v6 = 1 / |hess(6i+k, 6j+k)| 6] per edge, at most edge_capacity of them (the counts are exact even
 * when the arrays were too small or NULL: VXBA_ERR_ARG then); top_rounds 5 doubles per top-level round [factor voxels, residual before, after,
 * is_converge, fine parameters used] (top_max_iter x 5, may be NULL). */
int vxba_hba_pass(vxba_hba* h, const double* poses, const vxba_voxelize_params* coarse, const vxba_voxelize_params* fine, int wdsize, int mgsize, int tail,
                  int top_max_iter, int n_threads, double* submap_poses, int64_t* submap_sizes, int64_t edge_capacity, int32_t* edge_ij, double* edge_data,
                  int64_t* n_edges1, int64_t* n_edges2, double* top_rounds, int* n_top_rounds);
/* The same pass in its two halves, for one rank of N (one process per GPU; vxba_hba_pass is vxba_hba_bottom(0, 1) + vxba_hba_top on one device):
 *   vxba_hba_bottom          the windows w_first, w_first + w_stride, .. (rank r of N: r, N -- independent HBA_add_edge problems, no exchange): fills
 *                            THEIR rows of submap_poses / submap_sizes, leaves their submaps in device memory, returns their edges with the window
 *                            each came from (edge_window, may be NULL);
 *   vxba_hba_export_submaps  those submaps packed back to back (window order, float xyz) into the caller's DEVICE buffer -- what the rank hands to
 *                            the all-gather (ncclAllGather / torch.distributed over RCCL, device to device);
 *   vxba_hba_import_submaps  a peer's packed submaps (sizes: S entries, the selection's are read) into place;
 *   vxba_hba_top_factor      the factor of the top level (session-owned): attach the rank's collective to it (vxba_rccl_attach / vxba_peer_attach /
 *                            vxba_set_allreduce) when the top level is voxel-sharded;
 *   vxba_hba_top             ONE HBA_add_edge over all submap poses (voxelslam.cpp:2553-2560); coarse / fine may carry the rank's voxel shard
 *                            (shard_index / shard_count: every rank voxelises the same submaps and keeps the root voxels that hash to it; the packed
 *                            [Hess | JacT | residual] is summed by the attached collective, one all-reduce per sweep).  Needs every submap present. */
int vxba_hba_bottom(vxba_hba* h, const double* poses, const vxba_voxelize_params* coarse, const vxba_voxelize_params* fine, int wdsize, int mgsize, int tail,
                    int w_first, int w_stride, int n_threads, double* submap_poses, int64_t* submap_sizes, int64_t edge_capacity, int32_t* edge_ij,
                    double* edge_data, int32_t* edge_window, int64_t* n_edges);
int vxba_hba_export_submaps(vxba_hba* h, int w_first, int w_stride, float* d_out, int64_t capacity_points, int64_t* n_points);
int vxba_hba_import_submaps(vxba_hba* h, int w_first, int w_stride, const int64_t* sizes, const float* d_in);
int vxba_hba_top_factor(vxba_hba* h, vxba_factor** out);
int vxba_hba_top(vxba_hba* h, const double* poses, const vxba_voxelize_params* coarse, const vxba_voxelize_params* fine, int top_max_iter, double* submap_poses,
                 int64_t edge_capacity, int32_t* edge_ij, double* edge_data, int64_t* n_edges, double* top_rounds, int* n_top_rounds);

/* ---- measurement --------------------------------------------------------------------------------- */
/* The cluster-build kernel inside the voxeliser (vxba_voxelize_push*, vxba_hba_pass) -- the dominant kernel of a hierarchical-BA pass.
 * enable != 0: start a fresh measurement (every launch bracketed by events bound to its dispatch, one stream synchronisation per layer:
 * not for timed runs); enable == 0: stop and return the sums -- duration [ms], launches, algorithmic bytes (24 B per point + 8 B per
 * cell offset read, 80 B per cluster written). */
int vxba_voxelize_profile(int enable, double* ms_sum, long long* launches, double* algorithmic_bytes);
/* Bit mask of kernels to bracket with hipEvents on the launch stream: 1 = Hessian sweep (K3), 2 = residual sweep (K2),
 * 4 = K3 cross-block reduction, 8 = cluster build (K1), 16 = the all-reduce of a sharded factor, 32 = the fused solve + residual + Hessian
 * launch (VXBA_OPT_FUSED_SWEEPS); 0 = off. */
int vxba_set_profiling(vxba_factor* f, int mask);
/* Sum of kernel durations [ms] and launch counts since the last reset: index 0 = Hessian sweep (K3),
 * 1 = residual sweep (K2), 2 = K3 cross-block reduction, 3 = cluster build (K1). */
int vxba_get_kernel_times(vxba_factor* f, double ms_sum[4], int64_t calls[4], int reset);
/* Voxel-sharded factors (vxba_peer_*, vxba_rccl_*, vxba_set_allreduce): with bit 16 of the profiling mask set every all-reduce of the
 * packed [Hess | JacT | residual] buffer (the exchange step of the reference's thread fan-in, voxel_map.hpp:323-332, across GPUs) is
 * bracketed with hipEvents on the factor's stream; sum [ms] and count since the last reset.  The interval includes the wait for the
 * slowest peer. */
int vxba_get_collective_time(vxba_factor* f, double* ms_sum, int64_t* calls, int reset);
/* The fused launch of VXBA_OPT_FUSED_SWEEPS (in-launch solve, then evaluate_only_residual and acc_evaluate2 at the trial poses, voxel_map.hpp:132-279)
 * bracketed while bit 32 of the profiling mask was set: sum of launch durations [ms] and count since the last reset. */
int vxba_get_fused_time(vxba_factor* f, double* ms_sum, int64_t* calls, int reset);
/* Algorithmic bytes of one full sweep over the current factor: 0 = K3, 1 = K2. */
int vxba_algorithmic_bytes(const vxba_factor* f, double bytes[2]);
int vxba_nnz(vxba_factor* f, int64_t* nnz);
/* Device memory the factor holds right now, in bytes: [0] the factor itself (cluster storage -- frame-major planes, or the compressed
 * rows of a window wider than VXBA_MAX_WIN -- per-voxel planes, batch-major copy, the f32 cluster copy of VXBA_PRECISION_MIXED_F32_CLUSTERS, cache snapshot), [1] sweep work space (block
 * partials, packed system; for wide windows the pair index, the row buffer and the dense solver), [2] grow-only staging / scratch
 * left behind by the push calls, [3] the sum. */
int vxba_device_bytes(const vxba_factor* f, int64_t bytes[4]);

/* Debug: D(16x16, row-major) = A(16x4) B(4x16) through one v_mfma_f64_16x16x4_f64 with the lane maps the Hessian
 * kernel relies on (unit-tested on the GPU so a wrong operand layout is caught in isolation). */
int vxba_debug_mfma_probe(int device, const double* A16x4, const double* B4x16, double* D16x16);
/* test access to the structured solve of the LiDAR-inertial system (host code, no GPU needed): A is m x m (full symmetric storage),
 * m = lead_y + 15 nframes + tail_x, laid out [lead velocity/bias unknowns | nframes x (rot 3, pos 3, v 3, bg 3, ba 3) | tail]; solves A x = b.
 * Returns VXBA_ERR_STATE when the banded part is not positive definite (the library then falls back to the dense pivoted LDL^T). */
int vxba_debug_band_schur(int m, const double* A, const double* b, int nframes, int lead_y, int tail_x, double* x);

/* Debug: per-wave s_memtime stamps written by the instrumented kernel instantiations (env VXBA_DBG=1 /
 * VXBA_K3_SGB=5); 8 slots per wave. */
int vxba_debug_stamps(int clear, unsigned long long* out, size_t n);
int vxba_debug_partials(vxba_factor* f, double* out, size_t n);   /* development: workgroup partials of the last Hessian sweep */

#ifdef __cplusplus
}
#endif
#endif /* VXBA_H */
