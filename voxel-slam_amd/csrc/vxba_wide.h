// Launch interface of the wide-window sweeps (vxba_wide.hip): win_size up to WIDE_MAXW, sparse incidence.
#pragma once
#include <hip/hip_runtime.h>

#include "vxba_kernels.h"

namespace vxw {

constexpr int WIDE_MAXW = 128;

// Cluster store of a wide factor: compressed rows (loop_refine.hpp:358-405 -- a voxel of the top-level window is seen from
// a handful of ~100 submap poses).  Entries (voxel, frame) voxel-major, frames ascending inside a voxel; an entry's cluster is ten
// consecutive doubles (ecl[10 e ..]: the Hessian sweep GATHERS entries, so a cluster should be one or two cache lines, not ten):
// nnz x 80 bytes instead of V x W x 80 (0.8 GB at V = 100k, W = 99; 40 MB at five observers per voxel).  The per-voxel planes (fix,
// coe, cache) stay as they are.
struct WideStore {
  double* ecl = nullptr;        // [ES][10]
  long long* eptr = nullptr;    // [vcap + 1]; eptr[V] == nnz
  int* eframe = nullptr;        // [ES]
  int* evoxel = nullptr;        // [ES]
  long long ES = 0, nnz = 0;
  int vcap = 0;
  char* tmp = nullptr;          // grow-only scratch of the appends
  size_t tmp_cap = 0;
};
// what the wide kernels see: the factor's per-voxel planes (fv.cl is unused) + the entry store
struct WideView {
  vxk::FactorView fv;
  const double* ecl;
  const long long* eptr;
  const int* eframe;
  const int* evoxel;
  long long ES, nnz;
};
inline WideView wide_view(const vxk::FactorView& fv, const WideStore& st) { return WideView{fv, st.ecl, st.eptr, st.eframe, st.evoxel, st.ES, st.nnz}; }
void store_free(WideStore& st);
// room for vcap voxels and ecap entries (grows geometrically, keeps eptr[0..V] and the nnz entries); 0 or -1 with *err set
int store_reserve(WideStore& st, int vcap, long long ecap, int V, hipStream_t s, const char** err);
// Append n voxels given as compressed rows on the device: d_ptr[n + 1] (d_ptr[0] == 0), d_fr[nnz_new] frames, d_cl[nnz_new][10].  The
// caller has reserved the room and knows nnz_new.  *d_bad != 0 afterwards: a frame index was not strictly increasing inside a voxel
// or not below W (the store is then unchanged as far as st.nnz / V are concerned -- the caller simply does not advance them).
void store_append_csr(WideStore& st, int v0, int n, const long long* d_ptr, const int* d_fr, const double* d_cl, long long nnz_new, int W, int* d_bad,
                      hipStream_t s);
// Append n voxels given densely, d_dense[n][W][10] on the device with N == 0 marking an unobserved frame (the format of
// vxba_push_voxels and of the voxelisation's staging; frame_major: d_dense[W][n][10], the cell order of vxba_push_points): counts, scan,
// reserve, fill.  Returns the number of entries appended or -1.
long long store_append_dense(WideStore& st, int v0, int n, const double* d_dense, int W, int V, hipStream_t s, const char** err, int frame_major = 0);
// d_dense[n][W][10] <- voxels [head, head + n) (zeros where unobserved)
void store_expand(const WideStore& st, int head, int n, int W, double* d_dense, hipStream_t s);
// number of entries with N != 0 among voxels [0, V)
long long store_count_observed(const WideStore& st, int V, hipStream_t s, const char** err);

// Residual sweep (LidarFactor::evaluate_only_residual, voxel_map.hpp:243-279) for any W <= WIDE_MAXW: one lane per voxel walking its
// entries.  d_poses: W*12 f64 on the device.
// Wave partials of sum coe*lambda0 into d_partial[0 .. ceil((end-head)/64)); returns their number.
int launch_k2_wide(const WideView& wv, const double* d_poses, int head, int end, double* d_partial, hipStream_t s);

// Incidence structure of a wide factor (depends on the clusters only, not on the poses): for every 6x6 block of the Hessian, the run
// of entry pairs (entries = the store's) that contribute to it.  Device arrays owned by the index.
constexpr long long WIDE_TASK_RECORDS = 512;   // pair records per wave of the Hessian sweep
constexpr int WIDE_VREC = 18;                   // doubles of a voxel's packed record for the Hessian sweep
constexpr int WIDE_TASK_OUT = 42;               // a task's partial: 6x6 block + 6 gradient components
struct WideTask { long long lo, hi; int key, pad; };
struct WideIndex {
  int V = 0, nkeys = 0, ntasks = 0;
  long long nnz = 0, np = 0;
  unsigned int* sei = nullptr;           // [np] first entry of a pair, sorted by block key (frame_i * W + frame_j), voxel order inside
  unsigned int* sej = nullptr;           // [np] second entry
  unsigned int* key_list = nullptr;      // [nkeys] block keys present
  long long* key_ptr = nullptr;          // [nkeys + 1] runs of sei / sej
  WideTask* tasks = nullptr;             // [ntasks] pieces of the runs, key by key
  int* key_task_ptr = nullptr;           // [nkeys + 1] tasks of a key
  double* task_partial = nullptr;        // [ntasks][WIDE_TASK_OUT] work space of the sweep
  double* vrec = nullptr;                // [V][WIDE_VREC] packed per-voxel records, refreshed by every sweep
  struct WidePool* pool = nullptr;       // the index's own device blocks, kept between builds (vxba_wide.hip: a rebuild allocates ~22 arrays and released as many;
                                         // with hipMalloc / hipFree at 0.3-0.6 ms apiece that was half of a hierarchical pass's top level)
};
int build_index(const WideView& wv, int V, WideIndex& wi, hipStream_t s, const char** err);
size_t index_bytes(const WideIndex& wi, int W);
size_t store_bytes(const WideStore& st);   // 0, or -1 with *err set
void free_index(WideIndex& wi);      // the arrays go back to the index's pool (a rebuild follows)
void destroy_index(WideIndex& wi);   // ... and the pool itself is released (the factor is destroyed / changes its window size)

// Hessian sweep (LidarFactor::acc_evaluate2, voxel_map.hpp:132-241) for any W <= WIDE_MAXW, pair-major: (A) one lane per observed
// (voxel, frame) entry computes its rank-3 rows and gradient / block-diagonal terms, (B) one wave per 6x6 Hessian block sums its
// run of entry pairs in registers and a fixed butterfly, and writes the block and its mirror image.  No atomics: bitwise
// reproducible.  d_partial: scratch for the residual's wave partials (>= ceil((end - head) / 64) doubles).
void launch_k3_wide(const WideView& wv, const double* d_poses, const WideIndex& wi, int head, int end, double* d_packed, double* d_partial,
                    hipStream_t s);

// Device-side damped step of the wide LM shell ((H + u D) dxi = -JacT after the gauge fix; dense Cholesky from hipSOLVER, loaded on
// first use).  create returns nullptr if the library cannot be used; step returns non-zero if this step must be taken on the host.
struct DenseSolver;
DenseSolver* wide_solver_create(int n, hipStream_t s);
void wide_solver_free(DenseSolver*& ds);
size_t wide_solver_bytes(const DenseSolver* ds);
// debug_give_up (test hook): the device-wide barriers stop waiting at once -- the call must then report failure (host fallback), not hang
int wide_solver_step(DenseSolver* ds, const double* d_packed, double u, hipStream_t s, double* dxi, double* q1, double* residual1, bool debug_give_up = false);

}  // namespace vxw
