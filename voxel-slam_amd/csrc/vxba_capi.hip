// C-ABI implementation (include/vxba.h): device memory, streams, launch sequencing and the host part of
// the LM shell.  No CPU fallback anywhere: without a gfx950 device every entry point fails loudly.
#include "vxba_wait.hpp"
#include "vxba_capi_internal.hpp"

using namespace vxc;

extern "C" {

int vxba_create(int win_size, int device, vxba_factor** out) {
  if (!out) return VXBA_ERR_ARG;
  *out = nullptr;
  if (win_size < 1 || win_size > VXBA_MAX_WIN_WIDE) return VXBA_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return VXBA_ERR_NODEV;
  if (device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return VXBA_ERR_NODEV;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VXBA_ERR_NODEV;  // kernels are built for gfx950 only
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_HIP;
  vxba_factor* f = new vxba_factor();
  f->W = win_size;
  f->device = device;
  f->cus = prop.multiProcessorCount;
  options_from_env(f);
  auto bail = [&](hipError_t) { vxba_destroy(f); return VXBA_ERR_HIP; };
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&f->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e);
  f->stream = f->own_stream;
  if (ensure_exchange(f) != VXBA_OK) return bail(hipErrorOutOfMemory);
  if ((e = hipMalloc((void**)&f->d_count, sizeof(unsigned long long))) != hipSuccess) return bail(e);
  if ((e = hipMalloc((void**)&f->d_poses, sizeof(double) * 12 * VXBA_MAX_WIN_WIDE)) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc((void**)&f->h_poses, sizeof(double) * 8 * 12 * VXBA_MAX_WIN_WIDE, hipHostMallocDefault)) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc((void**)&f->h_scalar, 2 * sizeof(double), hipHostMallocDefault)) != hipSuccess) return bail(e);
  if ((e = hipMalloc((void**)&f->d_lm, sizeof(vxk::LMState))) != hipSuccess) return bail(e);
  if ((e = hipMemset(f->d_lm, 0, sizeof(vxk::LMState))) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc((void**)&f->h_lm, sizeof(vxk::LMState), hipHostMallocDefault)) != hipSuccess) return bail(e);
  *out = f;
  return VXBA_OK;
}

int vxba_destroy(vxba_factor* f) {
  if (!f) return VXBA_OK;
  hipSetDevice(f->device);
  if (f->stream) hipStreamSynchronize(f->stream);
  vxba_rccl_detach(f);
  vxba_peer_detach(f);
  if (f->peer.box) hipFree(f->peer.box);
  for (auto& ep : f->pending) { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
  for (auto e : f->free_events) hipEventDestroy(e);
  hipFree(f->planes); hipFree(f->clb); hipFree(f->cl32); hipFree(f->snapshot); hipFree(f->staging); hipFree(f->d_partial3); hipFree(f->d_partial2); if (f->h_partial2) hipHostFree(f->h_partial2);
  if (f->h_feed) (void)hipHostFree(f->h_feed);
  if (f->h_lirec) (void)hipHostFree(f->h_lirec);
  if (f->lirec_vram) (void)hipFree(f->lirec_vram);
  if (f->h_packed2) (void)hipHostFree(f->h_packed2);
  if (f->h_liout) (void)hipHostFree(f->h_liout);
  if (f->li_ev2) (void)hipEventDestroy(f->li_ev2);
  if (f->li_ev3) (void)hipEventDestroy(f->li_ev3);
  vxw::destroy_index(f->wide);
  vxw::store_free(f->wstore);
  vxw::wide_solver_free(f->wide_solver);
  hipFree(f->own_packed); hipFree(f->d_count); hipFree(f->d_poses);
  if (f->h_poses) hipHostFree(f->h_poses);
  for (auto& ev : f->pose_ev) if (ev) hipEventDestroy(ev);
  if (f->li_ev) hipEventDestroy(f->li_ev);
  if (f->h_packed) hipHostFree(f->h_packed);
  if (f->h_scalar) hipHostFree(f->h_scalar);
  hipFree(f->d_lm); hipFree(f->d_scratch);
  if (f->h_lm) hipHostFree(f->h_lm);
  if (f->own_stream) hipStreamDestroy(f->own_stream);
  delete f;
  return VXBA_OK;
}

int vxba_clear(vxba_factor* f) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  f->V = 0;
  f->cl32_built = 0;
  f->wstore.nnz = 0;
  f->wide_dirty = true;
  f->snapshot_v = 0;
  return VXBA_OK;
}

int vxba_set_win_size(vxba_factor* f, int win_size) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  if (win_size < 1 || win_size > VXBA_MAX_WIN_WIDE) return fail(f, VXBA_ERR_UNSUPPORTED, "win_size outside [1, VXBA_MAX_WIN_WIDE]");
  if (win_size == f->W) return VXBA_OK;
  if (f->V != 0) return fail(f, VXBA_ERR_STATE, "win_size can only change on an empty factor");
  hipSetDevice(f->device);
  if (f->planes) { VX_HIP(f, hipStreamSynchronize(f->stream)); VX_HIP(f, hipFree(f->planes)); if (f->clb) VX_HIP(f, hipFree(f->clb)); f->planes = nullptr; f->clb = nullptr; }
  if (f->cl32) { VX_HIP(f, hipFree(f->cl32)); f->cl32 = nullptr; f->cl32_vs = 0; f->cl32_built = 0; }
  f->VS = 0;
  vxw::store_free(f->wstore);
  vxw::destroy_index(f->wide);
  f->W = win_size;
  return ensure_exchange(f);
}

int vxba_win_size(const vxba_factor* f) { return f ? f->W : 0; }
int vxba_size(const vxba_factor* f) { return f ? f->V : 0; }
size_t vxba_packed_len(const vxba_factor* f) { return f ? (size_t)36 * f->W * f->W + 6 * f->W + 1 : 0; }
const char* vxba_last_error(const vxba_factor* f) { return f ? f->err.c_str() : "null factor"; }

int vxba_set_stream(vxba_factor* f, void* hip_stream) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  VX_HIP(f, hipStreamSynchronize(f->stream));
  f->stream = hip_stream ? (hipStream_t)hip_stream : f->own_stream;
  return VXBA_OK;
}

int vxba_reserve(vxba_factor* f, int n_voxels) {
  VX_LOCK(f);
  if (!f || n_voxels < 0) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  return ensure_capacity(f, n_voxels);
}

namespace {
// Wide factors are filled by a few big pushes (a top-level window of the hierarchical BA), not by a stream of small ones: the
// grow-only staging / scratch buffers of the push calls are handed back afterwards when they are large (they would otherwise
// outweigh the compressed-row store itself).
void release_push_buffers(vxba_factor* f) {
  if (!is_wide(f)) return;
  const size_t keep = (size_t)16 << 20;
  if (f->staging && f->staging_len * sizeof(double) > keep) { (void)hipStreamSynchronize(f->stream); (void)hipFree(f->staging); f->staging = nullptr; f->staging_len = 0; }
  if (f->d_scratch && f->scratch_cap > keep) { (void)hipStreamSynchronize(f->stream); (void)hipFree(f->d_scratch); f->d_scratch = nullptr; f->scratch_cap = 0; }
}
// wide factors: n voxels whose clusters sit densely on the device (d_dense[n][W][10], N == 0 = unobserved) -> appended to the store
int wide_append_dense(vxba_factor* f, int n, const double* d_dense, int frame_major = 0) {
  const char* emsg = nullptr;
  const long long added = vxw::store_append_dense(f->wstore, f->V, n, d_dense, f->W, f->V, f->stream, &emsg, frame_major);
  if (added < 0) return fail(f, VXBA_ERR_HIP, emsg ? emsg : "wide store: append failed");
  f->wstore.nnz += added;
  return VXBA_OK;
}
}  // namespace

int vxba_push_voxels(vxba_factor* f, int n, const double* clusters, const double* fix, const double* coe, const double* eig_val,
                     const double* eig_vec, const double* merged) {
  VX_LOCK(f);
  if (!f || n < 0 || (n > 0 && (!clusters || !fix || !coe))) return fail(f, VXBA_ERR_ARG, "push_voxels: null input");
  if (n == 0) return VXBA_OK;
  for (int a = 0; a < n; a++)
    if (!(coe[a] >= 0.0)) return fail(f, VXBA_ERR_ARG, "push_voxels: coe must be >= 0");
  hipSetDevice(f->device);
  int rc = ensure_capacity(f, f->V + n);
  if (rc) return rc;
  const size_t ncl = (size_t)n * f->W * 10;
  rc = ensure_staging(f, ncl);
  if (rc) return rc;
  VX_HIP(f, hipMemcpyAsync(f->staging, clusters, ncl * sizeof(double), hipMemcpyHostToDevice, f->stream));
  const long long nnz0 = f->wstore.nnz;
  if (is_wide(f)) { rc = wide_append_dense(f, n, f->staging); if (rc) return rc; }
  else {
    vxk::launch_scatter_clusters(f->staging, view(f), f->V, n, f->stream);
    vxk::launch_build_clb(view(f), f->V, n, f->stream);
    clusters_written(f, f->V);
  }
  VX_HIP(f, hipStreamSynchronize(f->stream));
  rc = append_meta(f, f->V, n, fix, coe, eig_val, eig_vec, merged);
  if (rc) { f->wstore.nnz = nnz0; return rc; }
  f->V += n;
  f->wide_dirty = true;
  release_push_buffers(f);
  return VXBA_OK;
}

namespace {
// the factor's grow-only device scratch (staging of host arrays on their way into the planes): no hipMalloc / hipFree per call
int ensure_scratch(vxba_factor* f, size_t need) {
  if (need > f->scratch_cap) {
    VX_HIP(f, hipStreamSynchronize(f->stream));
    if (f->d_scratch) VX_HIP(f, hipFree(f->d_scratch));
    f->d_scratch = nullptr; f->scratch_cap = 0;
    VX_HIP(f, hipMalloc((void**)&f->d_scratch, need + need / 4));
    f->scratch_cap = need + need / 4;
  }
  return VXBA_OK;
}
}  // namespace

// LidarFactor::push_voxel for sparse incidence (the top level of the hierarchical BA pushes voxels seen from a handful of ~100
// submap poses, loop_refine.hpp:358-405): the caller lists only the observed (voxel, frame) entries.  The planes are filled on the
// device, so neither a dense n x W x 10 host array (0.8 GB at n = 100k, W = 99) nor its PCIe transfer exists -- nnz x 88 bytes go up.
int vxba_push_voxels_csr(vxba_factor* f, int n, const int64_t* row_ptr, const int32_t* frame_idx, const double* clusters, const double* fix, const double* coe,
                         const double* eig_val, const double* eig_vec, const double* merged) {
  VX_LOCK(f);
  if (!f || n < 0 || (n > 0 && (!row_ptr || !fix || !coe))) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: null input");
  if (n == 0) return VXBA_OK;
  if (row_ptr[0] != 0) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: row_ptr[0] must be 0");
  const int64_t nnz = row_ptr[n];
  for (int a = 0; a < n; a++) {
    if (row_ptr[a + 1] < row_ptr[a] || row_ptr[a + 1] - row_ptr[a] > f->W) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: row_ptr must be non-decreasing with at most win_size entries per voxel");
    if (!(coe[a] >= 0.0)) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: coe must be >= 0");
  }
  if (nnz > 0 && (!frame_idx || !clusters)) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: null entries");
  hipSetDevice(f->device);
  int rc = ensure_capacity(f, f->V + n);
  if (rc) return rc;
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t b_ptr = up((size_t)(n + 1) * sizeof(int64_t)), b_fr = up((size_t)std::max<int64_t>(1, nnz) * sizeof(int32_t)), b_cl = up((size_t)std::max<int64_t>(1, nnz) * 10 * sizeof(double));
  rc = ensure_scratch(f, b_ptr + b_fr + b_cl + 256);
  if (rc) return rc;
  char* q = f->d_scratch;
  long long* d_ptr = (long long*)q; q += b_ptr;
  int* d_fr = (int*)q; q += b_fr;
  double* d_cl = (double*)q; q += b_cl;
  int* d_bad = (int*)q;
  VX_HIP(f, hipMemsetAsync(d_bad, 0, sizeof(int), f->stream));
  VX_HIP(f, hipMemcpyAsync(d_ptr, row_ptr, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->stream));
  if (nnz > 0) {
    VX_HIP(f, hipMemcpyAsync(d_fr, frame_idx, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, f->stream));
    VX_HIP(f, hipMemcpyAsync(d_cl, clusters, (size_t)nnz * 10 * sizeof(double), hipMemcpyHostToDevice, f->stream));
  }
  if (is_wide(f)) {   // straight into the compressed-row store: nothing dense exists anywhere
    const char* emsg = nullptr;
    if (vxw::store_reserve(f->wstore, f->V + n, f->wstore.nnz + nnz, f->V, f->stream, &emsg) != 0) return fail(f, VXBA_ERR_HIP, emsg ? emsg : "wide store: allocation failed");
    vxw::store_append_csr(f->wstore, f->V, n, d_ptr, d_fr, d_cl, nnz, f->W, d_bad, f->stream);
  } else {
    vxk::launch_scatter_clusters_csr(d_ptr, d_fr, d_cl, view(f), f->V, n, d_bad, f->stream);
    vxk::launch_build_clb(view(f), f->V, n, f->stream);
    clusters_written(f, f->V);
  }
  int bad = 0;
  VX_HIP(f, hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  if (bad) return fail(f, VXBA_ERR_ARG, "push_voxels_csr: frame indices must be strictly increasing inside a voxel and below win_size (nothing was appended)");
  rc = append_meta(f, f->V, n, fix, coe, eig_val, eig_vec, merged);
  if (rc) return rc;
  if (is_wide(f)) f->wstore.nnz += nnz;
  f->V += n;
  f->wide_dirty = true;
  release_push_buffers(f);
  return VXBA_OK;
}

int vxba_push_points(vxba_factor* f, int n_voxels, int64_t n_points, const double* xyz_body, const int64_t* cell_ptr, const double* fix,
                     const double* coe) {
  VX_LOCK(f);
  if (!f || n_voxels < 0 || n_points < 0 || !cell_ptr || (n_points > 0 && !xyz_body)) return fail(f, VXBA_ERR_ARG, "push_points: null input");
  if (n_voxels == 0) return VXBA_OK;
  const int64_t ncells = (int64_t)n_voxels * f->W;
  if (cell_ptr[0] != 0 || cell_ptr[ncells] != n_points) return fail(f, VXBA_ERR_ARG, "push_points: cell_ptr must span [0, n_points]");
  if (coe)
    for (int a = 0; a < n_voxels; a++)
      if (!(coe[a] >= 0.0)) return fail(f, VXBA_ERR_ARG, "push_points: coe must be >= 0");
  hipSetDevice(f->device);
  int rc = ensure_capacity(f, f->V + n_voxels);
  if (rc) return rc;
  const size_t b_xyz = ((std::max<size_t>(1, (size_t)n_points * 3) * sizeof(double)) + 255) / 256 * 256;
  rc = ensure_scratch(f, b_xyz + (size_t)(ncells + 1) * sizeof(int64_t));
  if (rc) return rc;
  double* d_xyz = (double*)f->d_scratch;
  int64_t* d_ptr = (int64_t*)(f->d_scratch + b_xyz);
  hipError_t e = hipSuccess;
  auto cleanup = [&]() {};
  if (n_points) e = hipMemcpyAsync(d_xyz, xyz_body, (size_t)n_points * 3 * sizeof(double), hipMemcpyHostToDevice, f->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_ptr, cell_ptr, (size_t)(ncells + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->stream);
  if (e != hipSuccess) { cleanup(); f->err = std::string("push_points H2D: ") + hipGetErrorString(e); return VXBA_ERR_HIP; }
  const long long nnz0 = f->wstore.nnz;
  if (is_wide(f)) {
    // cell sums (the caller's cell table is dense, frame-major: cell = frame * n_voxels + voxel) into a staging block of the same shape,
    // from there into the compressed-row store
    rc = ensure_staging(f, (size_t)n_voxels * f->W * 10);
    if (rc) return rc;
    {
      ScopedKernelTimer t(f, 3);
      vxk::launch_k1_build_aos(d_xyz, d_ptr, ncells, f->staging, f->stream);
    }
    rc = wide_append_dense(f, n_voxels, f->staging, 1);
    if (rc) { f->wstore.nnz = nnz0; return rc; }
  } else {
    {
      ScopedKernelTimer t(f, 3);
      vxk::launch_k1_build(d_xyz, d_ptr, n_voxels, f->W, view(f), f->V, f->stream);
    }
    vxk::launch_build_clb(view(f), f->V, n_voxels, f->stream);
    clusters_written(f, f->V);
  }
  e = hipStreamSynchronize(f->stream);
  if (e == hipSuccess) e = hipGetLastError();
  cleanup();
  if (e != hipSuccess) { f->wstore.nnz = nnz0; f->err = std::string("K1: ") + hipGetErrorString(e); return VXBA_ERR_HIP; }
  rc = append_meta(f, f->V, n_voxels, fix, coe, nullptr, nullptr, nullptr);
  if (rc) { f->wstore.nnz = nnz0; return rc; }
  f->V += n_voxels;
  f->wide_dirty = true;
  release_push_buffers(f);
  return VXBA_OK;
}

int vxba_read_clusters(vxba_factor* f, int head, int end, double* clusters) {
  VX_LOCK(f);
  if (!f || !clusters) return VXBA_ERR_ARG;
  int rc = check_range(f, head, end);
  if (rc) return rc;
  const int n = end - head;
  if (n == 0) return VXBA_OK;
  hipSetDevice(f->device);
  const size_t len = (size_t)n * f->W * 10;
  rc = ensure_staging(f, len);
  if (rc) return rc;
  if (is_wide(f)) vxw::store_expand(f->wstore, head, n, f->W, f->staging, f->stream);
  else vxk::launch_gather_clusters(view(f), head, n, f->staging, f->stream);
  VX_HIP(f, hipMemcpyAsync(clusters, f->staging, len * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  return VXBA_OK;
}

int vxba_acc_evaluate2(vxba_factor* f, const double* Rp, int head, int end, double* Hess, double* JacT, double* residual) {
  VX_LOCK(f);
  if (!f || !Rp || !Hess || !JacT || !residual) return fail(f, VXBA_ERR_ARG, "acc_evaluate2: null argument");
  int rc = check_range(f, head, end);
  if (rc) return rc;
  hipSetDevice(f->device);
  rc = sweep_hess_host(f, Rp, head, end);
  if (rc) return rc;
  const int n = 6 * f->W;
  std::memcpy(Hess, f->h_packed, sizeof(double) * n * n);
  std::memcpy(JacT, f->h_packed + (size_t)n * n, sizeof(double) * n);
  *residual = f->h_packed[(size_t)n * n + n];
  return VXBA_OK;
}

int vxba_evaluate_only_residual(vxba_factor* f, const double* Rp, int head, int end, double* residual) {
  VX_LOCK(f);
  if (!f || !Rp || !residual) return fail(f, VXBA_ERR_ARG, "evaluate_only_residual: null argument");
  int rc = check_range(f, head, end);
  if (rc) return rc;
  hipSetDevice(f->device);
  return sweep_residual_host(f, Rp, head, end, residual);
}

int vxba_acc_evaluate2_device(vxba_factor* f, const double* Rp, int head, int end, double* d_out) {
  VX_LOCK(f);
  if (!f || !Rp || !d_out) return fail(f, VXBA_ERR_ARG, "acc_evaluate2_device: null argument");
  int rc = check_range(f, head, end);
  if (rc) return rc;
  hipSetDevice(f->device);
  return sweep_hess_device(f, Rp, nullptr, nullptr, nullptr, head, end, d_out);
}

int vxba_evaluate_only_residual_device(vxba_factor* f, const double* Rp, int head, int end, double* d_out) {
  VX_LOCK(f);
  if (!f || !Rp || !d_out) return fail(f, VXBA_ERR_ARG, "evaluate_only_residual_device: null argument");
  int rc = check_range(f, head, end);
  if (rc) return rc;
  hipSetDevice(f->device);
  return sweep_residual_device(f, Rp, nullptr, 0, head, end, d_out);
}

int vxba_read_cache(vxba_factor* f, int head, int end, double* eig_val, double* eig_vec, double* merged) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  int rc = check_range(f, head, end);
  if (rc) return rc;
  const int n = end - head;
  if (n == 0) return VXBA_OK;
  hipSetDevice(f->device);
  rc = ensure_staging(f, (size_t)n * 22);
  if (rc) return rc;
  const FactorView fv = view(f);
  double* s = f->staging;
  if (eig_val) {
    vxk::launch_gather_rows(fv.eigval, f->VS, head, n, 3, s, f->stream);
    VX_HIP(f, hipMemcpyAsync(eig_val, s, sizeof(double) * n * 3, hipMemcpyDeviceToHost, f->stream));
  }
  s += (size_t)n * 3;
  if (eig_vec) {
    vxk::launch_gather_rows(fv.eigvec, f->VS, head, n, 9, s, f->stream);
    VX_HIP(f, hipMemcpyAsync(eig_vec, s, sizeof(double) * n * 9, hipMemcpyDeviceToHost, f->stream));
  }
  s += (size_t)n * 9;
  if (merged) {
    vxk::launch_gather_rows(fv.merged, f->VS, head, n, 10, s, f->stream);
    VX_HIP(f, hipMemcpyAsync(merged, s, sizeof(double) * n * 10, hipMemcpyDeviceToHost, f->stream));
  }
  VX_HIP(f, hipStreamSynchronize(f->stream));
  return VXBA_OK;
}

int vxba_snapshot_cache(vxba_factor* f) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  if (f->V == 0) return fail(f, VXBA_ERR_STATE, "snapshot_cache on an empty factor");
  hipSetDevice(f->device);
  if (f->snapshot_vs != f->VS) {
    if (f->snapshot) { VX_HIP(f, hipStreamSynchronize(f->stream)); VX_HIP(f, hipFree(f->snapshot)); f->snapshot = nullptr; }
    VX_HIP(f, hipMalloc((void**)&f->snapshot, (size_t)N_CACHE_PLANES * f->VS * sizeof(double)));
    f->snapshot_vs = f->VS;
  }
  VX_HIP(f, hipMemcpyAsync(f->snapshot, view(f).eigval, (size_t)N_CACHE_PLANES * f->VS * sizeof(double), hipMemcpyDeviceToDevice, f->stream));
  f->snapshot_v = f->V;
  return VXBA_OK;
}

int vxba_restore_cache(vxba_factor* f) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  if (!f->snapshot || f->snapshot_v != f->V || f->snapshot_vs != f->VS) return fail(f, VXBA_ERR_STATE, "no matching cache snapshot");
  hipSetDevice(f->device);
  VX_HIP(f, hipMemcpyAsync(view(f).eigval, f->snapshot, (size_t)N_CACHE_PLANES * f->VS * sizeof(double), hipMemcpyDeviceToDevice, f->stream));
  return VXBA_OK;
}

int vxba_plane_fit_judge(int device, int64_t n, const double* clusters, int min_point, double min_eigen_value, double eigen_ratio_thre,
                         double factor_ratio_max, double* eig_val, double* eig_vec, uint8_t* flags) {
  if (n < 0 || (n > 0 && (!clusters || !eig_val || !eig_vec))) return VXBA_ERR_ARG;
  if (n == 0) return VXBA_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_HIP;
  using vxs::Lease;
  Lease lease(device, Lease::padded((size_t)n * 10 * 8) + Lease::padded((size_t)n * 3 * 8) + Lease::padded((size_t)n * 9 * 8) + Lease::padded((size_t)n));
  if (!lease.ok()) return VXBA_ERR_HIP;
  double* d_c = lease.take<double>((size_t)n * 10 * 8);
  double* d_l = lease.take<double>((size_t)n * 3 * 8);
  double* d_u = lease.take<double>((size_t)n * 9 * 8);
  unsigned char* d_f = flags ? lease.take<unsigned char>((size_t)n) : nullptr;
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = hipMemcpy(d_c, clusters, (size_t)n * 10 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    vxk::PlaneCriteria pc{min_point, min_eigen_value, eigen_ratio_thre, factor_ratio_max};
    vxk::launch_k4_plane_fit(d_c, n, d_l, d_u, flags ? &pc : nullptr, d_f, nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(eig_val, d_l, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(eig_vec, d_u, (size_t)n * 9 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && flags) e = hipMemcpy(flags, d_f, (size_t)n, hipMemcpyDeviceToHost);
  return e == hipSuccess ? VXBA_OK : VXBA_ERR_HIP;
}

int vxba_plane_fit(int device, int64_t n, const double* clusters, double* eig_val, double* eig_vec) {
  return vxba_plane_fit_judge(device, n, clusters, 0, 0.0, 0.0, 0.0, eig_val, eig_vec, nullptr);
}

int vxba_build_clusters(int device, int64_t n_cells, int64_t n_points, const double* xyz, const int64_t* cell_ptr, double* clusters) {
  if (n_cells < 0 || n_points < 0 || !cell_ptr || !clusters || (n_points > 0 && !xyz)) return VXBA_ERR_ARG;
  if (n_cells == 0) return VXBA_OK;
  if (cell_ptr[0] != 0 || cell_ptr[n_cells] != n_points) return VXBA_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_HIP;
  using vxs::Lease;
  Lease lease(device, Lease::padded((size_t)n_points * 3 * 8) + Lease::padded((size_t)(n_cells + 1) * 8) + Lease::padded((size_t)n_cells * 10 * 8));
  if (!lease.ok()) return VXBA_ERR_HIP;
  double* d_xyz = lease.take<double>((size_t)n_points * 3 * 8);
  int64_t* d_ptr = lease.take<int64_t>((size_t)(n_cells + 1) * 8);
  double* d_cl = lease.take<double>((size_t)n_cells * 10 * 8);
  hipError_t e = hipSuccess;
  if (e == hipSuccess && n_points) e = hipMemcpy(d_xyz, xyz, (size_t)n_points * 3 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_ptr, cell_ptr, (size_t)(n_cells + 1) * sizeof(int64_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    vxk::launch_k1_build_aos(d_xyz, d_ptr, n_cells, d_cl, nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(clusters, d_cl, (size_t)n_cells * 10 * sizeof(double), hipMemcpyDeviceToHost);
  return e == hipSuccess ? VXBA_OK : VXBA_ERR_HIP;
}

// OctreeGBA::cut_voxel + recut on the GPU (vxba_voxelize.hip); the accepted voxels go straight from the staging arrays into
// the factor's planes -- nothing returns to the host except their count (and the ids, if asked for).
static int voxelize_push_impl(vxba_factor* f, int64_t n_points, const double* xyz_local, bool xyz_on_device, const int64_t* frame_ptr, const double* Rp,
                              const vxba_voxelize_params* params, int64_t* n_pushed, uint64_t* node_ids, int64_t ids_capacity) {
  VX_LOCK(f);
  if (!f || n_points < 0 || !frame_ptr || !Rp || !params || !n_pushed || (n_points > 0 && !xyz_local))
    return fail(f, VXBA_ERR_ARG, "voxelize_push: null argument");
  if (params->max_layer < 0 || params->max_layer > 3 || !(params->voxel_size > 0)) return fail(f, VXBA_ERR_ARG, "voxelize_push: max_layer in 0..3, voxel_size > 0");
  if (frame_ptr[0] != 0 || frame_ptr[f->W] != n_points) return fail(f, VXBA_ERR_ARG, "voxelize_push: frame_ptr must span [0, n_points]");
  if (n_points > 0xffffffffll) return fail(f, VXBA_ERR_UNSUPPORTED, "voxelize_push: more than 2^32 points");
  *n_pushed = 0;
  if (n_points == 0) return VXBA_OK;
  hipSetDevice(f->device);
  const int W = f->W;
  int floor_pts = params->min_points;
  for (int k = 0; k <= params->max_layer; k++)
    if (params->min_points_layer[k] > 0) floor_pts = std::min(floor_pts, params->min_points_layer[k]);
  const int64_t cap = n_points / (std::max(floor_pts, 0) + 1) + 1;   // a factor owns > min_points points, and no point twice
  // one grow-only scratch allocation per factor, carved up here (a hipMalloc / hipFree pair per buffer and call cost more than the sorts)
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  // wide windows take the accepted voxels as compressed rows (a point belongs to at most one factor voxel: n_points entries suffice)
  const bool csr = is_wide(f);
  const size_t b_xyz = xyz_on_device ? 0 : up((size_t)n_points * 3 * sizeof(double)), b_fp = up((size_t)(W + 1) * sizeof(long long)),
               b_cl = csr ? up((size_t)n_points * 10 * sizeof(double)) : up((size_t)cap * W * 10 * sizeof(double)),
               b_rp = csr ? up((size_t)(cap + 1) * sizeof(long long)) : 0, b_ef = csr ? up((size_t)n_points * sizeof(int)) : 0, b_ev = up((size_t)cap * 3 * sizeof(double)), b_evec = up((size_t)cap * 9 * sizeof(double)),
               b_m = up((size_t)cap * 10 * sizeof(double)), b_id = up((size_t)cap * sizeof(unsigned long long)), b_fix = up((size_t)cap * 10 * sizeof(double)),
               b_coe = up((size_t)cap * sizeof(double));
  const size_t need = b_xyz + b_fp + b_cl + b_rp + b_ef + b_ev + b_evec + b_m + b_id + b_fix + b_coe + 256;
  {
    int rcs = ensure_scratch(f, need);
    if (rcs) return rcs;
  }
  char* q = f->d_scratch;
  auto carve = [&](size_t bytes) { char* r = q; q += bytes; return r; };
  double* d_xyz_own = (double*)carve(b_xyz);
  long long* d_fp = (long long*)carve(b_fp);
  double* d_cl = (double*)carve(b_cl);
  long long* d_rp = (long long*)carve(b_rp); int* d_ef = (int*)carve(b_ef);
  double* d_ev = (double*)carve(b_ev); double* d_evec = (double*)carve(b_evec); double* d_m = (double*)carve(b_m);
  unsigned long long* d_id = (unsigned long long*)carve(b_id);
  double* d_fix = (double*)carve(b_fix); double* d_coe = (double*)carve(b_coe);
  int* d_bad = (int*)carve(256);
  const double* d_xyz = xyz_on_device ? xyz_local : d_xyz_own;
  if (!xyz_on_device) VX_HIP(f, hipMemcpyAsync(d_xyz_own, xyz_local, (size_t)n_points * 3 * sizeof(double), hipMemcpyHostToDevice, f->stream));
  VX_HIP(f, hipMemcpyAsync(d_fp, frame_ptr, (size_t)(W + 1) * sizeof(long long), hipMemcpyHostToDevice, f->stream));
  int rcp = upload_poses(f, Rp);
  if (rcp) return rcp;
  vxv::VoxelizeParams vp;
  vp.voxel_size = params->voxel_size; vp.max_layer = params->max_layer; vp.min_points = params->min_points;
  vp.min_eigen_value = params->min_eigen_value; vp.factor_ratio_max = params->factor_ratio_max;
  for (int k = 0; k < 4; k++) { vp.eigen_ratio[k] = params->eigen_ratio[k]; vp.min_points_layer[k] = params->min_points_layer[k]; }
  vp.min_frames = params->min_frames;
  vp.shard_index = params->shard_index; vp.shard_count = params->shard_count;
  if (vp.shard_count > 1 && (vp.shard_index < 0 || vp.shard_index >= vp.shard_count)) return fail(f, VXBA_ERR_ARG, "voxelize_push: shard_index outside 0 .. shard_count-1");
  vxv::VoxelizeOutput out{cap, d_cl, d_ev, d_evec, d_m, d_id};
  if (csr) { out.d_row_ptr = d_rp; out.d_eframe = d_ef; out.ecap = n_points; }
  const char* emsg = nullptr;
  const long long n = vxv::voxelize(W, n_points, d_xyz, d_fp, f->d_poses, vp, f->stream, &out, &emsg);
  if (n < 0) return fail(f, VXBA_ERR_STATE, emsg ? emsg : "voxelize failed");
  if (n > 0) {
    int rc = ensure_capacity(f, f->V + (int)n);
    if (rc) return rc;
    const FactorView fv = view(f);
    const int v0 = f->V;
    vxv::fill(d_fix, n * 10, 0.0, f->stream);
    vxv::fill(d_coe, n, 1.0, f->stream);
    if (csr) {
      const char* em2 = nullptr;
      if (vxw::store_reserve(f->wstore, f->V + (int)n, f->wstore.nnz + out.n_entries, f->V, f->stream, &em2) != 0) return fail(f, VXBA_ERR_HIP, em2 ? em2 : "wide store: allocation failed");
      VX_HIP(f, hipMemsetAsync(d_bad, 0, sizeof(int), f->stream));   // rows the voxelisation wrote itself: never set
      vxw::store_append_csr(f->wstore, f->V, (int)n, d_rp, d_ef, d_cl, out.n_entries, W, d_bad, f->stream);
      f->wstore.nnz += out.n_entries;
    } else {
      vxk::launch_scatter_clusters(d_cl, fv, v0, (int)n, f->stream);
      vxk::launch_build_clb(fv, v0, (int)n, f->stream);
      clusters_written(f, v0);
    }
    vxk::launch_scatter_voxel_records(d_fix, d_coe, d_ev, d_evec, d_m, fv, v0, (int)n, f->stream);
    vxk::launch_seed_aux(fv, v0, v0 + (int)n, f->stream);
    if (node_ids && ids_capacity > 0)
      VX_HIP(f, hipMemcpyAsync(node_ids, d_id, (size_t)std::min<int64_t>(n, ids_capacity) * sizeof(uint64_t), hipMemcpyDeviceToHost, f->stream));
    {   // completion by polling (a blocking wait parks the thread: ~25 us to wake up from, and a hierarchical pass makes thousands of them)
      hipError_t q;
      q = vxwait::stream_wait(f->stream);
      VX_HIP(f, q);
    }
    VX_HIP(f, hipGetLastError());
    f->V += (int)n;
    f->wide_dirty = true;
  }
  *n_pushed = n;
  release_push_buffers(f);
  return VXBA_OK;
}

int vxba_voxelize_push(vxba_factor* f, int64_t n_points, const double* xyz_local, const int64_t* frame_ptr, const double* Rp,
                       const vxba_voxelize_params* params, int64_t* n_pushed, uint64_t* node_ids, int64_t ids_capacity) {
  return voxelize_push_impl(f, n_points, xyz_local, false, frame_ptr, Rp, params, n_pushed, node_ids, ids_capacity);
}
int vxba_voxelize_push_device(vxba_factor* f, int64_t n_points, const double* d_xyz_local, const int64_t* frame_ptr, const double* Rp,
                              const vxba_voxelize_params* params, int64_t* n_pushed, uint64_t* node_ids, int64_t ids_capacity) {
  return voxelize_push_impl(f, n_points, d_xyz_local, true, frame_ptr, Rp, params, n_pushed, node_ids, ids_capacity);
}

int vxba_debug_mfma_probe(int device, const double* A16x4, const double* B4x16, double* D16x16) {
  if (!A16x4 || !B4x16 || !D16x16) return VXBA_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_NODEV;
  double* d = nullptr;
  if (hipMalloc((void**)&d, (64 + 64 + 256) * sizeof(double)) != hipSuccess) return VXBA_ERR_HIP;
  hipError_t e = hipMemcpy(d, A16x4, 64 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + 64, B4x16, 64 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) { vxk::launch_mfma_probe(d, d + 64, d + 128, nullptr); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpy(D16x16, d + 128, 256 * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(d);
  return e == hipSuccess ? VXBA_OK : VXBA_ERR_HIP;
}

// development: the Hessian sweep's workgroup partials as the last sweep left them (n doubles from the start of the buffer)
int vxba_debug_partials(vxba_factor* f, double* out, size_t n) {
  if (!f || !out || !f->d_partial3 || n > f->partial3_len) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  if (hipStreamSynchronize(f->stream) != hipSuccess) return VXBA_ERR_HIP;
  return hipMemcpy(out, f->d_partial3, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess ? VXBA_OK : VXBA_ERR_HIP;
}
int vxba_debug_stamps(int clear, unsigned long long* out, size_t n) {
  if (clear) vxk::debug_clear_stamps();
  if (out && n) { (void)hipDeviceSynchronize(); vxk::debug_read_stamps(out, n); }
  return VXBA_OK;
}

// ---- links for vxba_map.hip (vxba_internal.h) ----
int vxba_internal_push_voxels_device(vxba_factor* f, int n, const double* d_clusters, const double* d_fix, const double* d_coe, const double* d_eigval,
                                     const double* d_eigvec, const double* d_merged) {
  VX_LOCK(f);
  if (!f || n < 0 || (n > 0 && (!d_clusters || !d_fix || !d_coe || !d_eigval || !d_eigvec || !d_merged))) return fail(f, VXBA_ERR_ARG, "push_voxels_device: null argument");
  if (n == 0) return VXBA_OK;
  hipSetDevice(f->device);
  int rc = ensure_capacity(f, f->V + n);
  if (rc) return rc;
  const FactorView fv = view(f);
  const int v0 = f->V;
  if (is_wide(f)) { rc = wide_append_dense(f, n, d_clusters); if (rc) return rc; }
  else {
    vxk::launch_scatter_clusters(d_clusters, fv, v0, n, f->stream);
    vxk::launch_build_clb(fv, v0, n, f->stream);
    clusters_written(f, v0);
  }
  vxk::launch_scatter_voxel_records(d_fix, d_coe, d_eigval, d_eigvec, d_merged, fv, v0, n, f->stream);   // (five launches of scatter_rows until round 4)
  vxk::launch_seed_aux(fv, v0, v0 + n, f->stream);
  {   // completion by polling: the map's stage buffers are reused right after this call, and a blocking wait costs ~25 us to wake up from
    hipError_t q;
    q = vxwait::stream_wait(f->stream);
    VX_HIP(f, q);
  }
  VX_HIP(f, hipGetLastError());
  f->V += n;
  f->wide_dirty = true;
  return VXBA_OK;
}
int vxba_internal_factor_device(const vxba_factor* f) { return f ? f->device : -1; }
int vxba_internal_cache_view(vxba_factor* f, const double** eigval, const double** eigvec, const double** merged, int* VS, int* V) {
  VX_LOCK(f);
  if (!f || !eigval || !eigvec || !merged || !VS || !V) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  VX_HIP(f, hipStreamSynchronize(f->stream));
  const FactorView fv = view(f);
  *eigval = fv.eigval; *eigvec = fv.eigvec; *merged = fv.merged; *VS = f->VS; *V = f->V;
  return VXBA_OK;
}

// Test access to the structured LiDAR-inertial solve (host code, needs no GPU): A (m x m, full symmetric), b -> x.
int vxba_debug_band_schur(int m, const double* A, const double* b, int nframes, int lead_y, int tail_x, double* x) {
  if (!A || !b || !x || m != lead_y + 15 * nframes + tail_x) return VXBA_ERR_ARG;
  const vxh::LiIndexSets s = vxh::li_index_sets(nframes, lead_y, tail_x);
  vxh::BandSchurWork w;
  return vxh::band_schur_solve(m, A, m, nullptr, b, s.Y.data(), (int)s.Y.size(), s.bw, s.X.data(), (int)s.X.size(), s.xlo.data(), x, w) ? VXBA_OK : VXBA_ERR_STATE;
}

int vxba_set_option(vxba_factor* f, int option, int value) {
  if (!f) return VXBA_ERR_ARG;
  VX_LOCK(f);
  switch (option) {
    case VXBA_OPT_LI_DEVICE_LOOP:
      if (value != 0) return fail(f, VXBA_ERR_UNSUPPORTED, "vxba_set_option: the device-resident 15W loop was removed in round 4 (4x slower than the default shell); only 0 is accepted");
      break;
    case VXBA_OPT_FUSED_SOLVE: case VXBA_OPT_SPEC_COLLECTIVE: case VXBA_OPT_WIDE_DEVICE_SOLVE:
    case VXBA_OPT_LI_STRUCTURED_SOLVE: case VXBA_OPT_LI_QUEUED_SWEEPS: case VXBA_OPT_LI_DEVICE_POSE_SOLVE:
      if (value != 0 && value != 1) return fail(f, VXBA_ERR_ARG, "vxba_set_option: this option takes 0 or 1");
      break;
    case VXBA_OPT_FUSED_SWEEPS:
      if (value < 0 || value > 2) return fail(f, VXBA_ERR_ARG, "vxba_set_option: VXBA_OPT_FUSED_SWEEPS takes 0, 1 (fused unless the last call was reject-heavy) or 2 (always)");
      f->reject_heavy = false;   // a fresh setting starts without history
      break;
    case VXBA_OPT_DEBUG_SOLVE_TIMEOUT:
      if (value < 0 || value > 2) return fail(f, VXBA_ERR_ARG, "vxba_set_option: the test hook takes 0, 1 or 2");
      break;
    case VXBA_OPT_K2_VOXELS_PER_BLOCK:
      if (value < 32 || value > 64) return fail(f, VXBA_ERR_ARG, "vxba_set_option: voxels per block must be in [32, 64]");
      break;
    default: return fail(f, VXBA_ERR_ARG, "vxba_set_option: unknown option");
  }
  f->opt[option] = value;
  return VXBA_OK;
}
int vxba_get_option(const vxba_factor* f, int option, int* value) {
  if (f && value && option == VXBA_STAT_FUSED_FALLBACKS) { *value = f->fused_fallbacks; return VXBA_OK; }
  if (f && value && option == VXBA_STAT_LI_DEVICE_FALLBACKS) { *value = f->li_dev_fallbacks; return VXBA_OK; }
  if (f && value && option == VXBA_STAT_REJECT_HEAVY) { *value = f->reject_heavy ? 1 : 0; return VXBA_OK; }
  if (f && value && option == VXBA_STAT_LI_LAST_CALL_US) { *value = (int)(f->li_last_call_us + 0.5); return VXBA_OK; }
  if (!f || !value || option < 0 || option >= VXBA_OPT_COUNT) return VXBA_ERR_ARG;
  *value = f->opt[option];
  return VXBA_OK;
}

int vxba_set_precision(vxba_factor* f, int mode) {
  VX_LOCK(f);
  if (!f || (mode != VXBA_PRECISION_F64 && mode != VXBA_PRECISION_MIXED && mode != VXBA_PRECISION_MIXED_F32_CLUSTERS))
    return fail(f, VXBA_ERR_ARG, "set_precision: mode must be VXBA_PRECISION_F64, VXBA_PRECISION_MIXED or VXBA_PRECISION_MIXED_F32_CLUSTERS");
  if (mode != VXBA_PRECISION_F64) VX_NARROW_ONLY(f, "mixed precision");
  f->precision = mode;
  return VXBA_OK;
}

int vxba_set_profiling(vxba_factor* f, int on) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  f->profiling = on;
  return VXBA_OK;
}

int vxba_get_kernel_times(vxba_factor* f, double ms_sum[4], int64_t calls[4], int reset) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  int rc = drain_events(f);
  if (rc) return rc;
  for (int k = 0; k < 4; k++) {
    if (ms_sum) ms_sum[k] = f->ms_sum[k];
    if (calls) calls[k] = f->calls[k];
    if (reset) { f->ms_sum[k] = 0; f->calls[k] = 0; }
  }
  return VXBA_OK;
}

int vxba_get_fused_time(vxba_factor* f, double* ms_sum, int64_t* calls, int reset) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  int rc = drain_events(f);
  if (rc) return rc;
  if (ms_sum) *ms_sum = f->ms_sum[5];
  if (calls) *calls = f->calls[5];
  if (reset) { f->ms_sum[5] = 0; f->calls[5] = 0; }
  return VXBA_OK;
}

int vxba_get_collective_time(vxba_factor* f, double* ms_sum, int64_t* calls, int reset) {
  VX_LOCK(f);
  if (!f) return VXBA_ERR_ARG;
  hipSetDevice(f->device);
  int rc = drain_events(f);
  if (rc) return rc;
  if (ms_sum) *ms_sum = f->ms_sum[4];
  if (calls) *calls = f->calls[4];
  if (reset) { f->ms_sum[4] = 0; f->calls[4] = 0; }
  return VXBA_OK;
}

int vxba_nnz(vxba_factor* f, int64_t* nnz) {
  VX_LOCK(f);
  if (!f || !nnz) return VXBA_ERR_ARG;
  *nnz = 0;
  if (f->V == 0) return VXBA_OK;
  hipSetDevice(f->device);
  if (is_wide(f)) {
    const char* emsg = nullptr;
    const long long c = vxw::store_count_observed(f->wstore, f->V, f->stream, &emsg);
    if (c < 0) return fail(f, VXBA_ERR_HIP, emsg ? emsg : "nnz: count failed");
    *nnz = c;
    return VXBA_OK;
  }
  VX_HIP(f, hipMemsetAsync(f->d_count, 0, sizeof(unsigned long long), f->stream));
  vxk::launch_count_nnz(view(f), f->V, f->d_count, f->stream);
  unsigned long long h = 0;
  VX_HIP(f, hipMemcpyAsync(&h, f->d_count, sizeof h, hipMemcpyDeviceToHost, f->stream));
  VX_HIP(f, hipStreamSynchronize(f->stream));
  *nnz = (int64_t)h;
  return VXBA_OK;
}

int vxba_device_bytes(const vxba_factor* f, int64_t bytes[4]) {
  VX_LOCK(const_cast<vxba_factor*>(f));
  if (!f || !bytes) return VXBA_ERR_ARG;
  const size_t d = sizeof(double);
  size_t store = (size_t)n_planes(f) * f->VS * d + (f->snapshot ? (size_t)N_CACHE_PLANES * f->snapshot_vs * d : 0);
  if (is_wide(f)) store += vxw::store_bytes(f->wstore);
  else if (f->clb) store += vxk::k3_clb_len(f->W, f->VS) * d;
  if (f->cl32) store += (size_t)10 * f->W * f->cl32_vs * sizeof(float);
  size_t work = f->partial2_len * d + f->partial3_len * d + f->xlen * d + sizeof(vxk::LMState);
  if (is_wide(f)) work += vxw::index_bytes(f->wide, f->W) + vxw::wide_solver_bytes(f->wide_solver) + sizeof(double) * 12 * VXBA_MAX_WIN_WIDE;
  const size_t scratch = f->staging_len * d + f->scratch_cap;
  bytes[0] = (int64_t)store; bytes[1] = (int64_t)work; bytes[2] = (int64_t)scratch; bytes[3] = (int64_t)(store + work + scratch);
  return VXBA_OK;
}

int vxba_algorithmic_bytes(const vxba_factor* f, double bytes[2]) {
  VX_LOCK(const_cast<vxba_factor*>(f));
  if (!f || !bytes) return VXBA_ERR_ARG;
  int64_t nnz = 0;
  int rc = vxba_nnz(const_cast<vxba_factor*>(f), &nnz);
  if (rc) return rc;
  // K3 = 80 nnz + 136 V read;  K2 = 80 nnz + 88 V read + 176 V written
  bytes[0] = 80.0 * (double)nnz + 136.0 * f->V;
  bytes[1] = (f->precision == VXBA_PRECISION_MIXED_F32_CLUSTERS ? 40.0 : 80.0) * (double)nnz + 264.0 * f->V;   // f32 cluster rows: 10 floats per entry
  return VXBA_OK;
}

}  // extern "C"
