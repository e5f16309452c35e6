// The bottom-up pass of the hierarchical global BA below the C ABI (include/vxba.h: vxba_hba_*): BASELINE configs[4].
//
// Reference: thd_globalmapping's loop over windows of `wdsize` keyframes with stride `mgsize` (voxelslam.cpp:2485-2595), each refined
// by HBA_add_edge (:2320-2482: OctreeGBA::cut_voxel + OctreeGBA_multi_recut -> Lidar_BA_Optimizer::damping_iter(.., 4, ..) under the
// coarse -> fine re-voxelisation schedule :2362-2398, Hessian -> PGO edge weights :2405-2427, merged + voxel-filtered submap
// :2430-2450), then ONE HBA_add_edge over all submap poses.
//
// Rounds 1-4 ran that loop in Python (voxel_slam_amd/hba.py): per window a numpy concatenation, a 4.8 MB upload, the refinement, the
// edge loop, the submap transform in numpy and a voxel filter with an upload and a download of its own -- 0.52 s per 500-keyframe
// pass with 0.15 s of kernels in it (profiles/r05_cfg5/host_profile_python_path.txt).  Here the keyframe clouds are uploaded ONCE
// (vxba_hba_add_keyframes) and everything per window happens on the device:
//   widen (f32 -> f64 of the window's contiguous run of points)  ->  vxba_voxelize_push_device  ->  vxba_damping_iter
//   ->  transform into the first keyframe's coordinates (float, like PointType)  ->  voxel filter (device to device)
// with the submaps staying in HBM for the top level.  Up to eight host threads drive a stream and a bottom-level factor each (four by default: 0.162 / 0.113 / 0.088-0.094 / 0.092-0.095 s per
// 500-keyframe pass with 1 / 2 / 4 / 6 threads on one box): while one waits for the few counters a voxelisation brings back -- polled out of
// pinned host memory, not waited for in the runtime --, the others' kernels run.  Only poses, Hessians and counts cross PCIe.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vxba.h"
#include "vxba_downsample.h"

namespace vxhba {

__global__ void widen_f32_kernel(const float* __restrict__ src, long long n3, double* __restrict__ dst) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n3) dst[q] = (double)src[q];
}

// merged submap (voxelslam.cpp:2430-2446): point of frame i -> dR_i x + dp_i (double), stored as float.  T: per frame [dR row-major 9 | dp 3].
constexpr int MAX_WD = 16;
struct FrameXf { double T[MAX_WD][12]; long long fp[MAX_WD + 1]; int W; };
__global__ void merge_kernel(const float* __restrict__ src, long long n, FrameXf xf, float* __restrict__ dst) {
#pragma clang fp contract(off)
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  int i = 0;
  while (i + 1 < xf.W && q >= xf.fp[i + 1]) i++;
  const double x = src[3 * q], y = src[3 * q + 1], z = src[3 * q + 2];
  const double* t = xf.T[i];
#pragma unroll
  for (int j = 0; j < 3; j++) dst[3 * q + j] = (float)(((x * t[3 * j] + y * t[3 * j + 1]) + z * t[3 * j + 2]) + t[9 + j]);
}

constexpr int MAX_THREADS = 8;   // host threads / streams of the bottom level
struct Worker {
  vxba_factor* f = nullptr;
  hipStream_t s = nullptr;
  double* d_win = nullptr;     // the window's points, f64 (voxelize input)
  float* d_merged = nullptr;   // ... transformed into the first keyframe's frame, f32
  size_t cap = 0;              // points
  vxd::Scratch ds;
  std::string err;
  int rc = VXBA_OK;
};

}  // namespace vxhba

struct vxba_hba {
  int device = 0;
  std::string err;
  float* d_xyz = nullptr;              // every keyframe cloud, in order
  size_t cap_pts = 0, n_pts = 0;
  std::vector<int64_t> cloud_ptr{0};   // K + 1 offsets into d_xyz (points)
  vxhba::Worker wk[vxhba::MAX_THREADS];
  int bottom_w = 0;
  vxba_factor* tail_f = nullptr;       // factor of the closing short window (its own window size)
  int tail_w = 0;
  int threads_used = 0;                // host threads of the last pass
  vxba_factor* top = nullptr;
  int top_w = 0;
  float* d_sub = nullptr;              // the submaps, window w at sub_off[w]
  size_t cap_sub = 0;
  double* d_top = nullptr;
  size_t cap_top = 0;
  // geometry of the pass in progress (vxba_hba_bottom .. vxba_hba_top)
  int p_K = 0, p_wd = 0, p_mg = 0, p_tail = 0, p_S = 0;
  std::vector<int64_t> sub_off, sizes;     // S + 1 upper-bound offsets into d_sub; points per voxel-filtered submap (-1: not here yet)
};

namespace vxhba {

static int fail(vxba_hba* h, int rc, const std::string& m) { if (h) h->err = m; return rc; }
#define HB(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(h, VXBA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// R (column-major in the C-ABI pose record [R 9 | p 3]) as a row-major 3 x 3
static void pose_R(const double* Rp, double R[9]) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = Rp[3 * c + r]; }

struct Edge { int i, j; double rot[9], tra[3], v6[6]; };
// voxelslam.cpp:2405-2427: every frame pair whose six entries hess(6i+k, 6j+k) all reach min_abs in magnitude
static void edges_from_hessian(const double* poses, const double* hess, int W, const int64_t* ids, std::vector<Edge>& out) {
  const int n = 6 * W;
  for (int i = 0; i < W - 1; i++)
    for (int j = i + 1; j < W; j++) {
      double hc[6];
      bool ok = true;
      for (int k = 0; k < 6; k++) { hc[k] = std::fabs(hess[(size_t)(6 * j + k) * n + (6 * i + k)]); if (hc[k] < 1e-6) ok = false; }   // column-major: (row 6i+k, col 6j+k)
      if (!ok) continue;
      Edge e;
      e.i = (int)ids[i]; e.j = (int)ids[j];
      double Ri[9], Rj[9];
      pose_R(poses + 12 * i, Ri); pose_R(poses + 12 * j, Rj);
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) e.rot[3 * r + c] = Ri[r] * Rj[c] + Ri[3 + r] * Rj[3 + c] + Ri[6 + r] * Rj[6 + c];      // Ri^T Rj
      for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += Ri[3 * k + r] * (poses[12 * j + 9 + k] - poses[12 * i + 9 + k]);
        e.tra[r] = s;
      }
      for (int k = 0; k < 6; k++) e.v6[k] = 1.0 / hc[k];
      out.push_back(e);
    }
}

// n_threads <= 0: four host threads (measured best on one MI355X: 0.162 / 0.113 / 0.09 / 0.093 s per 500-keyframe pass with 1 / 2 / 4 / 6),
// fewer where the container's CPU quota (cgroup v2 cpu.max) is small: the threads poll while their streams run, and a container that spends
// its quota on polling is throttled as a whole (0.49 s per pass with four threads against 0.17 with one on such a box).
static int default_threads() {
  int n = 4;
  if (FILE* fp = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    double per = 0;
    if (std::fscanf(fp, "%31s %lf", q, &per) == 2 && q[0] != 'm' && per > 0) {
      const double cores = std::atof(q) / per;
      const int lim = (int)(cores / 3.0);      // the pass's threads + the process's other threads stay well below the quota
      n = lim < 1 ? 1 : (lim < n ? lim : n);
    }
    std::fclose(fp);
  }
  return n;
}

struct RoundLog { int64_t n_voxels; double r0, r1; int converged, fine; };
// HBA_add_edge's loop (voxelslam.cpp:2362-2398; voxel_slam_amd/hba.py window_refine): coarse parameters first, the odometry's finer ones
// for the last round.  d_xyz: the window's points (f64, device), fp: W + 1 offsets.  xs in / out.
static int window_refine(vxba_factor* f, int W, int64_t n, const double* d_xyz, const int64_t* fp, double* xs, const vxba_voxelize_params* coarse,
                         const vxba_voxelize_params* fine, int max_iter, std::vector<double>& hess, std::vector<RoundLog>* log, std::string& err) {
  int converge_flag = 0;
  double thre = 0.05;
  hess.assign((size_t)36 * W * W, 0.0);
  for (int it = 0; it < max_iter; it++) {
    const bool use_fine = converge_flag == 1 || it == max_iter - 1;
    int rc = vxba_clear(f);
    int64_t nv = 0;
    if (rc == VXBA_OK) rc = vxba_voxelize_push_device(f, n, d_xyz, fp, xs, use_fine ? fine : coarse, &nv, nullptr, 0);
    double resis[2] = {0, 0};
    int n_trace = 0, conv = 0;
    if (rc == VXBA_OK && nv == 0) {
      // no factor voxel in this window (too few points for any plane): upstream's damping_iter then runs on an all-zero system -- Eigen's LDLT
      // returns a zero step, the poses stay, *hess is zero (no edge passes the 1e-6 test), residuals 0 and 0 / 0 -- and the loop's schedule goes on
      // (voxelslam.cpp:2380-2398).  The optimiser entry point refuses an empty factor, so the same outcome is written down here.
      std::fill(hess.begin(), hess.end(), 0.0);
      resis[0] = 0.0; resis[1] = std::nan("");
    } else if (rc == VXBA_OK) rc = vxba_damping_iter(f, xs, 4, hess.data(), resis, nullptr, &n_trace, &conv);
    if (rc != VXBA_OK) { err = vxba_last_error(f); return rc; }
    if (log) log->push_back(RoundLog{nv, resis[0], resis[1], conv, use_fine ? 1 : 0});
    if ((std::fabs(resis[0] - resis[1]) / resis[0] < thre && conv) || (it == max_iter - 2 && converge_flag == 0)) {
      thre = 0.01;
      if (converge_flag == 0) converge_flag = 1;
      else if (converge_flag == 1) break;
    }
  }
  return VXBA_OK;
}

}  // namespace vxhba

using namespace vxhba;

extern "C" {

int vxba_hba_create(int device, vxba_hba** out) {
  if (!out) return VXBA_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  vxba_hba* h = new vxba_hba();
  h->device = device;
  *out = h;
  return VXBA_OK;
}

const char* vxba_hba_last_error(const vxba_hba* h) { return h ? h->err.c_str() : "null handle"; }

int vxba_hba_clear(vxba_hba* h) {
  if (!h) return VXBA_ERR_ARG;
  h->n_pts = 0;
  h->cloud_ptr.assign(1, 0);
  h->p_K = h->p_S = 0;
  return VXBA_OK;
}

int vxba_hba_destroy(vxba_hba* h) {
  if (!h) return VXBA_OK;
  hipSetDevice(h->device);
  for (Worker& w : h->wk) {
    if (w.f) vxba_destroy(w.f);
    if (w.s) { hipStreamSynchronize(w.s); hipStreamDestroy(w.s); }
    if (w.d_win) hipFree(w.d_win);
    if (w.d_merged) hipFree(w.d_merged);
    w.ds.release();
  }
  if (h->top) vxba_destroy(h->top);
  if (h->tail_f) vxba_destroy(h->tail_f);
  if (h->d_xyz) hipFree(h->d_xyz);
  if (h->d_sub) hipFree(h->d_sub);
  if (h->d_top) hipFree(h->d_top);
  delete h;
  return VXBA_OK;
}

int vxba_hba_num_keyframes(const vxba_hba* h) { return h ? (int)h->cloud_ptr.size() - 1 : 0; }
int vxba_hba_threads_used(const vxba_hba* h) { return h ? h->threads_used : 0; }

int vxba_hba_add_keyframes(vxba_hba* h, int64_t n_keyframes, const int64_t* cloud_ptr, const float* xyz) {
  if (!h || n_keyframes < 0 || !cloud_ptr || cloud_ptr[0] != 0) return fail(h, VXBA_ERR_ARG, "hba_add_keyframes: bad argument");
  for (int64_t k = 0; k < n_keyframes; k++) if (cloud_ptr[k + 1] < cloud_ptr[k]) return fail(h, VXBA_ERR_ARG, "hba_add_keyframes: cloud_ptr must be non-decreasing");
  const int64_t n = cloud_ptr[n_keyframes];
  if (n > 0 && !xyz) return fail(h, VXBA_ERR_ARG, "hba_add_keyframes: null points");
  HB(hipSetDevice(h->device));
  if (h->n_pts + (size_t)n > h->cap_pts) {     // grow, keeping what is there
    const size_t want = (h->n_pts + (size_t)n) * 5 / 4 + 1024;
    float* p = nullptr;
    HB(hipMalloc((void**)&p, want * 3 * sizeof(float)));
    if (h->n_pts && hipMemcpy(p, h->d_xyz, h->n_pts * 3 * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) {
      hipFree(p);
      return fail(h, VXBA_ERR_HIP, "hba_add_keyframes: copying the resident keyframes failed");
    }
    if (h->d_xyz) hipFree(h->d_xyz);
    h->d_xyz = p;
    h->cap_pts = want;
  }
  if (n > 0) HB(hipMemcpy(h->d_xyz + 3 * h->n_pts, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  for (int64_t k = 0; k < n_keyframes; k++) h->cloud_ptr.push_back((int64_t)h->n_pts + cloud_ptr[k + 1]);
  h->n_pts += (size_t)n;
  return VXBA_OK;
}

}  // extern "C"

// ---- windows of a pass (thd_globalmapping, voxelslam.cpp:2498-2575) ---------------------------------------------------------------
// Full windows of wdsize keyframes at stride mgsize: a window runs whenever localID has reached wdsize keyframes, then mgsize of them are
// popped (:2536-2541, :2571-2574).  tail != 0: the CLOSING iteration as well (total_ba == 1, :2519-2523 -- it skips the "localID.size() <
// wdsize" test and runs HBA_add_edge on whatever localID still holds: the keyframes left behind the last pop, [S mgsize, K), and their
// submap joins the top level).  A session shorter than one window is that closing window alone.
static int full_windows(int K, int wd, int mg) { return K >= wd ? (K - wd) / mg + 1 : 0; }
static int tail_first(int K, int wd, int mg) { return full_windows(K, wd, mg) * mg; }

extern "C" {

int vxba_hba_num_windows(int n_keyframes, int wdsize, int mgsize, int tail) {
  if (n_keyframes < 0 || wdsize < 1 || mgsize < 1) return 0;
  const int S = full_windows(n_keyframes, wdsize, mgsize);
  return S + ((tail && tail_first(n_keyframes, wdsize, mgsize) < n_keyframes) ? 1 : 0);
}
int vxba_hba_window(int n_keyframes, int wdsize, int mgsize, int tail, int w, int* first, int* count) {
  const int n = vxba_hba_num_windows(n_keyframes, wdsize, mgsize, tail), S = full_windows(n_keyframes, wdsize, mgsize);
  if (w < 0 || w >= n || !first || !count) return VXBA_ERR_ARG;
  *first = w * mgsize;
  *count = w < S ? wdsize : n_keyframes - S * mgsize;
  return VXBA_OK;
}

}  // extern "C"

namespace vxhba {

static int pass_geometry(vxba_hba* h, int wdsize, int mgsize, int tail) {
  const int K = (int)h->cloud_ptr.size() - 1;
  if (wdsize < 2 || wdsize > VXBA_MAX_WIN || wdsize > MAX_WD || mgsize < 1) return fail(h, VXBA_ERR_ARG, "hba pass: need 2 <= wdsize <= VXBA_MAX_WIN and mgsize >= 1");
  const int S = vxba_hba_num_windows(K, wdsize, mgsize, tail);
  if (S < 1) return fail(h, VXBA_ERR_ARG, "hba pass: no window (fewer keyframes than wdsize and no closing window asked for, or no keyframes)");
  if (S > VXBA_MAX_WIN_WIDE) return fail(h, VXBA_ERR_UNSUPPORTED, "hba pass: more submaps than VXBA_MAX_WIN_WIDE");
  if (h->p_K != K || h->p_wd != wdsize || h->p_mg != mgsize || h->p_tail != tail || h->p_S != S || (int)h->sub_off.size() != S + 1) {
    h->p_K = K; h->p_wd = wdsize; h->p_mg = mgsize; h->p_tail = tail; h->p_S = S;
    h->sub_off.assign(S + 1, 0);
    for (int w = 0; w < S; w++) {
      int f0 = 0, cnt = 0;
      vxba_hba_window(K, wdsize, mgsize, tail, w, &f0, &cnt);
      h->sub_off[w + 1] = h->sub_off[w] + (h->cloud_ptr[f0 + cnt] - h->cloud_ptr[f0]);     // upper bound: the filter only removes points
    }
  }
  h->sizes.assign(S, -1);
  HB(hipSetDevice(h->device));
  if ((size_t)h->sub_off[S] > h->cap_sub) {
    if (h->d_sub) hipFree(h->d_sub);
    h->d_sub = nullptr; h->cap_sub = 0;
    HB(hipMalloc((void**)&h->d_sub, std::max<size_t>(1, (size_t)h->sub_off[S]) * 3 * sizeof(float)));
    h->cap_sub = (size_t)h->sub_off[S];
  }
  return VXBA_OK;
}

}  // namespace vxhba

extern "C" {

// Bottom level: the windows w_first, w_first + w_stride, .. of the pass on this device (one rank: 0, 1; rank r of N: r, N -- the windows
// are independent HBA_add_edge problems), window k of them on host thread k mod n_threads.  Fills the rows of submap_poses / submap_sizes
// of THESE windows, leaves their merged, voxel-filtered submaps in device memory and returns their pose-graph edges with the window each
// one came from (edge_window, nullable).
int vxba_hba_bottom(vxba_hba* h, const double* poses, const vxba_voxelize_params* coarse, const vxba_voxelize_params* fine, int wdsize, int mgsize, int tail,
                    int w_first, int w_stride, int n_threads, double* submap_poses, int64_t* submap_sizes, int64_t edge_capacity, int32_t* edge_ij,
                    double* edge_data, int32_t* edge_window, int64_t* n_edges) {
  if (!h || !poses || !coarse || !fine || !submap_poses || !n_edges || w_first < 0 || w_stride < 1) return fail(h, VXBA_ERR_ARG, "hba_bottom: bad argument");
  int rc = pass_geometry(h, wdsize, mgsize, tail);
  if (rc != VXBA_OK) return rc;
  const int K = h->p_K, S = h->p_S;
  std::vector<int> mine;
  for (int w = w_first; w < S; w += w_stride) mine.push_back(w);
  const int nm = (int)mine.size();
  if (n_threads <= 0) n_threads = vxhba::default_threads();
  n_threads = n_threads > vxhba::MAX_THREADS ? vxhba::MAX_THREADS : n_threads;
  if (n_threads > nm) n_threads = nm > 0 ? nm : 1;
  h->threads_used = n_threads;
  // ---- resources -------------------------------------------------------------------------------------------------------------
  size_t max_win = 0;
  for (int w : mine) max_win = std::max(max_win, (size_t)(h->sub_off[w + 1] - h->sub_off[w]));
  if (h->bottom_w != wdsize) {
    for (Worker& w : h->wk) if (w.f) { vxba_destroy(w.f); w.f = nullptr; }
    h->bottom_w = wdsize;
  }
  for (int t = 0; t < n_threads; t++) {
    Worker& w = h->wk[t];
    if (!w.s) HB(hipStreamCreateWithFlags(&w.s, hipStreamNonBlocking));
    if (!w.f) {
      if (vxba_create(wdsize, h->device, &w.f) != VXBA_OK) return fail(h, VXBA_ERR_HIP, "hba_bottom: vxba_create failed");
      vxba_set_stream(w.f, (void*)w.s);
    }
    if (max_win > w.cap) {
      if (w.d_win) hipFree(w.d_win);
      if (w.d_merged) hipFree(w.d_merged);
      w.d_win = nullptr; w.d_merged = nullptr; w.cap = 0;
      HB(hipMalloc((void**)&w.d_win, max_win * 3 * sizeof(double)));
      HB(hipMalloc((void**)&w.d_merged, max_win * 3 * sizeof(float)));
      w.cap = max_win;
    }
  }
  // ---- the windows ----------------------------------------------------------------------------------------------------------------
  std::vector<std::vector<Edge>> w_edges(S);
  auto run = [&](int t) {
    Worker& wk = h->wk[t];
    if (hipSetDevice(h->device) != hipSuccess) { wk.rc = VXBA_ERR_HIP; wk.err = "hipSetDevice"; return; }
    std::vector<double> hess;
    for (int k = t; k < nm; k += n_threads) {
      const int w = mine[k];
      int base = 0, cnt = 0;
      vxba_hba_window(K, wdsize, mgsize, tail, w, &base, &cnt);
      const int64_t p0 = h->cloud_ptr[base], n = h->cloud_ptr[base + cnt] - p0;
      int64_t fp[MAX_WD + 1], ids[MAX_WD];
      for (int i = 0; i <= cnt; i++) fp[i] = h->cloud_ptr[base + i] - p0;
      for (int i = 0; i < cnt; i++) ids[i] = base + i;
      double xs[12 * MAX_WD];
      std::memcpy(xs, poses + 12 * (size_t)base, sizeof(double) * 12 * cnt);
      const float* src = h->d_xyz + 3 * (size_t)p0;
      if (cnt >= 2) {
        // the closing window has its own size: a factor of that size, created once and kept (the only window that uses it runs on one thread)
        vxba_factor* f = wk.f;
        if (cnt != wdsize) {
          if (h->tail_f && h->tail_w != cnt) { vxba_destroy(h->tail_f); h->tail_f = nullptr; }
          if (!h->tail_f) {
            if (vxba_create(cnt, h->device, &h->tail_f) != VXBA_OK) { wk.rc = VXBA_ERR_HIP; wk.err = "hba_bottom: vxba_create (closing window) failed"; return; }
            h->tail_w = cnt;
          }
          vxba_set_stream(h->tail_f, (void*)wk.s);
          f = h->tail_f;
        }
        if (n > 0) widen_f32_kernel<<<(unsigned)((3 * n + 255) / 256), 256, 0, wk.s>>>(src, 3 * n, wk.d_win);
        wk.rc = window_refine(f, cnt, n, wk.d_win, fp, xs, coarse, fine, 1, hess, nullptr, wk.err);
        if (wk.rc != VXBA_OK) return;
        edges_from_hessian(xs, hess.data(), cnt, ids, w_edges[w]);
      }   // a closing window of ONE keyframe: nothing to refine (the gauge fixes its only pose), no pair for an edge; its cloud is the submap
      // the merged submap in the first keyframe's coordinates, voxel-filtered at voxel_size / 8 (voxelslam.cpp:2430-2450)
      FrameXf xf;
      xf.W = cnt;
      double R0[9];
      pose_R(xs, R0);
      for (int i = 0; i < cnt; i++) {
        double Ri[9];
        pose_R(xs + 12 * i, Ri);
        for (int r = 0; r < 3; r++) {
          for (int c = 0; c < 3; c++) xf.T[i][3 * r + c] = R0[r] * Ri[c] + R0[3 + r] * Ri[3 + c] + R0[6 + r] * Ri[6 + c];   // R0^T Ri
          double s = 0;
          for (int q = 0; q < 3; q++) s += R0[3 * q + r] * (xs[12 * i + 9 + q] - xs[9 + q]);
          xf.T[i][9 + r] = s;
        }
        xf.fp[i] = fp[i];
      }
      xf.fp[cnt] = fp[cnt];
      if (n > 0) merge_kernel<<<(unsigned)((n + 255) / 256), 256, 0, wk.s>>>(src, n, xf, wk.d_merged);
      int64_t kept = 0;
      const int rcd = vxd::downsample_device(wk.ds, wk.s, wk.d_merged, n, fine->voxel_size / 8, h->d_sub + 3 * (size_t)h->sub_off[w], &kept);
      if (rcd != VXBA_OK) { wk.rc = rcd; wk.err = "hba_bottom: voxel filter of a submap failed"; return; }
      h->sizes[w] = kept;
      if (submap_sizes) submap_sizes[w] = kept;
      std::memcpy(submap_poses + 12 * (size_t)w, poses + 12 * (size_t)base, sizeof(double) * 12);   // the top level starts from the INPUT anchor poses
    }
  };
  for (int t = 0; t < n_threads; t++) { h->wk[t].rc = VXBA_OK; h->wk[t].err.clear(); }
  if (nm > 0) {
    if (n_threads == 1) run(0);
    else {
      std::vector<std::thread> th;
      for (int t = 1; t < n_threads; t++) th.emplace_back(run, t);
      run(0);
      for (std::thread& x : th) x.join();
    }
  }
  // every stream drained BEFORE any worker's status is looked at: a failed pass must not leave the others' merge / filter kernels writing the
  // session's buffers while the caller retries, clears or destroys it (round-5 advisor)
  for (int t = 0; t < n_threads; t++) if (h->wk[t].s) (void)hipStreamSynchronize(h->wk[t].s);
  for (int t = 0; t < n_threads; t++) if (h->wk[t].rc != VXBA_OK) return fail(h, h->wk[t].rc, h->wk[t].err);
  // ---- edges, in window order ----------------------------------------------------------------------------------------------------
  int64_t ne = 0;
  for (int w : mine)
    for (const Edge& e : w_edges[w]) {
      if (ne < edge_capacity && edge_ij && edge_data) {
        edge_ij[2 * ne] = e.i; edge_ij[2 * ne + 1] = e.j;
        std::memcpy(edge_data + 18 * ne, e.rot, sizeof e.rot);
        std::memcpy(edge_data + 18 * ne + 9, e.tra, sizeof e.tra);
        std::memcpy(edge_data + 18 * ne + 12, e.v6, sizeof e.v6);
        if (edge_window) edge_window[ne] = w;
      }
      ne++;
    }
  *n_edges = ne;
  if (ne > edge_capacity && edge_ij) return fail(h, VXBA_ERR_ARG, "hba_bottom: more edges than edge_capacity (the count is valid: call again with room for them)");
  return VXBA_OK;
}

// The submaps of windows w_first, w_first + w_stride, .. packed back to back (window order) into d_out (device memory, float xyz): what a rank
// hands to the all-gather in front of the top level.  *n_points = points written.
int vxba_hba_export_submaps(vxba_hba* h, int w_first, int w_stride, float* d_out, int64_t capacity_points, int64_t* n_points) {
  if (!h || w_first < 0 || w_stride < 1 || !n_points || h->p_S < 1) return fail(h, VXBA_ERR_ARG, "hba_export_submaps: bad argument / no pass in progress");
  HB(hipSetDevice(h->device));
  int64_t o = 0;
  for (int w = w_first; w < h->p_S; w += w_stride) {
    if (h->sizes[w] < 0) return fail(h, VXBA_ERR_STATE, "hba_export_submaps: a window of this selection has not been run here");
    if (o + h->sizes[w] > capacity_points || (!d_out && h->sizes[w] > 0)) return fail(h, VXBA_ERR_ARG, "hba_export_submaps: buffer too small");
    if (h->sizes[w] > 0) HB(hipMemcpy(d_out + 3 * (size_t)o, h->d_sub + 3 * (size_t)h->sub_off[w], (size_t)h->sizes[w] * 3 * sizeof(float), hipMemcpyDeviceToDevice));
    o += h->sizes[w];
  }
  *n_points = o;
  return VXBA_OK;
}
// ... and the other direction: a peer's packed submaps (sizes: points per window of the pass, the entries of the selection are read) into place.
int vxba_hba_import_submaps(vxba_hba* h, int w_first, int w_stride, const int64_t* sizes, const float* d_in) {
  if (!h || w_first < 0 || w_stride < 1 || !sizes || h->p_S < 1) return fail(h, VXBA_ERR_ARG, "hba_import_submaps: bad argument / no pass in progress");
  HB(hipSetDevice(h->device));
  int64_t o = 0;
  for (int w = w_first; w < h->p_S; w += w_stride) {
    if (sizes[w] < 0 || sizes[w] > h->sub_off[w + 1] - h->sub_off[w]) return fail(h, VXBA_ERR_ARG, "hba_import_submaps: a submap larger than its window");
    if (sizes[w] > 0) {
      if (!d_in) return fail(h, VXBA_ERR_ARG, "hba_import_submaps: null points");
      HB(hipMemcpy(h->d_sub + 3 * (size_t)h->sub_off[w], d_in + 3 * (size_t)o, (size_t)sizes[w] * 3 * sizeof(float), hipMemcpyDeviceToDevice));
    }
    h->sizes[w] = sizes[w];
    o += sizes[w];
  }
  return VXBA_OK;
}

// The factor the top level runs on (created for the pass's window count): a rank of a multi-GPU pass attaches its collective to it
// (vxba_rccl_attach / vxba_peer_attach / vxba_set_allreduce) before vxba_hba_top.  Owned by the session.
int vxba_hba_top_factor(vxba_hba* h, vxba_factor** out) {
  if (!h || !out || h->p_S < 1) return fail(h, VXBA_ERR_ARG, "hba_top_factor: no pass in progress");
  HB(hipSetDevice(h->device));
  const int S = h->p_S;
  if (h->top && h->top_w != S) { vxba_destroy(h->top); h->top = nullptr; }
  if (!h->top) {
    if (vxba_create(S, h->device, &h->top) != VXBA_OK) return fail(h, VXBA_ERR_HIP, "hba_top_factor: vxba_create failed");
    h->top_w = S;
    if (!h->wk[0].s) HB(hipStreamCreateWithFlags(&h->wk[0].s, hipStreamNonBlocking));
    vxba_set_stream(h->top, (void*)h->wk[0].s);
  }
  *out = h->top;
  return VXBA_OK;
}

// Top level: ONE HBA_add_edge over all submap poses (voxelslam.cpp:2553-2560), every submap of the pass present on this device (run here or
// imported).  coarse / fine may carry a voxel shard (shard_index / shard_count): the factor then holds that shard and needs its collective.
int vxba_hba_top(vxba_hba* h, const double* poses, const vxba_voxelize_params* coarse, const vxba_voxelize_params* fine, int top_max_iter, double* submap_poses,
                 int64_t edge_capacity, int32_t* edge_ij, double* edge_data, int64_t* n_edges, double* top_rounds, int* n_top_rounds) {
  if (!h || !poses || !coarse || !fine || !submap_poses || !n_edges || h->p_S < 1) return fail(h, VXBA_ERR_ARG, "hba_top: bad argument / no pass in progress");
  const int S = h->p_S, K = h->p_K;
  if (top_max_iter < 1) top_max_iter = 1;
  vxba_factor* topf = nullptr;
  int rc = vxba_hba_top_factor(h, &topf);
  if (rc != VXBA_OK) return rc;
  std::vector<int64_t> tfp(S + 1, 0), sid(S);
  for (int w = 0; w < S; w++) {
    if (h->sizes[w] < 0) return fail(h, VXBA_ERR_STATE, "hba_top: a submap of the pass is missing on this device (vxba_hba_bottom / vxba_hba_import_submaps)");
    int base = 0, cnt = 0;
    vxba_hba_window(K, h->p_wd, h->p_mg, h->p_tail, w, &base, &cnt);
    tfp[w + 1] = tfp[w] + h->sizes[w];
    sid[w] = base;
    std::memcpy(submap_poses + 12 * (size_t)w, poses + 12 * (size_t)base, sizeof(double) * 12);
  }
  const size_t ntop = (size_t)tfp[S];
  if (ntop > h->cap_top) {
    if (h->d_top) hipFree(h->d_top);
    h->d_top = nullptr; h->cap_top = 0;
    HB(hipMalloc((void**)&h->d_top, ntop * 3 * sizeof(double)));
    h->cap_top = ntop;
  }
  hipStream_t s0 = h->wk[0].s;
  for (int w = 0; w < S; w++)
    if (h->sizes[w] > 0)
      widen_f32_kernel<<<(unsigned)((3 * h->sizes[w] + 255) / 256), 256, 0, s0>>>(h->d_sub + 3 * (size_t)h->sub_off[w], 3 * h->sizes[w], h->d_top + 3 * (size_t)tfp[w]);
  std::vector<double> hess;
  std::vector<RoundLog> log;
  std::string terr;
  int64_t ne = 0;
  if (S >= 2) {
    const int rct = window_refine(topf, S, (int64_t)ntop, h->d_top, tfp.data(), submap_poses, coarse, fine, top_max_iter, hess, &log, terr);
    if (rct != VXBA_OK) return fail(h, rct, terr);
    std::vector<Edge> e2;
    edges_from_hessian(submap_poses, hess.data(), S, sid.data(), e2);
    for (const Edge& e : e2) {
      if (ne < edge_capacity && edge_ij && edge_data) {
        edge_ij[2 * ne] = e.i; edge_ij[2 * ne + 1] = e.j;
        std::memcpy(edge_data + 18 * ne, e.rot, sizeof e.rot);
        std::memcpy(edge_data + 18 * ne + 9, e.tra, sizeof e.tra);
        std::memcpy(edge_data + 18 * ne + 12, e.v6, sizeof e.v6);
      }
      ne++;
    }
  }   // one submap: nothing to refine at the top
  *n_edges = ne;
  if (n_top_rounds) *n_top_rounds = (int)log.size();
  if (top_rounds)
    for (size_t k = 0; k < log.size() && (int)k < top_max_iter; k++) {
      top_rounds[5 * k] = (double)log[k].n_voxels; top_rounds[5 * k + 1] = log[k].r0; top_rounds[5 * k + 2] = log[k].r1;
      top_rounds[5 * k + 3] = log[k].converged; top_rounds[5 * k + 4] = log[k].fine;
    }
  if (ne > edge_capacity && edge_ij) return fail(h, VXBA_ERR_ARG, "hba_top: more edges than edge_capacity (the count is valid: call again with room for them)");
  return VXBA_OK;
}

// One whole pass on one device: vxba_hba_bottom over every window, then vxba_hba_top.
int vxba_hba_pass(vxba_hba* h, const double* poses, const vxba_voxelize_params* coarse, const vxba_voxelize_params* fine, int wdsize, int mgsize, int tail,
                  int top_max_iter, int n_threads, double* submap_poses, int64_t* submap_sizes, int64_t edge_capacity, int32_t* edge_ij, double* edge_data,
                  int64_t* n_edges1, int64_t* n_edges2, double* top_rounds, int* n_top_rounds) {
  if (!h || !n_edges1 || !n_edges2) return fail(h, VXBA_ERR_ARG, "hba_pass: null argument");
  *n_edges1 = 0; *n_edges2 = 0;
  int rc = vxba_hba_bottom(h, poses, coarse, fine, wdsize, mgsize, tail, 0, 1, n_threads, submap_poses, submap_sizes, edge_capacity, edge_ij, edge_data, nullptr, n_edges1);
  const bool short1 = rc == VXBA_ERR_ARG && *n_edges1 > edge_capacity && edge_ij;   // too many edges: the counts stay valid, go on for the top level's count
  if (rc != VXBA_OK && !short1) return rc;
  const int64_t left = edge_capacity > *n_edges1 ? edge_capacity - *n_edges1 : 0;
  rc = vxba_hba_top(h, poses, coarse, fine, top_max_iter, submap_poses, left, (edge_ij && left > 0) ? edge_ij + 2 * *n_edges1 : nullptr,
                    (edge_data && left > 0) ? edge_data + 18 * *n_edges1 : nullptr, n_edges2, top_rounds, n_top_rounds);
  if (rc != VXBA_OK && !(rc == VXBA_ERR_ARG && *n_edges2 > left)) return rc;
  if (edge_ij && *n_edges1 + *n_edges2 > edge_capacity) return fail(h, VXBA_ERR_ARG, "hba_pass: more edges than edge_capacity (the counts are valid: call again with room for them)");
  return VXBA_OK;
}

}  // extern "C"
