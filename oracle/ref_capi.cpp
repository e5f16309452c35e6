// TEST INFRASTRUCTURE ONLY.  oracle/_ref/libref.so = THE REFERENCE'S OWN hot-path code behind the same extern "C" surface as
// liboracle.so (vxo_capi.cpp), so that tests/_ref.py can drive both with one Python wrapper and tests/test_ref_pin.py can pin the
// restatement (oracle/vxo_*.hpp), the golden vectors and the HIP library to what upstream's text computes.
//
// What is compiled here, UNMODIFIED and where it lies (never copied into this repository):
//     /root/reference/VoxelSLAM/src/tools.hpp            Exp/Log/hat/jr/jr_inv, IMUST, PointCluster
//     /root/reference/VoxelSLAM/src/preintegration.hpp   IMU_PRE
//     /root/reference/VoxelSLAM/src/voxel_map.hpp        Bf_var, LidarFactor, Lidar_BA_Optimizer, LI_BA_Optimizer,
//                                                        LI_BA_OptimizerGravity, SlideWindow, OctoTree, cut_voxel_multi, match
//     /root/reference/VoxelSLAM/src/loop_refine.hpp      OctreeGBA (cut_voxel, subdivide, recut), OctreeGBA_multi_recut
// against either a real Eigen (when the Makefile finds one) or the API shim in oracle/shim/ (Eigen/PCL/ROS are absent from this
// image; see oracle/shim/Eigen/Core for exactly what the shim is and is not).  This file adds only: packing between the flat f64
// formats of include/vxba.h and the reference's structs; the two map drivers `multi_recut` / `multi_margi`, which upstream keeps as
// members of the ROS node class (voxelslam.cpp:1321-1453, not compilable) and which are therefore restated here in a dozen lines
// each; and a printf hook that captures, at full precision, the LM trace `Lidar_BA_Optimizer::damping_iter` prints with
// is_display = true (voxel_map.hpp:415-416).
//
// Differences from liboracle.so that the tests account for: the LI optimizers' thread count is hard-wired to 5 upstream (the
// thd_num argument is ignored); factor voxels come out of `tras_opt` in unordered_map order (matched by node id in the tests);
// `Lidar_BA_Optimizer::only_residual` exit(0)s when there are fewer voxels than threads -- do not call it that way.
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>   // AngleAxisd: tools.hpp uses it and relies on PCL to have pulled the header in

namespace vxref {
struct TraceRow { double residual1, residual2, u, v, q, q1; };
static thread_local std::vector<TraceRow>* g_trace = nullptr;
static int hook_printf(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  int r = 0;
  if (g_trace && std::strncmp(fmt, "iter%d:", 7) == 0) {   // voxel_map.hpp:416: i, residual1, residual2, u, v, q/q1, q1, q
    (void)va_arg(ap, int);
    TraceRow t;
    t.residual1 = va_arg(ap, double); t.residual2 = va_arg(ap, double); t.u = va_arg(ap, double); t.v = va_arg(ap, double);
    (void)va_arg(ap, double);
    t.q1 = va_arg(ap, double); t.q = va_arg(ap, double);
    g_trace->push_back(t);
  } else {
    r = std::vfprintf(stderr, fmt, ap);
  }
  va_end(ap);
  return r;
}
}  // namespace vxref

#define printf(...) vxref::hook_printf(__VA_ARGS__)
#include "tools.hpp"
#include "preintegration.hpp"
#include "voxel_map.hpp"
#include "loop_refine.hpp"
#undef printf

namespace {

using Eigen::Matrix3d;
using Eigen::MatrixXd;
using Eigen::Vector3d;
using Eigen::VectorXd;

PointCluster unpack_cluster(const double* c) {
  PointCluster pc;
  pc.P(0, 0) = c[0]; pc.P(0, 1) = pc.P(1, 0) = c[1]; pc.P(0, 2) = pc.P(2, 0) = c[2];
  pc.P(1, 1) = c[3]; pc.P(1, 2) = pc.P(2, 1) = c[4]; pc.P(2, 2) = c[5];
  pc.v = Vector3d(c[6], c[7], c[8]);
  pc.N = (int)c[9];
  return pc;
}
void pack_cluster(const PointCluster& pc, double* c) {
  c[0] = pc.P(0, 0); c[1] = pc.P(0, 1); c[2] = pc.P(0, 2); c[3] = pc.P(1, 1); c[4] = pc.P(1, 2); c[5] = pc.P(2, 2);
  c[6] = pc.v[0]; c[7] = pc.v[1]; c[8] = pc.v[2]; c[9] = (double)pc.N;
}
Matrix3d unpack_m3(const double* m) { Matrix3d r; for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) r(rr, c) = m[3 * c + rr]; return r; }
void pack_m3(const Matrix3d& a, double* m) { for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) m[3 * c + r] = a(r, c); }
Vector3d unpack_v3(const double* a) { return Vector3d(a[0], a[1], a[2]); }
void pack_v3(const Vector3d& a, double* o) { for (int k = 0; k < 3; k++) o[k] = a[k]; }
template <typename M> void pack_mat(const M& a, double* o) { for (int c = 0; c < a.cols(); c++) for (int r = 0; r < a.rows(); r++) o[(size_t)c * a.rows() + r] = a(r, c); }

std::vector<IMUST> unpack_poses(const double* Rp, int W) {
  std::vector<IMUST> xs(W);
  for (int i = 0; i < W; i++) { xs[i].R = unpack_m3(Rp + 12 * i); xs[i].p = unpack_v3(Rp + 12 * i + 9); }
  return xs;
}
void pack_poses(const std::vector<IMUST>& xs, double* Rp) {
  for (size_t i = 0; i < xs.size(); i++) { pack_m3(xs[i].R, Rp + 12 * i); pack_v3(xs[i].p, Rp + 12 * i + 9); }
}

// flat formats of include/vxba.h (identical to vxo_capi.cpp):  state 24 f64 [R | p | v | bg | ba | g];  imu 304 f64
constexpr int ST = 24, IM = 304;
IMUST unpack_state(const double* s) {
  IMUST x;
  x.R = unpack_m3(s); x.p = unpack_v3(s + 9); x.v = unpack_v3(s + 12); x.bg = unpack_v3(s + 15); x.ba = unpack_v3(s + 18); x.g = unpack_v3(s + 21);
  return x;
}
void pack_state(const IMUST& x, double* s) {
  pack_m3(x.R, s); pack_v3(x.p, s + 9); pack_v3(x.v, s + 12); pack_v3(x.bg, s + 15); pack_v3(x.ba, s + 18); pack_v3(x.g, s + 21);
}
void unpack_imu(const double* b, IMU_PRE& f) {
  f.R_delta = unpack_m3(b);
  f.p_delta = unpack_v3(b + 9); f.v_delta = unpack_v3(b + 12); f.bg = unpack_v3(b + 15); f.ba = unpack_v3(b + 18);
  f.R_bg = unpack_m3(b + 21); f.p_bg = unpack_m3(b + 30); f.p_ba = unpack_m3(b + 39); f.v_bg = unpack_m3(b + 48); f.v_ba = unpack_m3(b + 57);
  f.dtime = b[66];
  f.dbg = unpack_v3(b + 67); f.dba = unpack_v3(b + 70); f.dbg_buf = unpack_v3(b + 73); f.dba_buf = unpack_v3(b + 76);
  for (int c = 0; c < 15; c++) for (int r = 0; r < 15; r++) f.cov(r, c) = b[79 + 15 * c + r];
}
void pack_imu(const IMU_PRE& f, double* b) {
  pack_m3(f.R_delta, b);
  pack_v3(f.p_delta, b + 9); pack_v3(f.v_delta, b + 12); pack_v3(f.bg, b + 15); pack_v3(f.ba, b + 18);
  pack_m3(f.R_bg, b + 21); pack_m3(f.p_bg, b + 30); pack_m3(f.p_ba, b + 39); pack_m3(f.v_bg, b + 48); pack_m3(f.v_ba, b + 57);
  b[66] = f.dtime;
  pack_v3(f.dbg, b + 67); pack_v3(f.dba, b + 70); pack_v3(f.dbg_buf, b + 73); pack_v3(f.dba_buf, b + 76);
  for (int c = 0; c < 15; c++) for (int r = 0; r < 15; r++) b[79 + 15 * c + r] = f.cov(r, c);
}
Eigen::Matrix<double, 6, 6> unpack_m6(const double* a) { Eigen::Matrix<double, 6, 6> m; for (int c = 0; c < 6; c++) for (int r = 0; r < 6; r++) m(r, c) = a[6 * c + r]; return m; }

struct Handle {
  LidarFactor factor;
  explicit Handle(int w) : factor(w) {}
};

struct LiCtx {
  std::vector<IMUST> xs;
  std::deque<IMU_PRE*> imus;
  LiCtx(const double* states, const double* blobs, int W) {
    xs.resize(W);
    for (int i = 0; i < W; i++) xs[i] = unpack_state(states + ST * i);
    for (int i = 0; i < W - 1; i++) { IMU_PRE* f = new IMU_PRE(); unpack_imu(blobs + IM * i, *f); imus.push_back(f); }
  }
  ~LiCtx() { for (IMU_PRE* f : imus) delete f; }
  void pack(double* states, double* blobs) {
    for (size_t i = 0; i < xs.size(); i++) pack_state(xs[i], states + ST * i);
    for (size_t i = 0; i < imus.size(); i++) pack_imu(*imus[i], blobs + IM * i);
  }
};

void write_trace(const std::vector<vxref::TraceRow>& tr, double* trace_out, int* n_trace) {
  if (n_trace) *n_trace = (int)tr.size();
  if (!trace_out) return;
  for (size_t i = 0; i < tr.size(); i++) {
    double* o = trace_out + 8 * i;
    o[0] = tr[i].residual1; o[1] = tr[i].residual2; o[2] = tr[i].u; o[3] = tr[i].v; o[4] = tr[i].q; o[5] = tr[i].q1;
    o[6] = tr[i].q > 0 ? 1.0 : 0.0;                                   // accepted  (voxel_map.hpp:418)
    o[7] = (i == 0 || tr[i - 1].q > 0) ? 1.0 : 0.0;                   // is_calc_hess at the top of this iteration (:386, :433)
  }
}

}  // namespace

extern "C" {

const char* vxo_backend() {
#if defined(VXREF_DROPIN)
  return "reference OctoTree / OctreeGBA / call sites + vxba_voxel_map.hpp (LidarFactor and the three optimizers on libvxba.so)";
#elif defined(VXO_EIGEN_SHIM)
  return "reference headers (unmodified) + Eigen API shim";
#else
  return "reference headers (unmodified) + real Eigen";
#endif
}

void* vxo_create(int win_size) { return new Handle(win_size); }
void vxo_destroy(void* h) { delete (Handle*)h; }
void vxo_clear(void* h) { ((Handle*)h)->factor.clear(); }
int vxo_size(void* h) { return (int)((Handle*)h)->factor.plvec_voxels.size(); }

void vxo_push_voxels(void* h, int n, const double* clusters, const double* fix, const double* coe, const double* eig_val,
                     const double* eig_vec, const double* merged) {
  LidarFactor& f = ((Handle*)h)->factor;
  const int W = f.win_size;
  for (int a = 0; a < n; a++) {
    std::vector<PointCluster> vec(W);
    for (int i = 0; i < W; i++) vec[i] = unpack_cluster(clusters + ((size_t)a * W + i) * 10);
    PointCluster fx = unpack_cluster(fix + (size_t)a * 10), mg = unpack_cluster(merged + (size_t)a * 10);
    Vector3d ev(eig_val[3 * a], eig_val[3 * a + 1], eig_val[3 * a + 2]);
    Matrix3d U = unpack_m3(eig_vec + 9 * (size_t)a);
    f.push_voxel(vec, fx, coe[a], ev, U, mg);
  }
}

void vxo_acc_evaluate2(void* h, const double* Rp, int head, int end, double* Hess, double* JacT, double* residual) {
  LidarFactor& f = ((Handle*)h)->factor;
  const int n = 6 * f.win_size;
  MatrixXd H(n, n);
  VectorXd J(n);
  f.acc_evaluate2(unpack_poses(Rp, f.win_size), head, end, H, J, *residual);
  pack_mat(H, Hess);
  pack_mat(J, JacT);
}

void vxo_evaluate_only_residual(void* h, const double* Rp, int head, int end, double* residual) {
  LidarFactor& f = ((Handle*)h)->factor;
  f.evaluate_only_residual(unpack_poses(Rp, f.win_size), head, end, *residual);
}

void vxo_read_cache(void* h, int head, int end, double* eig_val, double* eig_vec, double* merged) {
  LidarFactor& f = ((Handle*)h)->factor;
  for (int a = head; a < end; a++) {
    size_t k = a - head;
    for (int j = 0; j < 3; j++) eig_val[3 * k + j] = f.eig_values[a][j];
    pack_m3(f.eig_vectors[a], eig_vec + 9 * k);
    pack_cluster(f.pcr_adds[a], merged + 10 * k);
  }
}

double vxo_divide_thread(void* h, const double* Rp, int thd_num, double* Hess, double* JacT) {
  LidarFactor& f = ((Handle*)h)->factor;
  Lidar_BA_Optimizer opt;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.thd_num = thd_num;
  MatrixXd H(opt.jac_leng, opt.jac_leng);
  VectorXd J(opt.jac_leng);
  std::vector<IMUST> xs = unpack_poses(Rp, f.win_size);
  double r = opt.divide_thread(xs, f, H, J);
  if (Hess) pack_mat(H, Hess);
  if (JacT) pack_mat(J, JacT);
  return r;
}
double vxo_only_residual(void* h, const double* Rp, int thd_num) {
  LidarFactor& f = ((Handle*)h)->factor;
  Lidar_BA_Optimizer opt;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.thd_num = thd_num;
  std::vector<IMUST> xs = unpack_poses(Rp, f.win_size);
  return opt.only_residual(xs, f);
}

// Lidar_BA_Optimizer::damping_iter (voxel_map.hpp:367-442), is_display = true so that the hook sees every iteration
int vxo_damping_iter(void* h, double* Rp, int thd_num, int max_iter, double* hess_out, double* resis_out, double* trace_out, int* n_trace) {
  LidarFactor& f = ((Handle*)h)->factor;
  Lidar_BA_Optimizer opt;
  opt.thd_num = thd_num;
  std::vector<IMUST> xs = unpack_poses(Rp, f.win_size);
  MatrixXd hess;
  std::vector<double> resis;
  std::vector<vxref::TraceRow> tr;
  vxref::g_trace = &tr;
  bool conv = opt.damping_iter(xs, f, &hess, resis, max_iter, true);
  vxref::g_trace = nullptr;
  pack_poses(xs, Rp);
  if (hess_out) pack_mat(hess, hess_out);
  if (resis_out) { resis_out[0] = resis[0]; resis_out[1] = resis[1]; }
  write_trace(tr, trace_out, n_trace);
  return conv ? 1 : 0;
}

// The cpu_baseline leg: one accepted-step BA iteration built from the reference's own members -- divide_thread (:298-335), the
// gauge fix / damped LDLT solve / state update exactly as damping_iter spells them (:397-410), only_residual (:337-365) -- timed,
// `iters` times at fixed poses.  Returns the median seconds per iteration.
double vxo_time_ba_iteration(void* h, const double* Rp, int thd_num, int warmup, int iters, double* hess_s, double* resid_s) {
  LidarFactor& f = ((Handle*)h)->factor;
  Lidar_BA_Optimizer opt;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.thd_num = thd_num;
  const int n = opt.jac_leng, win_size = f.win_size;
  std::vector<IMUST> x_stats = unpack_poses(Rp, f.win_size), x_stats_temp = x_stats;
  MatrixXd D(n, n), Hess(n, n);
  VectorXd JacT(n), dxi(n);
  D.setIdentity();
  const double u = 0.01;
  std::vector<double> tot, th, tr;
  for (int it = 0; it < warmup + iters; it++) {
    auto t0 = std::chrono::steady_clock::now();
    opt.divide_thread(x_stats, f, Hess, JacT);
    auto t1 = std::chrono::steady_clock::now();
    Hess.topRows(6).setZero();
    Hess.leftCols(6).setZero();
    Hess.block<6, 6>(0, 0).setIdentity();
    JacT.head(6).setZero();
    D.diagonal() = Hess.diagonal();
    dxi = (Hess + u * D).ldlt().solve(-JacT);
    for (int j = 0; j < win_size; j++) {
      x_stats_temp[j].R = x_stats[j].R * Exp(dxi.block<3, 1>(6 * j, 0));
      x_stats_temp[j].p = x_stats[j].p + dxi.block<3, 1>(6 * j + 3, 0);
    }
    auto t2 = std::chrono::steady_clock::now();
    opt.only_residual(x_stats_temp, f);
    auto t3 = std::chrono::steady_clock::now();
    if (it >= warmup) {
      tot.push_back(std::chrono::duration<double>(t3 - t0).count());
      th.push_back(std::chrono::duration<double>(t1 - t0).count());
      tr.push_back(std::chrono::duration<double>(t3 - t2).count());
    }
  }
  opt.only_residual(x_stats, f);
  auto median = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  if (hess_s) *hess_s = median(th);
  if (resid_s) *resid_s = median(tr);
  return median(tot);
}

// ---- building blocks ----------------------------------------------------------------------------------------------------------
void vxo_eig_sym3(const double* C_colmajor, double* val, double* vec_colmajor) {
  Eigen::SelfAdjointEigenSolver<Matrix3d> saes(unpack_m3(C_colmajor));
  Vector3d l = saes.eigenvalues();
  Matrix3d U = saes.eigenvectors();
  for (int i = 0; i < 3; i++) val[i] = l[i];
  pack_m3(U, vec_colmajor);
}
void vxo_exp(const double* ang, double* R_colmajor) { pack_m3(Exp(unpack_v3(ang)), R_colmajor); }
void vxo_log(const double* R_colmajor, double* ang) { pack_v3(Log(unpack_m3(R_colmajor)), ang); }
void vxo_ldlt_solve(int n, const double* A_colmajor, const double* b, double* x) {
  MatrixXd A(n, n);
  VectorXd rhs(n);
  for (int c = 0; c < n; c++) for (int r = 0; r < n; r++) A(r, c) = A_colmajor[(size_t)c * n + r];
  for (int r = 0; r < n; r++) rhs(r) = b[r];
  VectorXd s = A.ldlt().solve(rhs);
  for (int r = 0; r < n; r++) x[r] = s(r);
}
void vxo_cluster_transform(const double* cluster, const double* Rp, double* out) {
  PointCluster o;
  o.transform(unpack_cluster(cluster), unpack_poses(Rp, 1)[0]);
  pack_cluster(o, out);
}
void vxo_build_clusters(int64_t n_cells, const int64_t* cell_ptr, const double* xyz, double* out) {
  for (int64_t c = 0; c < n_cells; c++) {
    PointCluster pc;
    for (int64_t k = cell_ptr[c]; k < cell_ptr[c + 1]; k++) pc.push(Vector3d(xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]));
    pack_cluster(pc, out + 10 * c);
  }
}
// eig(pcr.cov()) as OctoTree::recut does it (voxel_map.hpp:1161-1163)
void vxo_plane_fit(int64_t n, const double* clusters, double* eig_val, double* eig_vec) {
  for (int64_t a = 0; a < n; a++) {
    PointCluster pc = unpack_cluster(clusters + 10 * a);
    Eigen::SelfAdjointEigenSolver<Matrix3d> saes(pc.cov());
    Vector3d l = saes.eigenvalues();
    Matrix3d U = saes.eigenvectors();
    for (int i = 0; i < 3; i++) eig_val[3 * a + i] = l[i];
    pack_m3(U, eig_vec + 9 * a);
  }
}

// ---- inertial half ---------------------------------------------------------------------------------------------------------------
void vxo_jr(const double* vec, double* out_colmajor) { pack_m3(jr(unpack_v3(vec)), out_colmajor); }
void vxo_jr_inv(const double* R_colmajor, double* out_colmajor) { pack_m3(jr_inv(unpack_m3(R_colmajor)), out_colmajor); }
void vxo_mat_inverse(int n, const double* A_colmajor, double* out) {
  MatrixXd A(n, n);
  for (int c = 0; c < n; c++) for (int r = 0; r < n; r++) A(r, c) = A_colmajor[(size_t)c * n + r];
  MatrixXd inv = A.inverse();
  pack_mat(inv, out);
}
void vxo_imu_init(double* blob, const double* bg, const double* ba) { IMU_PRE f(unpack_v3(bg), unpack_v3(ba)); pack_imu(f, blob); }
void vxo_imu_add(double* blob, const double* gyr, const double* acc, double dt, const double* noiseMeas36, const double* noiseWalk36) {
  noiseMeas = unpack_m6(noiseMeas36);     // the reference's globals (preintegration.hpp:9)
  noiseWalk = unpack_m6(noiseWalk36);
  IMU_PRE f;
  unpack_imu(blob, f);
  Vector3d g = unpack_v3(gyr), a = unpack_v3(acc);
  f.add_imu(g, a, dt);
  pack_imu(f, blob);
}
double vxo_imu_evaluate(const double* blob, const double* st1, const double* st2, double* jtj, double* gg, int jac_enable) {
  IMU_PRE f;
  unpack_imu(blob, f);
  MatrixXd J(2 * DIM, 2 * DIM);
  VectorXd g(2 * DIM);
  J.setZero(); g.setZero();
  IMUST a = unpack_state(st1), b = unpack_state(st2);
  const double r = f.give_evaluate(a, b, J, g, jac_enable != 0);
  if (jac_enable) { if (jtj) pack_mat(J, jtj); if (gg) pack_mat(g, gg); }
  return r;
}
double vxo_imu_evaluate_g(const double* blob, const double* st1, const double* st2, double* jtj, double* gg, int jac_enable) {
  IMU_PRE f;
  unpack_imu(blob, f);
  MatrixXd J(2 * DIM + 3, 2 * DIM + 3);
  VectorXd g(2 * DIM + 3);
  J.setZero(); g.setZero();
  IMUST a = unpack_state(st1), b = unpack_state(st2);
  const double r = f.give_evaluate_g(a, b, J, g, jac_enable != 0);
  if (jac_enable) { if (jtj) pack_mat(J, jtj); if (gg) pack_mat(g, gg); }
  return r;
}
// LI_BA_Optimizer (voxel_map.hpp:446-655).  thd_num is hard-wired to 5 upstream; the argument is ignored.
double vxo_li_divide_thread(void* h, const double* states, const double* blobs, int /*thd_num*/, double imu_coef_, double* Hess, double* JacT) {
  LidarFactor& f = ((Handle*)h)->factor;
  imu_coef = imu_coef_;
  LiCtx c(states, blobs, f.win_size);
  LI_BA_Optimizer opt;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.imu_leng = DIM * f.win_size;
  MatrixXd H(opt.imu_leng, opt.imu_leng);
  VectorXd J(opt.imu_leng);
  const double r = opt.divide_thread(c.xs, f, c.imus, H, J);
  pack_mat(H, Hess);
  pack_mat(J, JacT);
  return r;
}
double vxo_li_only_residual(void* h, const double* states, const double* blobs, int /*thd_num*/, double imu_coef_) {
  LidarFactor& f = ((Handle*)h)->factor;
  imu_coef = imu_coef_;
  LiCtx c(states, blobs, f.win_size);
  LI_BA_Optimizer opt;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.imu_leng = DIM * f.win_size;
  return opt.only_residual(c.xs, f, c.imus);
}
// LI_BA_Optimizer::damping_iter runs exactly 3 iterations upstream (:579); max_iter must be 3.  No trace is printed upstream
// (the printf at :620 is commented out): n_trace = 0.
void vxo_li_damping_iter(void* h, double* states, double* blobs, int /*thd_num*/, double imu_coef_, int max_iter, double* hess_out,
                         double* /*trace_out*/, int* n_trace) {
  LidarFactor& f = ((Handle*)h)->factor;
  if (max_iter != 3) { std::fprintf(stderr, "libref: LI_BA_Optimizer::damping_iter always runs 3 iterations\n"); std::abort(); }
  imu_coef = imu_coef_;
  LiCtx c(states, blobs, f.win_size);
  LI_BA_Optimizer opt;
  MatrixXd hess;
  opt.damping_iter(c.xs, f, c.imus, &hess);
  c.pack(states, blobs);
  if (hess_out) pack_mat(hess, hess_out);
  if (n_trace) *n_trace = 0;
}
// LI_BA_OptimizerGravity::divide_thread / only_residual (voxel_map.hpp:673-773), the reference's own members
double vxo_li_divide_thread_gravity(void* h, const double* states, const double* blobs, int /*thd_num*/, double imu_coef_, double* Hess, double* JacT) {
  LidarFactor& f = ((Handle*)h)->factor;
  imu_coef = imu_coef_;
  LiCtx c(states, blobs, f.win_size);
  LI_BA_OptimizerGravity opt;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.imu_leng = DIM * f.win_size + 3;
  MatrixXd H(opt.imu_leng, opt.imu_leng);
  VectorXd J(opt.imu_leng);
  const double r = opt.divide_thread(c.xs, f, c.imus, H, J);
  pack_mat(H, Hess);
  pack_mat(J, JacT);
  return r;
}
double vxo_li_only_residual_gravity(void* h, const double* states, const double* blobs, int /*thd_num*/, double imu_coef_) {
  LidarFactor& f = ((Handle*)h)->factor;
  imu_coef = imu_coef_;
  LiCtx c(states, blobs, f.win_size);
  LI_BA_OptimizerGravity opt;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.imu_leng = DIM * f.win_size + 3;
  return opt.only_residual(c.xs, f, c.imus);
}
// LI_BA_OptimizerGravity::damping_iter (:775-862).  `hess` is never resized upstream (:775-790): the caller's matrix must already
// have the (15W+3)^2 shape, as voxelslam.cpp's caller provides.
void vxo_li_damping_iter_gravity(void* h, double* states, double* blobs, int /*thd_num*/, double imu_coef_, int max_iter, double* hess_out,
                                 double* resis_out, double* /*trace_out*/, int* n_trace) {
  LidarFactor& f = ((Handle*)h)->factor;
  imu_coef = imu_coef_;
  LiCtx c(states, blobs, f.win_size);
  LI_BA_OptimizerGravity opt;
  const int n = DIM * f.win_size + 3;
  MatrixXd hess(n, n);
  std::vector<double> resis;
  opt.damping_iter(c.xs, f, c.imus, resis, &hess, max_iter);
  c.pack(states, blobs);
  if (hess_out) pack_mat(hess, hess_out);
  if (resis_out && resis.size() >= 2) { resis_out[0] = resis[0]; resis_out[1] = resis[1]; }
  if (n_trace) *n_trace = 0;
}

// ---- per-leaf plane producers ----------------------------------------------------------------------------------------------------
// Sum of Bf_var over the points of each cell (voxel_map.hpp:91-106 as OctoTree::push uses it, :990-992)
void vxo_cov_add_build(int64_t n_cells, const int64_t* cell_ptr, const double* xyz_world, const double* var9, double* cov_add) {
  for (int64_t c = 0; c < n_cells; c++) {
    Eigen::Matrix<double, 9, 9> acc, Bi;
    acc.setZero();
    for (int64_t q = cell_ptr[c]; q < cell_ptr[c + 1]; q++) {
      pointVar pv;
      pv.pnt = unpack_v3(xyz_world + 3 * q);
      pv.var = unpack_m3(var9 + 9 * q);
      Bf_var(pv, Bi, pv.pnt);
      acc += Bi;
    }
    pack_mat(acc, cov_add + 81 * c);
  }
}
// OctoTree::plane_update (voxel_map.hpp:1118-1146) on a free-standing node
void vxo_plane_update(int64_t n, const double* clusters, const double* eig_val, const double* eig_vec, const double* cov_add, double* center, double* normal,
                      double* plane_var, double* radius) {
  for (int64_t a = 0; a < n; a++) {
    OctoTree ot(0, 1);
    ot.pcr_add = unpack_cluster(clusters + 10 * a);
    ot.eig_value = unpack_v3(eig_val + 3 * a);
    ot.eig_vector = unpack_m3(eig_vec + 9 * a);
    for (int c = 0; c < 9; c++) for (int r = 0; r < 9; r++) ot.cov_add(r, c) = cov_add[81 * a + 9 * c + r];
    ot.plane_update();
    pack_v3(ot.plane.center, center + 3 * a);
    pack_v3(ot.plane.normal, normal + 3 * a);
    pack_mat(ot.plane.plane_var, plane_var + 36 * a);
    radius[a] = ot.plane.radius;
  }
}

// ---- incremental local map: the reference's OctoTree / cut_voxel_multi driven the way the local-mapping thread drives them ----------
namespace {
struct RefLocalMap {
  int win_size, thread_num;
  double p_voxel_size, p_min_eigen_value;
  int p_max_layer, p_max_points;
  double p_min_point[4], p_thre[4];
  std::vector<int> mp_store;
  std::unordered_map<VOXEL_LOC, OctoTree*> surf_map, surf_map_slide;
  std::vector<std::vector<SlideWindow*>> sws;

  // the reference keeps its map parameters and the slot ring in globals (voxel_map.hpp:83-89, 933)
  void bind() {
    voxel_size = p_voxel_size; max_layer = p_max_layer; max_points = p_max_points; min_eigen_value = p_min_eigen_value;
    for (int k = 0; k < 4; k++) min_point[k] = p_min_point[k];
    plane_eigen_value_thre.assign(p_thre, p_thre + 4);
    mp = mp_store.data();
  }
};
void free_tree(OctoTree* ot) {
  std::vector<OctoTree*> rel;
  ot->tras_ptr(rel);                       // the reference's own release walk (voxelslam.cpp:589-598)
  std::vector<SlideWindow*> sink;
  ot->clear_slwd(sink);
  for (OctoTree* o : rel) delete o;
  for (SlideWindow* s : sink) delete s;
  delete ot;
}
struct RefLeaf { uint64_t id; const OctoTree* node; bool in_slide; };
void walk_leaves(const OctoTree* n, uint64_t root48, uint64_t path, bool in_slide, std::vector<RefLeaf>& out) {
  if (n->octo_state == 0) { out.push_back(RefLeaf{(root48 << 16) | (path << 7) | (uint64_t)n->layer, n, in_slide}); return; }
  for (int i = 0; i < 8; i++)
    if (n->leaves[i] != nullptr) walk_leaves(n->leaves[i], root48, path | ((uint64_t)i << (3 * (2 - n->layer))), in_slide, out);
}
bool all_leaves(RefLocalMap* lm, std::vector<RefLeaf>& out) {
  bool in_range = true;
  for (auto& kv : lm->surf_map) {
    const VOXEL_LOC& k = kv.first;
    if (k.x < -32768 || k.x > 32767 || k.y < -32768 || k.y > 32767 || k.z < -32768 || k.z > 32767) { in_range = false; continue; }
    const uint64_t root48 = ((uint64_t)(k.x + 32768) << 32) | ((uint64_t)(k.y + 32768) << 16) | (uint64_t)(k.z + 32768);
    walk_leaves(kv.second, root48, 0, lm->surf_map_slide.count(k) != 0, out);
  }
  std::sort(out.begin(), out.end(), [](const RefLeaf& a, const RefLeaf& b) { return a.id < b.id; });
  return in_range;
}
}  // namespace

// params = [voxel_size, max_layer, min_point[4], min_eigen_value, plane_eigen_value_thre[4], max_points, win_size, thread_num]
void* vxo_localmap_create(const double* p) {
  RefLocalMap* lm = new RefLocalMap();
  lm->p_voxel_size = p[0]; lm->p_max_layer = (int)p[1];
  for (int k = 0; k < 4; k++) lm->p_min_point[k] = p[2 + k];
  lm->p_min_eigen_value = p[6];
  for (int k = 0; k < 4; k++) lm->p_thre[k] = p[7 + k];
  lm->p_max_points = (int)p[11]; lm->win_size = (int)p[12]; lm->thread_num = (int)p[13];
  lm->mp_store.resize(lm->win_size);
  for (int i = 0; i < lm->win_size; i++) lm->mp_store[i] = i;      // voxelslam.cpp:1312-1313
  lm->sws.resize(lm->thread_num);                                   // voxelslam.cpp:1470
  return lm;
}
void vxo_localmap_destroy(void* m) {
  RefLocalMap* lm = (RefLocalMap*)m;
  lm->bind();
  for (auto& kv : lm->surf_map) free_tree(kv.second);
  for (auto& v : lm->sws) for (SlideWindow* s : v) delete s;
  delete lm;
}
// cut_voxel_multi(surf_map, pvec, win_count-1, surf_map_slide, win_size, pwld, sws)   (voxelslam.cpp:1609)
void vxo_localmap_cut_voxel(void* m, int ord, int64_t n, const double* pnt, const double* var, const double* pwld_in) {
  RefLocalMap* lm = (RefLocalMap*)m;
  lm->bind();
  PVecPtr pvec(new PVec((size_t)n));
  PLV(3) pwld((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    (*pvec)[i].pnt = unpack_v3(pnt + 3 * i);
    (*pvec)[i].var = unpack_m3(var + 9 * i);
    pwld[i] = unpack_v3(pwld_in + 3 * i);
  }
  cut_voxel_multi(lm->surf_map, pvec, ord, lm->surf_map_slide, lm->win_size, pwld, lm->sws);
}
// multi_recut (voxelslam.cpp:1396-1453), restated: it is a member of the ROS node class.  Same partition of the slide map over
// `thread_num` lists, same early return, every list's recut() run with its own SlideWindow pool, pools merged, then tras_opt over the
// slide map in its hash order.  (The lists are processed one after another here instead of on threads -- they touch disjoint trees.)
void vxo_localmap_recut(void* m, int win_count, const double* Rp, void* factor) {
  RefLocalMap* lm = (RefLocalMap*)m;
  lm->bind();
  LidarFactor& voxopt = ((Handle*)factor)->factor;
  std::vector<IMUST> xs = unpack_poses(Rp, win_count);
  auto& feat_map = lm->surf_map_slide;
  auto& sws = lm->sws;
  const int thd_num = lm->thread_num;
  std::vector<std::vector<OctoTree*>> octss(thd_num);
  const int g_size = (int)feat_map.size();
  if (g_size < thd_num) return;
  const double part = 1.0 * g_size / thd_num;
  int cnt = 0;
  for (auto iter = feat_map.begin(); iter != feat_map.end(); iter++) {
    octss[cnt].push_back(iter->second);
    if (octss[cnt].size() >= part && cnt < thd_num - 1) cnt++;
  }
  for (int i = 0; i < thd_num; i++)
    for (OctoTree* oc : octss[i]) oc->recut(win_count, xs, sws[i]);
  for (size_t i = 1; i < sws.size(); i++) {
    sws[0].insert(sws[0].end(), sws[i].begin(), sws[i].end());
    sws[i].clear();
  }
  for (auto iter = feat_map.begin(); iter != feat_map.end(); iter++) iter->second->tras_opt(voxopt);
}
// multi_margi (voxelslam.cpp:1321-1394), restated for the same reason: margi(win_count, 1, xs, voxopt) on every root of the slide map
// (early return when there are fewer roots than threads), then roots without live content leave the slide map.
int vxo_localmap_margi(void* m, int win_count, const double* Rp, void* factor) {
  RefLocalMap* lm = (RefLocalMap*)m;
  lm->bind();
  LidarFactor& voxopt = ((Handle*)factor)->factor;
  std::vector<IMUST> xs = unpack_poses(Rp, win_count);
  auto& feat_map = lm->surf_map_slide;
  const int g_size = (int)feat_map.size();
  if (g_size < lm->thread_num) return 0;
  for (auto iter = feat_map.begin(); iter != feat_map.end(); iter++) {
    if (iter->second->opt_state >= int(voxopt.pcr_adds.size())) return -1;     // upstream: printf + exit(0) inside margi (:1211-1215)
    iter->second->margi(win_count, 1, xs, voxopt);
  }
  for (auto iter = feat_map.begin(); iter != feat_map.end();) {
    if (iter->second->isexist) iter++;
    else {
      iter->second->clear_slwd(lm->sws[0]);
      feat_map.erase(iter++);
    }
  }
  return 0;
}
// voxelslam.cpp:1683-1687
void vxo_localmap_slide(void* m, int mgsize) {
  RefLocalMap* lm = (RefLocalMap*)m;
  for (int i = 0; i < lm->win_size; i++) {
    lm->mp_store[i] += mgsize;
    if (lm->mp_store[i] >= lm->win_size) lm->mp_store[i] -= lm->win_size;
  }
}
void vxo_localmap_counts(void* m, int64_t* out) {
  RefLocalMap* lm = (RefLocalMap*)m;
  std::vector<RefLeaf> lv;
  all_leaves(lm, lv);
  out[0] = (int64_t)lm->surf_map.size(); out[1] = (int64_t)lm->surf_map_slide.size(); out[2] = (int64_t)lv.size(); out[3] = lm->mp_store[0];
}
// same record layout as liboracle's vxo_localmap_leaves; leaves in ascending node id
int64_t vxo_localmap_leaves(void* m, int64_t capacity, uint64_t* ids, int32_t* ints, double* dbl) {
  RefLocalMap* lm = (RefLocalMap*)m;
  std::vector<RefLeaf> lv;
  if (!all_leaves(lm, lv)) return -1;
  const int W = lm->win_size;
  const size_t rec = 156 + 11 * (size_t)W;
  for (int64_t a = 0; a < (int64_t)lv.size() && a < capacity; a++) {
    const OctoTree* n = lv[a].node;
    ids[a] = lv[a].id;
    int32_t* I = ints + 8 * a;
    I[0] = n->layer; I[1] = n->isexist; I[2] = n->plane.is_plane; I[3] = n->sw != nullptr; I[4] = n->opt_state; I[5] = n->last_num;
    I[6] = (int32_t)n->point_fix.size(); I[7] = lv[a].in_slide;
    double* D = dbl + rec * a;
    pack_cluster(n->pcr_add, D); pack_cluster(n->pcr_fix, D + 10);
    for (int k = 0; k < 3; k++) D[20 + k] = n->eig_value[k];
    pack_m3(n->eig_vector, D + 23);
    for (int k = 0; k < 3; k++) { D[32 + k] = n->plane.center[k]; D[35 + k] = n->plane.normal[k]; }
    D[38] = n->plane.radius;
    if (n->plane.is_plane) pack_mat(n->plane.plane_var, D + 39); else for (int k = 0; k < 36; k++) D[39 + k] = 0.0;
    pack_mat(n->cov_add, D + 75);
    for (int i = 0; i < W; i++) {
      if (n->sw != nullptr) { pack_cluster(n->sw->pcrs_local[lm->mp_store[i]], D + 156 + 10 * i); D[156 + 10 * W + i] = (double)n->sw->points[lm->mp_store[i]].size(); }
      else { for (int k = 0; k < 10; k++) D[156 + 10 * i + k] = 0; D[156 + 10 * W + i] = 0; }
    }
  }
  return (int64_t)lv.size();
}
int64_t vxo_localmap_leaf_points(void* m, uint64_t node_id, int which, int64_t capacity, double* out) {
  RefLocalMap* lm = (RefLocalMap*)m;
  std::vector<RefLeaf> lv;
  all_leaves(lm, lv);
  for (const RefLeaf& v : lv) {
    if (v.id != node_id) continue;
    static const PVec none;
    const PVec* src = which < 0 ? &v.node->point_fix : (v.node->sw != nullptr ? &v.node->sw->points[lm->mp_store[which]] : &none);
    for (int64_t i = 0; i < (int64_t)src->size() && i < capacity; i++) {
      for (int k = 0; k < 3; k++) out[12 * i + k] = (*src)[i].pnt[k];
      pack_m3((*src)[i].var, out + 12 * i + 3);
    }
    return (int64_t)src->size();
  }
  return -1;
}

// ---- batch factor construction of the hierarchical BA: OctreeGBA::cut_voxel + OctreeGBA_multi_recut (loop_refine.hpp:273-537) as
// HBA_add_edge calls them (voxelslam.cpp:2374-2379).  Same signature as liboracle's vxo_voxelize; the criteria OctreeGBA hard-codes
// (N > 10, >= 2 observing frames, lambda0/lambda1 <= 0.12) must be the ones asked for.  Scan points are pcl floats upstream: pass
// float-representable coordinates.  Factor voxels come out in hash order and carry no id (node_id = 0): match them by content.
int64_t vxo_voxelize(int W, int64_t n_points, const double* xyz_local, const int64_t* frame_ptr, const double* Rp, const double* params,
                     int64_t capacity, uint64_t* node_id, double* clusters, double* eig_val, double* eig_vec, double* merged) {
  (void)n_points;
  if ((int)params[2] != 10 || params[8] != 0.12 || (int)params[13] > 2) { std::fprintf(stderr, "libref: OctreeGBA hard-codes N > 10, >= 2 frames, ratio 0.12\n"); return -2; }
  for (int k = 0; k < 4; k++) if ((int)params[9 + k] != 0 && (int)params[9 + k] != 10) return -2;
  gba_voxel_size = params[0];
  max_layer = (int)params[1];
  gba_min_eigen_value = params[3];
  gba_eigen_value_array.assign(params + 4, params + 8);
  std::vector<IMUST> xs = unpack_poses(Rp, W);
  std::unordered_map<VOXEL_LOC, OctreeGBA*> oct_map;
  for (int i = 0; i < W; i++) {
    pcl::PointCloud<PointType>::Ptr pl(new pcl::PointCloud<PointType>());
    for (int64_t q = frame_ptr[i]; q < frame_ptr[i + 1]; q++) {
      PointType ap;
      ap.x = (float)xyz_local[3 * q]; ap.y = (float)xyz_local[3 * q + 1]; ap.z = (float)xyz_local[3 * q + 2];
      pl->push_back(ap);
    }
    OctreeGBA::cut_voxel(oct_map, xs[i], pl, i, W);
  }
  LidarFactor voxhess(W);
  OctreeGBA_multi_recut(oct_map, voxhess, 2);
  const int64_t n = (int64_t)voxhess.plvec_voxels.size();
  for (int64_t a = 0; a < n && a < capacity; a++) {
    node_id[a] = 0;
    for (int i = 0; i < W; i++) pack_cluster(voxhess.plvec_voxels[a][i], clusters + ((size_t)a * W + i) * 10);
    for (int k = 0; k < 3; k++) eig_val[3 * a + k] = voxhess.eig_values[a][k];
    pack_m3(voxhess.eig_vectors[a], eig_vec + 9 * a);
    pack_cluster(voxhess.pcr_adds[a], merged + 10 * a);
  }
  return n;
}

// down_sampling_voxel (tools.hpp:201-238); output in the reference's unordered_map order (sort before comparing)
int64_t vxo_down_sampling_voxel(int64_t n, const float* xyz, double voxel_size_, float* out) {
  pcl::PointCloud<PointType> pl;
  for (int64_t i = 0; i < n; i++) { PointType ap; ap.x = xyz[3 * i]; ap.y = xyz[3 * i + 1]; ap.z = xyz[3 * i + 2]; pl.push_back(ap); }
  down_sampling_voxel(pl, voxel_size_);
  for (size_t i = 0; i < pl.size(); i++) { out[3 * i] = pl[i].x; out[3 * i + 1] = pl[i].y; out[3 * i + 2] = pl[i].z; }
  return (int64_t)pl.size();
}

// ---- odometry: the plane association of `match` (voxel_map.hpp:1335-1392, 1674-1698) against the map above -------------------------
// For n world points with their world covariances: flag, sigma_d, and the id of the leaf whose plane was taken (0 when none).
void vxo_localmap_match(void* m, int64_t n, const double* wld, const double* var9, int32_t* flag, double* sigma_d, uint64_t* leaf_id) {
  RefLocalMap* lm = (RefLocalMap*)m;
  lm->bind();
  std::vector<RefLeaf> lv;
  all_leaves(lm, lv);
  std::map<const OctoTree*, uint64_t> id_of;
  for (const RefLeaf& v : lv) id_of[v.node] = v.id;
  for (int64_t i = 0; i < n; i++) {
    Vector3d w = unpack_v3(wld + 3 * i);
    Matrix3d vw = unpack_m3(var9 + 9 * i);
    Plane* pla = nullptr;
    OctoTree* oc = nullptr;
    double sd = 0;
    flag[i] = match(lm->surf_map, w, pla, vw, sd, oc);
    sigma_d[i] = flag[i] ? sd : 0.0;
    leaf_id[i] = (flag[i] && oc) ? id_of[oc] : 0;
  }
}

#ifdef VXREF_LIO_INC
}  // extern "C"
// ---- odometry: VOXEL_SLAM::lio_state_estimation (voxelslam.cpp:856-958), the reference's own text ----------------------------------
// The member function is extracted at build time (oracle/Makefile: _ref/extracted/lio_state_estimation.inc) and compiled inside this harness,
// which supplies the two members it touches: the state x_curr and the voxel map surf_map (a reference to the local map's).
namespace {
struct RefLioHarness {
  IMUST x_curr;
  unordered_map<VOXEL_LOC, OctoTree*>& surf_map;
  explicit RefLioHarness(unordered_map<VOXEL_LOC, OctoTree*>& m) : surf_map(m) {}
#include "lio_state_estimation.inc"
};
}  // namespace
extern "C" {
// state: [R 9 col-major | p 3 | v 3 | bg 3 | ba 3 | g 3] in / out, cov 15 x 15 column-major in / out; points and their body-frame covariances
// (pointVar, voxel_map.hpp:14-19).  Returns the function's verdict (the degeneracy test :950-957).
int vxo_ref_lio_state_estimation(void* m, double* state, double* cov, int64_t n, const double* pnt, const double* var9) {
  RefLocalMap* lm = (RefLocalMap*)m;
  lm->bind();
  RefLioHarness h(lm->surf_map);
  h.x_curr.R = unpack_m3(state);
  h.x_curr.p = unpack_v3(state + 9); h.x_curr.v = unpack_v3(state + 12); h.x_curr.bg = unpack_v3(state + 15); h.x_curr.ba = unpack_v3(state + 18);
  h.x_curr.g = unpack_v3(state + 21);
  for (int c = 0; c < DIM; c++) for (int r = 0; r < DIM; r++) h.x_curr.cov(r, c) = cov[(size_t)c * DIM + r];
  PVecPtr pptr(new PVec((size_t)n));
  for (int64_t i = 0; i < n; i++) { (*pptr)[i].pnt = unpack_v3(pnt + 3 * i); (*pptr)[i].var = unpack_m3(var9 + 9 * i); }
  const bool ok = h.lio_state_estimation(pptr);
  pack_state(h.x_curr, state);
  for (int c = 0; c < DIM; c++) for (int r = 0; r < DIM; r++) cov[(size_t)c * DIM + r] = h.x_curr.cov(r, c);
  return ok ? 1 : 0;
}
}  // extern "C"
#else
}  // extern "C"
#endif
