// TEST INFRASTRUCTURE ONLY (see vxo_linalg.hpp header).  What is and is not pinned to the reference's own code: the headers of the vxo_*.hpp files.
//
// extern "C" surface of the CPU oracle so tests/ and bench.py's cpu_baseline leg
// can drive it through ctypes.  Packed formats are the same as include/vxba.h:
//   cluster  : 10 f64  [Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz N]
//   pose     : 12 f64  [R col-major (9) | p (3)]
//   eig_vec  :  9 f64  col-major (column k = eigenvector k)
//   Hess     : (6W)x(6W) f64 col-major
#include <algorithm>
#include <chrono>
#include <cstdint>

#include "vxo_ba.hpp"
#include "vxo_imu.hpp"
#include "vxo_lio.hpp"
#include "vxo_octree.hpp"
#include "vxo_voxelize.hpp"

using namespace vxo;

namespace {

PointCluster unpack_cluster(const double* c) {
  PointCluster pc;
  pc.P(0,0) = c[0]; pc.P(0,1) = pc.P(1,0) = c[1]; pc.P(0,2) = pc.P(2,0) = c[2];
  pc.P(1,1) = c[3]; pc.P(1,2) = pc.P(2,1) = c[4]; pc.P(2,2) = c[5];
  pc.v = v3(c[6], c[7], c[8]);
  pc.N = (int)c[9];
  return pc;
}
void pack_cluster(const PointCluster& pc, double* c) {
  c[0] = pc.P(0,0); c[1] = pc.P(0,1); c[2] = pc.P(0,2); c[3] = pc.P(1,1); c[4] = pc.P(1,2); c[5] = pc.P(2,2);
  c[6] = pc.v[0]; c[7] = pc.v[1]; c[8] = pc.v[2]; c[9] = (double)pc.N;
}
M3 unpack_m3_colmajor(const double* m) { M3 r; for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) r(rr, c) = m[3 * c + rr]; return r; }
void pack_m3_colmajor(const M3& a, double* m) { for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) m[3 * c + r] = a(r, c); }
std::vector<Pose> unpack_poses(const double* Rp, int W) {
  std::vector<Pose> xs(W);
  for (int i = 0; i < W; i++) {
    xs[i].R = unpack_m3_colmajor(Rp + 12 * i);
    xs[i].p = v3(Rp[12 * i + 9], Rp[12 * i + 10], Rp[12 * i + 11]);
  }
  return xs;
}
void pack_poses(const std::vector<Pose>& xs, double* Rp) {
  for (size_t i = 0; i < xs.size(); i++) {
    pack_m3_colmajor(xs[i].R, Rp + 12 * i);
    for (int k = 0; k < 3; k++) Rp[12 * i + 9 + k] = xs[i].p[k];
  }
}


// ---- inertial half: flat formats shared with include/vxba.h -------------------------------------------------
//   state : 24 f64  [R col-major (9) | p | v | bg | ba | g]
//   imu   : 304 f64 [R_delta 9 | p_delta 3 | v_delta 3 | bg 3 | ba 3 | R_bg 9 | p_bg 9 | p_ba 9 | v_bg 9 | v_ba 9 |
//                    dtime 1 | dbg 3 | dba 3 | dbg_buf 3 | dba_buf 3 | cov 225 col-major]
constexpr int ST = 24, IM = 304;
V3 unpack_v3(const double* a) { return v3(a[0], a[1], a[2]); }
void pack_v3(const V3& a, double* o) { for (int k = 0; k < 3; k++) o[k] = a[k]; }
ImuState unpack_state(const double* s) {
  ImuState x;
  x.R = unpack_m3_colmajor(s);
  x.p = unpack_v3(s + 9); x.v = unpack_v3(s + 12); x.bg = unpack_v3(s + 15); x.ba = unpack_v3(s + 18); x.g = unpack_v3(s + 21);
  return x;
}
void pack_state(const ImuState& x, double* s) {
  pack_m3_colmajor(x.R, s);
  pack_v3(x.p, s + 9); pack_v3(x.v, s + 12); pack_v3(x.bg, s + 15); pack_v3(x.ba, s + 18); pack_v3(x.g, s + 21);
}
IMU_PRE unpack_imu(const double* b) {
  IMU_PRE f;
  f.R_delta = unpack_m3_colmajor(b);
  f.p_delta = unpack_v3(b + 9); f.v_delta = unpack_v3(b + 12); f.bg = unpack_v3(b + 15); f.ba = unpack_v3(b + 18);
  f.R_bg = unpack_m3_colmajor(b + 21); f.p_bg = unpack_m3_colmajor(b + 30); f.p_ba = unpack_m3_colmajor(b + 39);
  f.v_bg = unpack_m3_colmajor(b + 48); f.v_ba = unpack_m3_colmajor(b + 57);
  f.dtime = b[66];
  f.dbg = unpack_v3(b + 67); f.dba = unpack_v3(b + 70); f.dbg_buf = unpack_v3(b + 73); f.dba_buf = unpack_v3(b + 76);
  for (int k = 0; k < 225; k++) f.cov.a[k] = b[79 + k];
  return f;
}
void pack_imu(const IMU_PRE& f, double* b) {
  pack_m3_colmajor(f.R_delta, b);
  pack_v3(f.p_delta, b + 9); pack_v3(f.v_delta, b + 12); pack_v3(f.bg, b + 15); pack_v3(f.ba, b + 18);
  pack_m3_colmajor(f.R_bg, b + 21); pack_m3_colmajor(f.p_bg, b + 30); pack_m3_colmajor(f.p_ba, b + 39);
  pack_m3_colmajor(f.v_bg, b + 48); pack_m3_colmajor(f.v_ba, b + 57);
  b[66] = f.dtime;
  pack_v3(f.dbg, b + 67); pack_v3(f.dba, b + 70); pack_v3(f.dbg_buf, b + 73); pack_v3(f.dba_buf, b + 76);
  for (int k = 0; k < 225; k++) b[79 + k] = f.cov.a[k];
}
MatX unpack_mat(const double* a, int r, int c) { MatX m(r, c); for (int k = 0; k < r * c; k++) m.a[k] = a[k]; return m; }

struct Handle {
  LidarFactor factor;
  explicit Handle(int w) : factor(w) {}
};

}  // namespace

extern "C" {

void* vxo_create(int win_size) { return new Handle(win_size); }
void vxo_destroy(void* h) { delete (Handle*)h; }
void vxo_clear(void* h) { ((Handle*)h)->factor.clear(); }
int vxo_size(void* h) { return (int)((Handle*)h)->factor.plvec_voxels.size(); }

// batched LidarFactor::push_voxel (voxel_map.hpp:122-130)
void vxo_push_voxels(void* h, int n, const double* clusters, const double* fix, const double* coe, const double* eig_val,
                     const double* eig_vec, const double* merged) {
  LidarFactor& f = ((Handle*)h)->factor;
  const int W = f.win_size;
  for (int a = 0; a < n; a++) {
    std::vector<PointCluster> vec(W);
    for (int i = 0; i < W; i++) vec[i] = unpack_cluster(clusters + ((size_t)a * W + i) * 10);
    f.push_voxel(vec, unpack_cluster(fix + (size_t)a * 10), coe[a], v3(eig_val[3 * a], eig_val[3 * a + 1], eig_val[3 * a + 2]),
                 unpack_m3_colmajor(eig_vec + 9 * (size_t)a), unpack_cluster(merged + (size_t)a * 10));
  }
}

void vxo_acc_evaluate2(void* h, const double* Rp, int head, int end, double* Hess, double* JacT, double* residual) {
  LidarFactor& f = ((Handle*)h)->factor;
  const int n = 6 * f.win_size;
  MatX H(n, n);
  std::vector<double> J(n, 0.0);
  f.acc_evaluate2(unpack_poses(Rp, f.win_size), head, end, H, J, *residual);
  std::memcpy(Hess, H.a.data(), sizeof(double) * n * n);
  std::memcpy(JacT, J.data(), sizeof(double) * n);
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
    pack_m3_colmajor(f.eig_vectors[a], eig_vec + 9 * k);
    pack_cluster(f.pcr_adds[a], merged + 10 * k);
  }
}

// Lidar_BA_Optimizer::divide_thread / only_residual with an explicit thread count
double vxo_divide_thread(void* h, const double* Rp, int thd_num, double* Hess, double* JacT) {
  LidarFactor& f = ((Handle*)h)->factor;
  Lidar_BA_Optimizer opt;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.thd_num = thd_num;
  MatX H(opt.jac_leng, opt.jac_leng);
  std::vector<double> J(opt.jac_leng, 0.0);
  std::vector<Pose> xs = unpack_poses(Rp, f.win_size);
  double r = opt.divide_thread(xs, f, H, J);
  if (Hess) std::memcpy(Hess, H.a.data(), sizeof(double) * H.a.size());
  if (JacT) std::memcpy(JacT, J.data(), sizeof(double) * J.size());
  return r;
}
double vxo_only_residual(void* h, const double* Rp, int thd_num) {
  LidarFactor& f = ((Handle*)h)->factor;
  Lidar_BA_Optimizer opt;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.thd_num = thd_num;
  std::vector<Pose> xs = unpack_poses(Rp, f.win_size);
  return opt.only_residual(xs, f);
}

// Lidar_BA_Optimizer::damping_iter (voxel_map.hpp:367-442).
// Rp is in/out.  resis_out[2].  trace_out: up to max_iter rows of 8 doubles
// [residual1 residual2 u v q q1 accepted recomputed_hess]; *n_trace = rows written.
int vxo_damping_iter(void* h, double* Rp, int thd_num, int max_iter, double* hess_out, double* resis_out, double* trace_out,
                     int* n_trace) {
  LidarFactor& f = ((Handle*)h)->factor;
  Lidar_BA_Optimizer opt;
  opt.thd_num = thd_num;
  std::vector<Pose> xs = unpack_poses(Rp, f.win_size);
  MatX hess;
  std::vector<double> resis;
  bool conv = opt.damping_iter(xs, f, &hess, resis, max_iter);
  pack_poses(xs, Rp);
  if (hess_out) std::memcpy(hess_out, hess.a.data(), sizeof(double) * hess.a.size());
  if (resis_out) { resis_out[0] = resis[0]; resis_out[1] = resis[1]; }
  if (n_trace) *n_trace = (int)opt.trace.size();
  if (trace_out)
    for (size_t i = 0; i < opt.trace.size(); i++) {
      const LMTraceEntry& t = opt.trace[i];
      double* o = trace_out + 8 * i;
      o[0] = t.residual1; o[1] = t.residual2; o[2] = t.u; o[3] = t.v; o[4] = t.q; o[5] = t.q1; o[6] = t.accepted; o[7] = t.recomputed_hess;
    }
  return conv ? 1 : 0;
}

// Timed accepted-step BA iteration for the cpu_baseline leg: Hessian sweep
// (divide_thread) + gauge fix + damped solve + state update + residual sweep
// (only_residual), `iters` times; poses are NOT advanced so every iteration does
// identical work.  Returns seconds per iteration (median of iters).
double vxo_time_ba_iteration(void* h, const double* Rp, int thd_num, int warmup, int iters, double* hess_s, double* resid_s) {
  LidarFactor& f = ((Handle*)h)->factor;
  Lidar_BA_Optimizer opt;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.thd_num = thd_num;
  const int n = opt.jac_leng;
  std::vector<Pose> xs = unpack_poses(Rp, f.win_size), xt = xs;
  MatX H(n, n), A(n, n);
  std::vector<double> J(n), rhs(n);
  std::vector<double> tot, th, tr;
  for (int it = 0; it < warmup + iters; it++) {
    auto t0 = std::chrono::steady_clock::now();
    opt.divide_thread(xs, f, H, J);
    auto t1 = std::chrono::steady_clock::now();
    for (int r = 0; r < 6; r++) for (int c = 0; c < n; c++) { H(r, c) = 0.0; H(c, r) = 0.0; }
    for (int r = 0; r < 6; r++) { H(r, r) = 1.0; J[r] = 0.0; }
    for (int c = 0; c < n; c++) for (int r = 0; r < n; r++) A(r, c) = H(r, c) + (r == c ? 0.01 * H(r, r) : 0.0);
    for (int r = 0; r < n; r++) rhs[r] = -J[r];
    std::vector<double> dxi = ldlt_solve(A, rhs);
    for (int j = 0; j < f.win_size; j++) {
      xt[j].R = xs[j].R * Exp(v3(dxi[6 * j], dxi[6 * j + 1], dxi[6 * j + 2]));
      xt[j].p = xs[j].p + v3(dxi[6 * j + 3], dxi[6 * j + 4], dxi[6 * j + 5]);
    }
    auto t2 = std::chrono::steady_clock::now();
    opt.only_residual(xt, f);
    auto t3 = std::chrono::steady_clock::now();
    if (it >= warmup) {
      tot.push_back(std::chrono::duration<double>(t3 - t0).count());
      th.push_back(std::chrono::duration<double>(t1 - t0).count());
      tr.push_back(std::chrono::duration<double>(t3 - t2).count());
    }
  }
  opt.only_residual(xs, f);  // leave the cache at the linearisation point, as the caller seeded it
  auto median = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  if (hess_s) *hess_s = median(th);
  if (resid_s) *resid_s = median(tr);
  return median(tot);
}

// ---- building blocks exposed for unit tests ---------------------------------
void vxo_eig_sym3(const double* C_colmajor, double* val, double* vec_colmajor) {
  V3 l; M3 U;
  eig_sym3(unpack_m3_colmajor(C_colmajor), l, U);
  for (int i = 0; i < 3; i++) val[i] = l[i];
  pack_m3_colmajor(U, vec_colmajor);
}
void vxo_exp(const double* ang, double* R_colmajor) { pack_m3_colmajor(Exp(v3(ang[0], ang[1], ang[2])), R_colmajor); }
void vxo_log(const double* R_colmajor, double* ang) { V3 a = Log(unpack_m3_colmajor(R_colmajor)); for (int i = 0; i < 3; i++) ang[i] = a[i]; }
void vxo_ldlt_solve(int n, const double* A_colmajor, const double* b, double* x) {
  MatX A(n, n);
  std::memcpy(A.a.data(), A_colmajor, sizeof(double) * n * n);
  std::vector<double> r = ldlt_solve(A, std::vector<double>(b, b + n));
  std::memcpy(x, r.data(), sizeof(double) * n);
}
// PointCluster::transform (tools.hpp:357-363)
void vxo_cluster_transform(const double* cluster, const double* Rp, double* out) {
  PointCluster o;
  cluster_transform(o, unpack_cluster(cluster), unpack_poses(Rp, 1)[0]);
  pack_cluster(o, out);
}
// K1: PointCluster::push over bucketed points (tools.hpp:326-331, call sites
// voxel_map.hpp:988, loop_refine.hpp:383-385).  cell_ptr has n_cells+1 entries.
void vxo_build_clusters(int64_t n_cells, const int64_t* cell_ptr, const double* xyz, double* out) {
  for (int64_t c = 0; c < n_cells; c++) {
    PointCluster pc;
    for (int64_t k = cell_ptr[c]; k < cell_ptr[c + 1]; k++) pc.push(v3(xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]));
    pack_cluster(pc, out + 10 * c);
  }
}
// K4: plane fit eig(pcr.cov()) (voxel_map.hpp:1161-1163)
void vxo_plane_fit(int64_t n, const double* clusters, double* eig_val, double* eig_vec) {
  for (int64_t a = 0; a < n; a++) {
    PointCluster pc = unpack_cluster(clusters + 10 * a);
    V3 l; M3 U;
    eig_sym3(pc.cov(), l, U);
    for (int i = 0; i < 3; i++) eig_val[3 * a + i] = l[i];
    pack_m3_colmajor(U, eig_vec + 9 * a);
  }
}

// ---- inertial half ----
void vxo_jr(const double* vec, double* out_colmajor) { pack_m3_colmajor(jr(unpack_v3(vec)), out_colmajor); }
void vxo_jr_inv(const double* R_colmajor, double* out_colmajor) { pack_m3_colmajor(jr_inv(unpack_m3_colmajor(R_colmajor)), out_colmajor); }
void vxo_mat_inverse(int n, const double* A_colmajor, double* out) {
  MatX inv = mat_inverse(unpack_mat(A_colmajor, n, n));
  std::memcpy(out, inv.a.data(), sizeof(double) * n * n);
}
// IMU_PRE::IMU_PRE(bg, ba) + add_imu (preintegration.hpp:32-48, 75-135)
void vxo_imu_init(double* blob, const double* bg, const double* ba) { pack_imu(IMU_PRE(unpack_v3(bg), unpack_v3(ba)), blob); }
void vxo_imu_add(double* blob, const double* gyr, const double* acc, double dt, const double* noiseMeas36, const double* noiseWalk36) {
  IMU_PRE f = unpack_imu(blob);
  f.add_imu(unpack_v3(gyr), unpack_v3(acc), dt, unpack_mat(noiseMeas36, 6, 6), unpack_mat(noiseWalk36, 6, 6));
  pack_imu(f, blob);
}
// IMU_PRE::give_evaluate (:137-212): returns r^T cov^-1 r; jtj 30x30 col-major, gg 30 (written when jac_enable)
double vxo_imu_evaluate(const double* blob, const double* st1, const double* st2, double* jtj, double* gg, int jac_enable) {
  IMU_PRE f = unpack_imu(blob);
  MatX J(2 * DIM, 2 * DIM);
  std::vector<double> g(2 * DIM, 0.0);
  const double r = f.give_evaluate(unpack_state(st1), unpack_state(st2), J, g, jac_enable != 0);
  if (jac_enable) {
    if (jtj) std::memcpy(jtj, J.a.data(), sizeof(double) * 900);
    if (gg) std::memcpy(gg, g.data(), sizeof(double) * 30);
  }
  return r;
}
namespace {
struct LiCtx {
  std::vector<ImuState> xs;
  std::vector<IMU_PRE> store;
  std::deque<IMU_PRE*> imus;
  LiCtx(const double* states, const double* blobs, int W) {
    xs.resize(W);
    for (int i = 0; i < W; i++) xs[i] = unpack_state(states + ST * i);
    store.resize(W - 1);
    for (int i = 0; i < W - 1; i++) store[i] = unpack_imu(blobs + IM * i);
    for (int i = 0; i < W - 1; i++) imus.push_back(&store[i]);
  }
};
}  // namespace
// LI_BA_Optimizer::divide_thread (voxel_map.hpp:465-523): Hess (15W)^2 col-major, JacT 15W; returns the residual
double vxo_li_divide_thread(void* h, const double* states, const double* blobs, int thd_num, double imu_coef, double* Hess, double* JacT) {
  LidarFactor& f = ((Handle*)h)->factor;
  LiCtx c(states, blobs, f.win_size);
  LI_BA_Optimizer opt;
  opt.thd_num = thd_num; opt.imu_coef = imu_coef;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.imu_leng = DIM * f.win_size;
  MatX H(opt.imu_leng, opt.imu_leng);
  std::vector<double> J(opt.imu_leng);
  const double r = opt.divide_thread(c.xs, f, c.imus, H, J);
  std::memcpy(Hess, H.a.data(), sizeof(double) * H.a.size());
  std::memcpy(JacT, J.data(), sizeof(double) * J.size());
  return r;
}
double vxo_li_only_residual(void* h, const double* states, const double* blobs, int thd_num, double imu_coef) {
  LidarFactor& f = ((Handle*)h)->factor;
  LiCtx c(states, blobs, f.win_size);
  LI_BA_Optimizer opt;
  opt.thd_num = thd_num; opt.imu_coef = imu_coef;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.imu_leng = DIM * f.win_size;
  return opt.only_residual(c.xs, f, c.imus);
}
// LI_BA_Optimizer::damping_iter (:562-653).  states / blobs are in/out (dbg, dba and their _buf copies change).
void vxo_li_damping_iter(void* h, double* states, double* blobs, int thd_num, double imu_coef, int max_iter, double* hess_out,
                         double* trace_out, int* n_trace) {
  LidarFactor& f = ((Handle*)h)->factor;
  const int W = f.win_size;
  LiCtx c(states, blobs, W);
  LI_BA_Optimizer opt;
  opt.thd_num = thd_num; opt.imu_coef = imu_coef;
  MatX hess;
  opt.damping_iter(c.xs, f, c.imus, &hess, max_iter);
  for (int i = 0; i < W; i++) pack_state(c.xs[i], states + ST * i);
  for (int i = 0; i < W - 1; i++) pack_imu(c.store[i], blobs + IM * i);
  if (hess_out) std::memcpy(hess_out, hess.a.data(), sizeof(double) * hess.a.size());
  if (n_trace) *n_trace = (int)opt.trace.size();
  if (trace_out)
    for (size_t i = 0; i < opt.trace.size(); i++) {
      const LMTraceEntry& t = opt.trace[i];
      double* o = trace_out + 8 * i;
      o[0] = t.residual1; o[1] = t.residual2; o[2] = t.u; o[3] = t.v; o[4] = t.q; o[5] = t.q1; o[6] = t.accepted; o[7] = t.recomputed_hess;
    }
}

// IMU_PRE::give_evaluate_g (:214-294): jtj 33x33, gg 33
double vxo_imu_evaluate_g(const double* blob, const double* st1, const double* st2, double* jtj, double* gg, int jac_enable) {
  IMU_PRE f = unpack_imu(blob);
  MatX J(2 * DIM + 3, 2 * DIM + 3);
  std::vector<double> g(2 * DIM + 3, 0.0);
  const double r = f.give_evaluate_g(unpack_state(st1), unpack_state(st2), J, g, jac_enable != 0);
  if (jac_enable) {
    if (jtj) std::memcpy(jtj, J.a.data(), sizeof(double) * 33 * 33);
    if (gg) std::memcpy(gg, g.data(), sizeof(double) * 33);
  }
  return r;
}
// LI_BA_OptimizerGravity::divide_thread (voxel_map.hpp:673-736): Hess (15W+3)^2 col-major, JacT 15W+3; returns the residual
double vxo_li_divide_thread_gravity(void* h, const double* states, const double* blobs, int thd_num, double imu_coef, double* Hess, double* JacT) {
  LidarFactor& f = ((Handle*)h)->factor;
  LiCtx c(states, blobs, f.win_size);
  LI_BA_OptimizerGravity opt;
  opt.thd_num = thd_num; opt.imu_coef = imu_coef;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.imu_leng = DIM * f.win_size + 3;
  MatX H(opt.imu_leng, opt.imu_leng);
  std::vector<double> J(opt.imu_leng);
  const double r = opt.divide_thread(c.xs, f, c.imus, H, J);
  std::memcpy(Hess, H.a.data(), sizeof(double) * H.a.size());
  std::memcpy(JacT, J.data(), sizeof(double) * J.size());
  return r;
}
// LI_BA_OptimizerGravity::only_residual (voxel_map.hpp:738-773)
double vxo_li_only_residual_gravity(void* h, const double* states, const double* blobs, int thd_num, double imu_coef) {
  LidarFactor& f = ((Handle*)h)->factor;
  LiCtx c(states, blobs, f.win_size);
  LI_BA_OptimizerGravity opt;
  opt.thd_num = thd_num; opt.imu_coef = imu_coef;
  opt.win_size = f.win_size; opt.jac_leng = 6 * f.win_size; opt.imu_leng = DIM * f.win_size + 3;
  return opt.only_residual(c.xs, f, c.imus);
}
// LI_BA_OptimizerGravity::damping_iter (voxel_map.hpp:775-862): hess_out (15W+3)^2, resis_out[2]
void vxo_li_damping_iter_gravity(void* h, double* states, double* blobs, int thd_num, double imu_coef, int max_iter, double* hess_out,
                                 double* resis_out, double* trace_out, int* n_trace) {
  LidarFactor& f = ((Handle*)h)->factor;
  const int W = f.win_size;
  LiCtx c(states, blobs, W);
  LI_BA_OptimizerGravity opt;
  opt.thd_num = thd_num; opt.imu_coef = imu_coef;
  MatX hess;
  std::vector<double> resis;
  opt.damping_iter(c.xs, f, c.imus, resis, &hess, max_iter);
  for (int i = 0; i < W; i++) pack_state(c.xs[i], states + ST * i);
  for (int i = 0; i < W - 1; i++) pack_imu(c.store[i], blobs + IM * i);
  if (hess_out) std::memcpy(hess_out, hess.a.data(), sizeof(double) * hess.a.size());
  if (resis_out && resis.size() >= 2) { resis_out[0] = resis[0]; resis_out[1] = resis[1]; }
  if (n_trace) *n_trace = (int)opt.trace.size();
  if (trace_out)
    for (size_t i = 0; i < opt.trace.size(); i++) {
      const LMTraceEntry& t = opt.trace[i];
      double* o = trace_out + 8 * i;
      o[0] = t.residual1; o[1] = t.residual2; o[2] = t.u; o[3] = t.v; o[4] = t.q; o[5] = t.q1; o[6] = t.accepted; o[7] = t.recomputed_hess;
    }
}

// OctreeGBA::cut_voxel + recut (loop_refine.hpp:358-405, 446-476; voxelslam.cpp:2374-2379).
// params = [voxel_size, max_layer, min_points, min_eigen_value, ratio0, ratio1, ratio2, ratio3, factor_ratio_max].
// Returns the number of factor voxels (canonical order) or -1 if a point leaves the id range; fills up to `capacity` of them.
int64_t vxo_voxelize(int W, int64_t n_points, const double* xyz_local, const int64_t* frame_ptr, const double* Rp, const double* params,
                     int64_t capacity, uint64_t* node_id, double* clusters, double* eig_val, double* eig_vec, double* merged) {
  VoxelizeParams p;
  p.voxel_size = params[0]; p.max_layer = (int)params[1]; p.min_points = (int)params[2]; p.min_eigen_value = params[3];
  for (int k = 0; k < 4; k++) p.eigen_ratio[k] = params[4 + k];
  p.factor_ratio_max = params[8];
  for (int k = 0; k < 4; k++) p.min_points_layer[k] = (int)params[9 + k];
  p.min_frames = (int)params[13];
  std::vector<std::vector<V3>> clouds(W);
  for (int i = 0; i < W; i++)
    for (int64_t q = frame_ptr[i]; q < frame_ptr[i + 1]; q++) clouds[i].push_back(v3(xyz_local[3 * q], xyz_local[3 * q + 1], xyz_local[3 * q + 2]));
  (void)n_points;
  std::vector<FactorVoxel> out;
  if (!voxelize(W, clouds, unpack_poses(Rp, W), p, out)) return -1;
  const int64_t n = (int64_t)out.size();
  for (int64_t a = 0; a < n && a < capacity; a++) {
    const FactorVoxel& f = out[a];
    node_id[a] = f.node_id;
    for (int i = 0; i < W; i++) pack_cluster(f.pcrs[i], clusters + ((size_t)a * W + i) * 10);
    for (int k = 0; k < 3; k++) eig_val[3 * a + k] = f.eig_value[k];
    pack_m3_colmajor(f.eig_vector, eig_vec + 9 * a);
    pack_cluster(f.pcr_add, merged + 10 * a);
  }
  return n;
}

// ---- odometry point-to-plane update (vxo_lio.hpp) ------------------------------------------------------------------------
namespace {
struct LioHandle {
  PlaneMap map;
  std::vector<PointVar> pts;
  std::vector<OctoNode*> octos;
};
void pack_sweep(const LioSweep& sw, double* out) {   // [HTH 36 col-major | HTz 6 | nnt 9 col-major | match_num]
  std::memcpy(out, sw.HTH, sizeof(double) * 36);
  std::memcpy(out + 36, sw.HTz, sizeof(double) * 6);
  std::memcpy(out + 42, sw.nnt, sizeof(double) * 9);
  out[51] = (double)sw.match_num;
}
M3 block33(const double* cov225, int r0) { M3 m; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m(r, c) = cov225[15 * (r0 + c) + r0 + r]; return m; }
}  // namespace

void* vxo_lio_create(double voxel_size, int max_layer) {
  LioHandle* h = new LioHandle();
  h->map.prm.voxel_size = voxel_size; h->map.prm.max_layer = max_layer;
  return h;
}
void vxo_lio_destroy(void* h) { delete (LioHandle*)h; }
void vxo_lio_map_add(void* hv, int64_t n, const int64_t* loc, const int32_t* layer, const int32_t* path, const int32_t* is_plane, const double* center,
                     const double* normal, const double* plane_var, const double* radius) {
  LioHandle* h = (LioHandle*)hv;
  for (int64_t i = 0; i < n; i++) h->map.add_leaf(loc + 3 * i, layer[i], path[i], is_plane ? is_plane[i] : 1, center + 3 * i, normal + 3 * i, plane_var + 36 * i, radius[i]);
}
void vxo_lio_scan_set(void* hv, int64_t n, const double* pnt, const double* var9) {
  LioHandle* h = (LioHandle*)hv;
  h->pts.resize(n);
  for (int64_t i = 0; i < n; i++) { h->pts[i].pnt = unpack_v3(pnt + 3 * i); h->pts[i].var = unpack_m3_colmajor(var9 + 9 * i); }
  h->octos.assign(n, nullptr);
}
void vxo_lio_scan_raw(void* hv, int64_t n, const float* xyz, const double* ext12, double dept_err, double beam_err) {
  LioHandle* h = (LioHandle*)hv;
  var_init(unpack_m3_colmajor(ext12), unpack_v3(ext12 + 9), n, xyz, (float)dept_err, (float)beam_err, h->pts);
  h->octos.assign(n, nullptr);
}
int64_t vxo_lio_scan_size(void* hv) { return (int64_t)((LioHandle*)hv)->pts.size(); }
void vxo_lio_scan_read(void* hv, double* pnt, double* var9) {
  LioHandle* h = (LioHandle*)hv;
  for (size_t i = 0; i < h->pts.size(); i++) {
    for (int k = 0; k < 3; k++) pnt[3 * i + k] = h->pts[i].pnt[k];
    pack_m3_colmajor(h->pts[i].var, var9 + 9 * i);
  }
}
// one pass of voxelslam.cpp:873-919; state = VXBA state layout [R 9 | p 3 | ...], cov 15x15 col-major
void vxo_lio_sweep(void* hv, const double* state, const double* cov225, int reset_cache, double* out52, int32_t* plane_of_point, double* sigma_of_point) {
  LioHandle* h = (LioHandle*)hv;
  if (reset_cache) h->octos.assign(h->pts.size(), nullptr);
  ImuState x = unpack_state(state);
  LioSweep sw;
  std::vector<int> pop(h->pts.size(), -1);
  std::vector<double> sig(h->pts.size(), 0.0);
  lio_sweep(h->map, h->pts, x, block33(cov225, 0), block33(cov225, 3), h->octos, sw, &pop, &sig);
  pack_sweep(sw, out52);
  if (plane_of_point) for (size_t i = 0; i < pop.size(); i++) plane_of_point[i] = pop[i];
  if (sigma_of_point) for (size_t i = 0; i < sig.size(); i++) sigma_of_point[i] = sig[i];
}
// lio_state_estimation (voxelslam.cpp:855-958).  info = [ok, iterations, match_num, min eigenvalue of nnt]; sweeps_out 4 x 52.
void vxo_lio_state_estimation(void* hv, double* state, double* cov225, double* info, double* sweeps_out, int32_t* plane_of_point, double* sigma_of_point) {
  LioHandle* h = (LioHandle*)hv;
  LioState st;
  st.x = unpack_state(state);
  std::memcpy(st.cov.a.data(), cov225, sizeof(double) * 225);
  LioResult r = lio_state_estimation(h->map, h->pts, st);
  pack_state(st.x, state);
  std::memcpy(cov225, st.cov.a.data(), sizeof(double) * 225);
  info[0] = r.ok ? 1.0 : 0.0; info[1] = r.iterations; info[2] = r.match_num; info[3] = r.min_eig;
  if (sweeps_out) for (size_t k = 0; k < r.sweeps.size(); k++) pack_sweep(r.sweeps[k], sweeps_out + 52 * k);
  if (plane_of_point) for (size_t i = 0; i < r.plane_of_point.size(); i++) plane_of_point[i] = r.plane_of_point[i];
  if (sigma_of_point) for (size_t i = 0; i < r.sigma_of_point.size(); i++) sigma_of_point[i] = r.sigma_of_point[i];
}
double vxo_time_lio_state_estimation(void* hv, const double* state, const double* cov225, int reps) {
  LioHandle* h = (LioHandle*)hv;
  double best = 1e300;
  for (int k = 0; k < reps; k++) {
    LioState st;
    st.x = unpack_state(state);
    std::memcpy(st.cov.a.data(), cov225, sizeof(double) * 225);
    auto t0 = std::chrono::steady_clock::now();
    LioResult r = lio_state_estimation(h->map, h->pts, st);
    best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    if (r.iterations < 0) return -1;
  }
  return best;
}
void vxo_lio_pvec_update(void* hv, const double* state, const double* cov225, double* pwld, double* var9) {
  LioHandle* h = (LioHandle*)hv;
  std::vector<PointVar> w;
  pvec_update(h->pts, unpack_state(state), block33(cov225, 0), block33(cov225, 3), w);
  for (size_t i = 0; i < w.size(); i++) {
    for (int k = 0; k < 3; k++) pwld[3 * i + k] = w[i].pnt[k];
    pack_m3_colmajor(w[i].var, var9 + 9 * i);
  }
}
// cov_add of OctoTree::push (voxel_map.hpp:990-992): sum of Bf_var over the points of each cell, 81 f64 col-major per cell
void vxo_cov_add_build(int64_t n_cells, const int64_t* cell_ptr, const double* xyz_world, const double* var9, double* cov_add) {
  for (int64_t c = 0; c < n_cells; c++) {
    double acc[81] = {0}, b[81];
    for (int64_t q = cell_ptr[c]; q < cell_ptr[c + 1]; q++) {
      bf_var(unpack_m3_colmajor(var9 + 9 * q), unpack_v3(xyz_world + 3 * q), b);
      for (int i = 0; i < 81; i++) acc[i] += b[i];
    }
    std::memcpy(cov_add + 81 * c, acc, sizeof acc);
  }
}
// OctoTree::plane_update (voxel_map.hpp:1118-1146), batched
void vxo_plane_update(int64_t n, const double* clusters, const double* eig_val, const double* eig_vec, const double* cov_add, double* center, double* normal,
                      double* plane_var, double* radius) {
  for (int64_t a = 0; a < n; a++) {
    PointCluster pc = unpack_cluster(clusters + 10 * a);
    float rad;
    plane_update(pc.P, pc.v, (double)pc.N, unpack_v3(eig_val + 3 * a), unpack_m3_colmajor(eig_vec + 9 * a), cov_add + 81 * a, center + 3 * a, normal + 3 * a,
                 plane_var + 36 * a, rad);
    radius[a] = rad;
  }
}

int64_t vxo_down_sampling_voxel(int64_t n, const float* xyz, double voxel_size, float* out) {
  std::vector<float> v(xyz, xyz + 3 * n);
  down_sampling_voxel(v, voxel_size);
  std::memcpy(out, v.data(), v.size() * sizeof(float));
  return (int64_t)(v.size() / 3);
}


// ---- incremental local map (vxo_octree.hpp) ----
// params = [voxel_size, max_layer, min_point[4], min_eigen_value, plane_eigen_value_thre[4], max_points, win_size, thread_num]
void* vxo_localmap_create(const double* p) {
  LocalMapParams q;
  q.voxel_size = p[0]; q.max_layer = (int)p[1];
  for (int k = 0; k < 4; k++) q.min_point[k] = p[2 + k];
  q.min_eigen_value = p[6];
  for (int k = 0; k < 4; k++) q.plane_eigen_value_thre[k] = p[7 + k];
  q.max_points = (int)p[11]; q.win_size = (int)p[12]; q.thread_num = (int)p[13];
  return new LocalMap(q);
}
void vxo_localmap_destroy(void* m) { delete (LocalMap*)m; }
// scan `ord` of the window: body points (n*3), their WORLD covariances (n*9 col-major, as pvec_update leaves them), world points (n*3)
void vxo_localmap_cut_voxel(void* m, int ord, int64_t n, const double* pnt, const double* var, const double* pwld) {
  std::vector<PointVar> pv((size_t)n);
  std::vector<V3> pw((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    pv[i].pnt = v3(pnt[3 * i], pnt[3 * i + 1], pnt[3 * i + 2]);
    pv[i].var = unpack_m3_colmajor(var + 9 * i);
    pw[i] = v3(pwld[3 * i], pwld[3 * i + 1], pwld[3 * i + 2]);
  }
  ((LocalMap*)m)->cut_voxel(ord, pv, pw);
}
// multi_recut: recut + tras_opt into the factor of an oracle handle (vxo_create)
void vxo_localmap_recut(void* m, int win_count, const double* Rp, void* factor) {
  ((LocalMap*)m)->recut(win_count, unpack_poses(Rp, win_count), ((Handle*)factor)->factor);
}
// multi_margi with mgsize = 1; the factor is the one tras_opt filled (its pcr_adds / eig_* as the optimiser left them)
int vxo_localmap_margi(void* m, int win_count, const double* Rp, void* factor) {
  try { ((LocalMap*)m)->margi(win_count, unpack_poses(Rp, win_count), ((Handle*)factor)->factor); } catch (const std::exception&) { return -1; }
  return 0;
}
void vxo_localmap_slide(void* m, int mgsize) { ((LocalMap*)m)->slide(mgsize); }
void vxo_localmap_counts(void* m, int64_t* out) {   // [roots, roots in the slide map, leaves, mp[0]]
  LocalMap* lm = (LocalMap*)m;
  std::vector<LeafView> lv;
  lm->leaves(lv);
  out[0] = (int64_t)lm->surf_map.size(); out[1] = (int64_t)lm->slide_order.size(); out[2] = (int64_t)lv.size(); out[3] = lm->mp[0];
}
// Every leaf (octo_state == 0).  ints: n x 8 = [layer, isexist, is_plane, has_sw, opt_state, last_num, point_fix.size(), root in slide map];
// dbl: n x (156 + 11 W) = [pcr_add 10 | pcr_fix 10 | eig_value 3 | eig_vector 9 col-major | center 3 | normal 3 | radius | plane_var 36
// col-major | cov_add 81 col-major | pcrs_local W x 10 in window order | points kept per window slot W].  Returns the leaf count, -1 if a
// root lies outside the id range.
int64_t vxo_localmap_leaves(void* m, int64_t capacity, uint64_t* ids, int32_t* ints, double* dbl) {
  LocalMap* lm = (LocalMap*)m;
  std::vector<LeafView> lv;
  if (!lm->leaves(lv)) return -1;
  const int W = lm->prm.win_size;
  const size_t rec = 156 + 11 * (size_t)W;
  for (int64_t a = 0; a < (int64_t)lv.size() && a < capacity; a++) {
    const TreeNode* n = lv[a].node;
    ids[a] = lv[a].node_id;
    const uint64_t r = lv[a].node_id >> 16;
    const LocKey key{(int64_t)((r >> 32) & 0xffff) - 32768, (int64_t)((r >> 16) & 0xffff) - 32768, (int64_t)(r & 0xffff) - 32768};
    int32_t* I = ints + 8 * a;
    I[0] = n->layer; I[1] = n->isexist; I[2] = n->plane.is_plane; I[3] = n->has_sw; I[4] = n->opt_state; I[5] = n->last_num;
    I[6] = (int32_t)n->point_fix.size(); I[7] = (int32_t)lm->slide_set.count(key);
    double* D = dbl + rec * a;
    pack_cluster(n->pcr_add, D); pack_cluster(n->pcr_fix, D + 10);
    for (int k = 0; k < 3; k++) D[20 + k] = n->eig_value[k];
    pack_m3_colmajor(n->eig_vector, D + 23);
    for (int k = 0; k < 3; k++) { D[32 + k] = n->plane.center[k]; D[35 + k] = n->plane.normal[k]; }
    D[38] = n->plane.radius;
    for (int k = 0; k < 36; k++) D[39 + k] = n->plane.is_plane ? n->plane.plane_var[k] : 0.0;
    for (int k = 0; k < 81; k++) D[75 + k] = n->cov_add[k];
    for (int i = 0; i < W; i++) {
      if (n->has_sw) { pack_cluster(n->pcrs_local[lm->mp[i]], D + 156 + 10 * i); D[156 + 10 * W + i] = (double)n->points[lm->mp[i]].size(); }
      else { for (int k = 0; k < 10; k++) D[156 + 10 * i + k] = 0; D[156 + 10 * W + i] = 0; }
    }
  }
  return (int64_t)lv.size();
}

// The points a leaf keeps for a later subdivision: which = -1 -> point_fix (world frame), which = i >= 0 -> the window's i-th scan
// (sw->points[mp[i]], body frame).  out: n x 12 = [pnt 3 | var 9 col-major].  Returns the count (fills up to `capacity`), -1 if no such leaf.
int64_t vxo_localmap_leaf_points(void* m, uint64_t node_id, int which, int64_t capacity, double* out) {
  LocalMap* lm = (LocalMap*)m;
  std::vector<LeafView> lv;
  lm->leaves(lv);
  for (const LeafView& v : lv) {
    if (v.node_id != node_id) continue;
    const std::vector<PointVar>* src = nullptr;
    static const std::vector<PointVar> none;
    if (which < 0) src = &v.node->point_fix;
    else src = v.node->has_sw ? &v.node->points[lm->mp[which]] : &none;
    for (int64_t i = 0; i < (int64_t)src->size() && i < capacity; i++) {
      for (int k = 0; k < 3; k++) out[12 * i + k] = (*src)[i].pnt[k];
      pack_m3_colmajor((*src)[i].var, out + 12 * i + 3);
    }
    return (int64_t)src->size();
  }
  return -1;
}
}  // extern "C"
