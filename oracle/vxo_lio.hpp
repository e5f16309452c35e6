// TEST INFRASTRUCTURE ONLY (see vxo_linalg.hpp header).  PINNED against the reference's own code compiled here (oracle/_ref/libref.so, `make -C oracle ref`: tests/test_ref_pin.py):
// `match` / OctoTree::match on the reference's own tree (test_plane_match_...) and, since round 5, the EKF loop itself: the TEXT of
// VOXEL_SLAM::lio_state_estimation (voxelslam.cpp:856-958) is extracted at build time into oracle/_ref/extracted/ and compiled inside a
// harness in libref.so (test_lio_state_estimation_restatement_matches_the_reference_text).  calcBodyVar / var_init / pvec_update
// (voxelslam.hpp:163-215, ROS headers) remain pinned through mathematics only.
//
// CPU restatement of the odometry's point-to-plane update:
//   calcBodyVar / var_init          VoxelSLAM/src/voxelslam.hpp:163-201
//   pvec_update                     VoxelSLAM/src/voxelslam.hpp:203-215
//   lio_state_estimation            VoxelSLAM/src/voxelslam.cpp:855-958
//   OctoTree::match / inside        VoxelSLAM/src/voxel_map.hpp:1335-1392, 1471-1480
//   match(feat_map, ...)            VoxelSLAM/src/voxel_map.hpp:1674-1698
//   OctoTree::plane_update, Bf_var  VoxelSLAM/src/voxel_map.hpp:91-106, 1118-1146
// The map is rebuilt here as a pointer tree (hash of roots, eight children per node, centres derived exactly as
// cut_voxel :1529-1535 and allocate :1037-1043 derive them) from a flat list of plane-carrying leaves, so that the walk,
// the float-typed voxel index, the float-typed distance tests and the per-point node cache across EKF iterations are the
// reference's own.  DEG2RAD is PCL's macro ((x)*0.017453293, pcl/pcl_macros.h, PCL 1.10 per README.md:25).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <unordered_map>
#include <vector>

#include "vxo_imu.hpp"
#include "vxo_linalg.hpp"

namespace vxo {

struct PointVar {  // voxel_map.hpp:14-19
  V3 pnt;
  M3 var;
};

// voxelslam.hpp:164-185
inline void calc_body_var(V3& pb, const float range_inc, const float degree_inc, M3& var) {
  if (pb[2] == 0) pb[2] = 0.0001;
  float range = std::sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]);
  float range_var = range_inc * range_inc;
  const double dv = std::pow(std::sin((degree_inc) * 0.017453293), 2);
  V3 direction = pb / norm(pb);
  M3 dhat = hat(direction);
  V3 b1 = v3(1, 1, -(direction[0] + direction[1]) / direction[2]);
  b1 = b1 / norm(b1);
  V3 b2 = v3(b1[1] * direction[2] - b1[2] * direction[1], b1[2] * direction[0] - b1[0] * direction[2], b1[0] * direction[1] - b1[1] * direction[0]);
  b2 = b2 / norm(b2);
  // A = range * direction_hat * N, N = [b1 b2]
  V3 a1 = (double)range * (dhat * b1), a2 = (double)range * (dhat * b2);
  var = outer(direction * (double)range_var, direction) + outer(a1 * dv, a1) + outer(a2 * dv, a2);
}

// voxelslam.hpp:187-201
inline void var_init(const M3& extR, const V3& extp, int64_t n, const float* xyz, float dept_err, float beam_err, std::vector<PointVar>& out) {
  out.resize(n);
  for (int64_t i = 0; i < n; i++) {
    PointVar& pv = out[i];
    pv.pnt = v3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    calc_body_var(pv.pnt, dept_err, beam_err, pv.var);
    pv.pnt = extR * pv.pnt + extp;
    pv.var = extR * pv.var * transpose(extR);
  }
}

struct Plane {  // voxel_map.hpp:66-80
  V3 center = zero3(), normal = zero3();
  double plane_var[36];  // column-major 6x6
  float radius = 0;
  bool is_plane = false;
  int id = -1;  // position in the caller's leaf list (bookkeeping of this restatement only)
};

struct MapParams {
  double voxel_size = 1.0;
  int max_layer = 2;
};

struct OctoNode {
  int layer = 0, octo_state = 0;
  std::unique_ptr<OctoNode> leaves[8];
  double voxel_center[3] = {0, 0, 0};
  float quater_length = 0;
  Plane plane;

  // voxel_map.hpp:1471-1480
  bool inside(const V3& wld) const {
    double hl = quater_length * 2;
    return (wld[0] >= voxel_center[0] - hl && wld[0] <= voxel_center[0] + hl && wld[1] >= voxel_center[1] - hl && wld[1] <= voxel_center[1] + hl &&
            wld[2] >= voxel_center[2] - hl && wld[2] <= voxel_center[2] + hl);
  }

  // voxel_map.hpp:1335-1392
  int match(const V3& wld, Plane*& pla, const M3& var_wld, double& sigma_d, OctoNode*& oc) {
    int flag = 0;
    if (octo_state == 0) {
      if (plane.is_plane) {
        V3 d = wld - plane.center;
        float dis_to_plane = std::fabs(dot(plane.normal, d));
        float dis_to_center = dot(d, d);
        float range_dis = (dis_to_center - dis_to_plane * dis_to_plane);
        if (range_dis <= 3 * 3 * plane.radius) {
          double J[6] = {d[0], d[1], d[2], -plane.normal[0], -plane.normal[1], -plane.normal[2]};
          double sigma_l = 0;
          for (int c = 0; c < 6; c++) {
            double t = 0;
            for (int r = 0; r < 6; r++) t += J[r] * plane.plane_var[6 * c + r];
            sigma_l += t * J[c];
          }
          sigma_l += dot(plane.normal, var_wld * plane.normal);
          if (dis_to_plane < 3 * std::sqrt(sigma_l)) {
            oc = this;
            sigma_d = sigma_l;
            pla = &plane;
            flag = 1;
          }
        }
      }
    } else {
      int xyz[3] = {0, 0, 0};
      for (int k = 0; k < 3; k++)
        if (wld[k] > voxel_center[k]) xyz[k] = 1;
      int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
      if (leaves[leafnum]) flag = leaves[leafnum]->match(wld, pla, var_wld, sigma_d, oc);
    }
    return flag;
  }
};

struct LocKey {
  int64_t x, y, z;
  bool operator==(const LocKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct LocHash {
  size_t operator()(const LocKey& k) const { return (size_t)((((uint64_t)k.z * 116101ull) % 10000000000ull + (uint64_t)k.y) * 116101ull) % 10000000000ull + (uint64_t)k.x; }
};

// the float-typed voxel index of cut_voxel / match (voxel_map.hpp:1511-1518, 1678-1685)
inline LocKey voxel_of(const V3& w, double voxel_size) {
  float loc[3];
  for (int j = 0; j < 3; j++) {
    loc[j] = w[j] / voxel_size;
    if (loc[j] < 0) loc[j] -= 1;
  }
  return LocKey{(int64_t)loc[0], (int64_t)loc[1], (int64_t)loc[2]};
}

class PlaneMap {
 public:
  MapParams prm;
  std::unordered_map<LocKey, std::unique_ptr<OctoNode>, LocHash> roots;
  int n_leaves = 0;

  // One plane-carrying leaf: root voxel, depth, octant taken at each level (3 bits per level, first level in the low bits).
  // is_plane == 0 plants the node without a plane (a leaf the reference keeps but never matches).
  void add_leaf(const int64_t loc[3], int layer, int path, int is_plane, const double* center, const double* normal, const double* plane_var, double radius) {
    LocKey k{loc[0], loc[1], loc[2]};
    auto it = roots.find(k);
    if (it == roots.end()) {
      std::unique_ptr<OctoNode> r(new OctoNode());
      r->layer = 0;
      r->voxel_center[0] = (0.5 + k.x) * prm.voxel_size;
      r->voxel_center[1] = (0.5 + k.y) * prm.voxel_size;
      r->voxel_center[2] = (0.5 + k.z) * prm.voxel_size;
      r->quater_length = prm.voxel_size / 4.0;
      it = roots.emplace(k, std::move(r)).first;
    }
    OctoNode* nd = it->second.get();
    for (int l = 0; l < layer; l++) {
      const int leafnum = (path >> (3 * l)) & 7;
      const int xyz[3] = {(leafnum >> 2) & 1, (leafnum >> 1) & 1, leafnum & 1};
      nd->octo_state = 1;
      if (!nd->leaves[leafnum]) {
        std::unique_ptr<OctoNode> c(new OctoNode());
        c->layer = nd->layer + 1;
        c->voxel_center[0] = nd->voxel_center[0] + (2 * xyz[0] - 1) * nd->quater_length;
        c->voxel_center[1] = nd->voxel_center[1] + (2 * xyz[1] - 1) * nd->quater_length;
        c->voxel_center[2] = nd->voxel_center[2] + (2 * xyz[2] - 1) * nd->quater_length;
        c->quater_length = nd->quater_length / 2;
        nd->leaves[leafnum] = std::move(c);
      }
      nd = nd->leaves[leafnum].get();
    }
    nd->plane.is_plane = is_plane != 0;
    nd->plane.center = v3(center[0], center[1], center[2]);
    nd->plane.normal = v3(normal[0], normal[1], normal[2]);
    for (int i = 0; i < 36; i++) nd->plane.plane_var[i] = plane_var[i];
    nd->plane.radius = (float)radius;
    nd->plane.id = n_leaves++;
  }

  // voxel_map.hpp:1674-1698
  int match(const V3& wld, Plane*& pla, const M3& var_wld, double& sigma_d, OctoNode*& oc) {
    int flag = 0;
    auto iter = roots.find(voxel_of(wld, prm.voxel_size));
    if (iter != roots.end()) flag = iter->second->match(wld, pla, var_wld, sigma_d, oc);
    return flag;
  }
};

struct LioState {  // the IMUST fields lio_state_estimation touches (tools.hpp:135-199)
  ImuState x;
  MatX cov = MatX(15, 15);
};

// IMUST::operator- (tools.hpp:164-173): a - b
inline void state_minus(const ImuState& a, const ImuState& b, double out[15]) {
  V3 r = Log(transpose(b.R) * a.R), dp = a.p - b.p, dv = a.v - b.v, dbg = a.bg - b.bg, dba = a.ba - b.ba;
  for (int k = 0; k < 3; k++) { out[k] = r[k]; out[3 + k] = dp[k]; out[6 + k] = dv[k]; out[9 + k] = dbg[k]; out[12 + k] = dba[k]; }
}
// IMUST::operator+= (tools.hpp:154-162)
inline void state_plus(ImuState& a, const double d[15]) {
  a.R = a.R * Exp(v3(d[0], d[1], d[2]));
  for (int k = 0; k < 3; k++) { a.p[k] += d[3 + k]; a.v[k] += d[6 + k]; a.bg[k] += d[9 + k]; a.ba[k] += d[12 + k]; }
}

struct LioSweep {   // what one pass over the scan produces (voxelslam.cpp:873-923)
  double HTH[36];   // column-major 6x6
  double HTz[6];
  double nnt[9];    // column-major 3x3
  int match_num;
};

struct LioResult {
  bool ok = false;          // nnt's smallest eigenvalue >= 14 (voxelslam.cpp:951-957)
  int iterations = 0, match_num = 0;
  double min_eig = 0;
  std::vector<LioSweep> sweeps;
  std::vector<int> plane_of_point;   // of the last sweep; -1 = unmatched
  std::vector<double> sigma_of_point;
};

// one sweep of voxelslam.cpp:873-919 with the node cache `octos`
inline void lio_sweep(PlaneMap& map, const std::vector<PointVar>& pts, const ImuState& x, const M3& rot_var, const M3& tsl_var, std::vector<OctoNode*>& octos,
                      LioSweep& out, std::vector<int>* plane_of_point = nullptr, std::vector<double>* sigma_of_point = nullptr) {
  for (double& v : out.HTH) v = 0;
  for (double& v : out.HTz) v = 0;
  for (double& v : out.nnt) v = 0;
  out.match_num = 0;
  const int psize = (int)pts.size();
  for (int i = 0; i < psize; i++) {
    const PointVar& pv = pts[i];
    M3 phat = hat(pv.pnt);
    M3 var_world = x.R * pv.var * transpose(x.R) + phat * rot_var * transpose(phat) + tsl_var;
    V3 wld = x.R * pv.pnt + x.p;
    double sigma_d = 0;
    Plane* pla = nullptr;
    int flag = 0;
    if (octos[i] != nullptr && octos[i]->inside(wld))
      flag = octos[i]->match(wld, pla, var_world, sigma_d, octos[i]);
    else
      flag = map.match(wld, pla, var_world, sigma_d, octos[i]);
    if (plane_of_point) { (*plane_of_point)[i] = flag ? pla->id : -1; (*sigma_of_point)[i] = flag ? sigma_d : 0.0; }
    if (flag) {
      Plane& pp = *pla;
      double R_inv = 1.0 / (0.0005 + sigma_d);
      double resi = dot(pp.normal, wld - pp.center);
      V3 jh = phat * (transpose(x.R) * pp.normal);
      double jac[6] = {jh[0], jh[1], jh[2], pp.normal[0], pp.normal[1], pp.normal[2]};
      for (int c = 0; c < 6; c++)
        for (int r = 0; r < 6; r++) out.HTH[6 * c + r] += R_inv * jac[r] * jac[c];
      for (int r = 0; r < 6; r++) out.HTz[r] -= R_inv * jac[r] * resi;
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) out.nnt[3 * c + r] += pp.normal[r] * pp.normal[c];
      out.match_num++;
    }
  }
}

// voxelslam.cpp:855-958.  `st` is x_curr (in/out, including cov).
inline LioResult lio_state_estimation(PlaneMap& map, const std::vector<PointVar>& pts, LioState& st) {
  LioResult res;
  const int DIM = 15;
  const ImuState x_prop = st.x;
  const int num_max_iter = 4;
  bool EKF_stop_flg = 0, flg_EKF_converged = 0;
  MatX G(DIM, DIM), H_T_H(DIM, DIM);
  int rematch_num = 0;
  const int psize = (int)pts.size();
  std::vector<OctoNode*> octos(psize, nullptr);
  res.plane_of_point.assign(psize, -1);
  res.sigma_of_point.assign(psize, 0.0);
  LioSweep sw;
  const MatX cov_inv = mat_inverse(st.cov);
  for (int iterCount = 0; iterCount < num_max_iter; iterCount++) {
    M3 rot_var, tsl_var;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) { rot_var(r, c) = st.cov(r, c); tsl_var(r, c) = st.cov(3 + r, 3 + c); }
    lio_sweep(map, pts, st.x, rot_var, tsl_var, octos, sw, &res.plane_of_point, &res.sigma_of_point);
    res.sweeps.push_back(sw);
    res.match_num = sw.match_num;
    for (int c = 0; c < 6; c++)
      for (int r = 0; r < 6; r++) H_T_H(r, c) = sw.HTH[6 * c + r];
    MatX S(DIM, DIM);
    for (int c = 0; c < DIM; c++)
      for (int r = 0; r < DIM; r++) S(r, c) = H_T_H(r, c) + cov_inv(r, c);
    const MatX K_1 = mat_inverse(S);
    for (int r = 0; r < DIM; r++)
      for (int c = 0; c < 6; c++) {
        double t = 0;
        for (int k = 0; k < 6; k++) t += K_1(r, k) * sw.HTH[6 * c + k];
        G(r, c) = t;
      }
    double vec[15], solution[15];
    state_minus(x_prop, st.x, vec);
    for (int r = 0; r < DIM; r++) {
      double t = 0;
      for (int k = 0; k < 6; k++) t += K_1(r, k) * sw.HTz[k];
      t += vec[r];
      for (int k = 0; k < 6; k++) t -= G(r, k) * vec[k];
      solution[r] = t;
    }
    state_plus(st.x, solution);
    V3 rot_add = v3(solution[0], solution[1], solution[2]), tra_add = v3(solution[3], solution[4], solution[5]);
    EKF_stop_flg = false;
    flg_EKF_converged = false;
    if ((norm(rot_add) * 57.3 < 0.01) && (norm(tra_add) * 100 < 0.015)) flg_EKF_converged = true;
    if (flg_EKF_converged || ((rematch_num == 0) && (iterCount == num_max_iter - 2))) rematch_num++;
    res.iterations = iterCount + 1;
    if (rematch_num >= 2 || (iterCount == num_max_iter - 1)) {
      MatX nc(DIM, DIM);
      for (int r = 0; r < DIM; r++)
        for (int c = 0; c < DIM; c++) {
          double t = 0;
          for (int k = 0; k < DIM; k++) t += ((r == k ? 1.0 : 0.0) - G(r, k)) * st.cov(k, c);
          nc(r, c) = t;
        }
      st.cov = nc;
      EKF_stop_flg = true;
    }
    if (EKF_stop_flg) break;
  }
  M3 nnt;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) nnt(r, c) = sw.nnt[3 * c + r];
  V3 ev; M3 evec;
  eig_sym3(nnt, ev, evec);
  res.min_eig = ev[0];
  res.ok = !(ev[0] < 14);
  return res;
}

// voxelslam.hpp:203-215
inline void pvec_update(const std::vector<PointVar>& pts, const ImuState& x, const M3& rot_var, const M3& tsl_var, std::vector<PointVar>& world) {
  world.resize(pts.size());
  for (size_t i = 0; i < pts.size(); i++) {
    M3 phat = hat(pts[i].pnt);
    world[i].var = x.R * pts[i].var * transpose(x.R) + phat * rot_var * transpose(phat) + tsl_var;
    world[i].pnt = x.R * pts[i].pnt + x.p;
  }
}

// Bf_var (voxel_map.hpp:91-106): 9x9 column-major
inline void bf_var(const M3& var, const V3& vec, double bcov[81]) {
  double Bi[6][3] = {{2 * vec[0], 0, 0}, {vec[1], vec[0], 0}, {vec[2], 0, vec[0]}, {0, 2 * vec[1], 0}, {0, vec[2], vec[1]}, {0, 0, 2 * vec[2]}};
  double Biup[6][3];
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 3; c++) { double t = 0; for (int k = 0; k < 3; k++) t += Bi[r][k] * var(k, c); Biup[r][c] = t; }
  for (int i = 0; i < 81; i++) bcov[i] = 0;
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) { double t = 0; for (int k = 0; k < 3; k++) t += Biup[r][k] * Bi[c][k]; bcov[9 * c + r] = t; }
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 3; c++) { bcov[9 * (6 + c) + r] = Biup[r][c]; bcov[9 * r + (6 + c)] = Biup[r][c]; }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) bcov[9 * (6 + c) + (6 + r)] = var(r, c);
}

// OctoTree::plane_update (voxel_map.hpp:1118-1146) from the world cluster (P, v, N), its eigen-decomposition and cov_add.
inline void plane_update(const M3& /*P*/, const V3& v, double N, const V3& eig_value, const M3& eig_vector, const double cov_add[81], double center[3], double normal[3],
                         double plane_var[36], float& radius) {
  V3 c = v / N;
  const int l = 0;
  V3 u[3] = {col(eig_vector, 0), col(eig_vector, 1), col(eig_vector, 2)};
  double nv = 1.0 / N;
  double u_c[3][9];
  for (auto& row : u_c) for (double& x : row) x = 0;
  for (int k = 0; k < 3; k++)
    if (k != l) {
      M3 ukl = outer(u[k], u[l]);
      double fkl[9] = {ukl(0, 0), ukl(1, 0) + ukl(0, 1), ukl(2, 0) + ukl(0, 2), ukl(1, 1), ukl(1, 2) + ukl(2, 1), ukl(2, 2), 0, 0, 0};
      V3 tail = -1.0 * (dot(u[k], c) * u[l] + dot(u[l], c) * u[k]);
      fkl[6] = tail[0]; fkl[7] = tail[1]; fkl[8] = tail[2];
      const double s = nv / (eig_value[l] - eig_value[k]);
      for (int r = 0; r < 3; r++)
        for (int q = 0; q < 9; q++) u_c[r][q] += s * u[k][r] * fkl[q];
    }
  double Jc[3][9];
  for (int r = 0; r < 3; r++)
    for (int q = 0; q < 9; q++) { double t = 0; for (int k = 0; k < 9; k++) t += u_c[r][k] * cov_add[9 * q + k]; Jc[r][q] = t; }
  for (int i = 0; i < 36; i++) plane_var[i] = 0;
  for (int r = 0; r < 3; r++)
    for (int q = 0; q < 3; q++) {
      double t = 0;
      for (int k = 0; k < 9; k++) t += Jc[r][k] * u_c[q][k];
      plane_var[6 * q + r] = t;                                  // block (0,0)
      const double jn = nv * Jc[r][6 + q];
      plane_var[6 * (3 + q) + r] = jn;                           // block (0,3)
      plane_var[6 * r + (3 + q)] = jn;                           // block (3,0) = transpose
      plane_var[6 * (3 + q) + (3 + r)] = nv * nv * cov_add[9 * (6 + q) + (6 + r)];   // block (3,3)
    }
  for (int k = 0; k < 3; k++) { center[k] = c[k]; normal[k] = u[0][k]; }
  radius = (float)eig_value[2];
}

// down_sampling_voxel (tools.hpp:201-238): running mean per voxel in cloud order, float arithmetic.  Upstream emits the voxels in
// unordered_map iteration order; this restatement emits them in ascending (x, y, z) voxel index so that outputs can be compared.
inline void down_sampling_voxel(std::vector<float>& xyz, double voxel_size) {
  if (voxel_size < 0.001) return;
  struct Acc { float x, y, z, curvature; };
  std::unordered_map<LocKey, Acc, LocHash> feat_map;
  float loc_xyz[3];
  const size_t n = xyz.size() / 3;
  for (size_t i = 0; i < n; i++) {
    const float* p_c = &xyz[3 * i];
    for (int j = 0; j < 3; j++) {
      loc_xyz[j] = p_c[j] / voxel_size;
      if (loc_xyz[j] < 0) loc_xyz[j] -= 1.0;
    }
    LocKey position{(int64_t)loc_xyz[0], (int64_t)loc_xyz[1], (int64_t)loc_xyz[2]};
    auto iter = feat_map.find(position);
    if (iter == feat_map.end()) {
      feat_map[position] = Acc{p_c[0], p_c[1], p_c[2], 1};
    } else {
      Acc& pp = iter->second;
      pp.x = (pp.x * pp.curvature + p_c[0]) / (pp.curvature + 1);
      pp.y = (pp.y * pp.curvature + p_c[1]) / (pp.curvature + 1);
      pp.z = (pp.z * pp.curvature + p_c[2]) / (pp.curvature + 1);
      pp.curvature += 1;
    }
  }
  std::vector<std::pair<LocKey, Acc>> cells(feat_map.begin(), feat_map.end());
  std::sort(cells.begin(), cells.end(), [](const std::pair<LocKey, Acc>& a, const std::pair<LocKey, Acc>& b) {
    if (a.first.x != b.first.x) return a.first.x < b.first.x;
    if (a.first.y != b.first.y) return a.first.y < b.first.y;
    return a.first.z < b.first.z;
  });
  xyz.clear();
  for (auto& c : cells) { xyz.push_back(c.second.x); xyz.push_back(c.second.y); xyz.push_back(c.second.z); }
}

}  // namespace vxo
