// Drives include/vxba_lio_estimator.hpp the way voxelslam.cpp drives lio_state_estimation, with stand-ins for the Eigen / reference types:
// builds a pointer octree (unordered_map of roots, eight children per node) from a flat leaf list, lets the adapter flatten it again by
// walking it, uploads a scan and runs the estimation.  Input / output: flat binary files.
#include <cstdio>
#include <map>
#include <memory>
#include <vector>

#include "vxba_lio_estimator.hpp"

struct Vec3 { double d[3] = {0, 0, 0}; double& operator[](int i) { return d[i]; } const double& operator[](int i) const { return d[i]; } };
struct Mat3 { double d[9] = {0}; double& operator()(int r, int c) { return d[3 * c + r]; } const double& operator()(int r, int c) const { return d[3 * c + r]; } };
struct Mat6 { double d[36] = {0}; double& operator()(int r, int c) { return d[6 * c + r]; } const double& operator()(int r, int c) const { return d[6 * c + r]; } };
struct Mat15 { double d[225] = {0}; double& operator()(int r, int c) { return d[15 * c + r]; } const double& operator()(int r, int c) const { return d[15 * c + r]; } };
struct IMUST { double t = 0; Mat3 R; Vec3 p, v, bg, ba, g; Mat15 cov; };                 // tools.hpp:135-199
struct pointVar { Vec3 pnt; Mat3 var; };                                                  // voxel_map.hpp:14-19
struct Plane { Vec3 center, normal; Mat6 plane_var; float radius = 0; bool is_plane = false; };   // voxel_map.hpp:66-80
struct OctoTree { int layer = 0, octo_state = 0; OctoTree* leaves[8] = {nullptr}; Plane plane; };
struct VOXEL_LOC { int64_t x, y, z; bool operator<(const VOXEL_LOC& o) const { return x != o.x ? x < o.x : (y != o.y ? y < o.y : z < o.z); } };

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* fi = std::fopen(argv[1], "rb");
  if (!fi) return 2;
  double hdr[4];
  if (std::fread(hdr, 8, 4, fi) != 4) return 2;
  const double voxel_size = hdr[0];
  const int max_layer = (int)hdr[1];
  const long n_leaf = (long)hdr[2], n_pts = (long)hdr[3];
  std::vector<double> leaf((size_t)n_leaf * 49), pts((size_t)n_pts * 12), st(24 + 225);   // leaf: loc3 layer path is_plane center3 normal3 var36 radius
  if (std::fread(leaf.data(), 8, leaf.size(), fi) != leaf.size() || std::fread(pts.data(), 8, pts.size(), fi) != pts.size() || std::fread(st.data(), 8, st.size(), fi) != st.size()) return 2;
  std::fclose(fi);
  // the map as the reference holds it
  std::map<VOXEL_LOC, OctoTree*> surf_map;
  std::vector<std::unique_ptr<OctoTree>> pool;
  auto make = [&](int layer) { pool.emplace_back(new OctoTree()); pool.back()->layer = layer; return pool.back().get(); };
  for (long i = 0; i < n_leaf; i++) {
    const double* L = &leaf[(size_t)i * 49];
    VOXEL_LOC loc{(int64_t)L[0], (int64_t)L[1], (int64_t)L[2]};
    const int layer = (int)L[3], path = (int)L[4];
    OctoTree*& root = surf_map[loc];
    if (!root) root = make(0);
    OctoTree* nd = root;
    for (int l = 0; l < layer; l++) {
      const int k = (path >> (3 * l)) & 7;
      nd->octo_state = 1;
      if (!nd->leaves[k]) nd->leaves[k] = make(l + 1);
      nd = nd->leaves[k];
    }
    nd->plane.is_plane = L[5] != 0;
    for (int k = 0; k < 3; k++) { nd->plane.center[k] = L[6 + k]; nd->plane.normal[k] = L[9 + k]; }
    for (int c = 0; c < 6; c++) for (int r = 0; r < 6; r++) nd->plane.plane_var(r, c) = L[12 + 6 * c + r];
    nd->plane.radius = (float)L[48];
  }
  vxba::LioEstimatorT<IMUST, pointVar> est(voxel_size, max_layer);
  for (auto& kv : surf_map) est.stage_voxel(kv.first, kv.second);
  est.flush_map();
  std::vector<pointVar> pvec(n_pts);
  for (long i = 0; i < n_pts; i++) {
    for (int k = 0; k < 3; k++) pvec[i].pnt[k] = pts[(size_t)12 * i + k];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) pvec[i].var(r, c) = pts[(size_t)12 * i + 3 + 3 * c + r];
  }
  est.set_scan(pvec);
  IMUST x;
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) x.R(r, c) = st[3 * c + r];
  for (int k = 0; k < 3; k++) { x.p[k] = st[9 + k]; x.v[k] = st[12 + k]; x.bg[k] = st[15 + k]; x.ba[k] = st[18 + k]; x.g[k] = st[21 + k]; }
  for (int c = 0; c < 15; c++) for (int r = 0; r < 15; r++) x.cov(r, c) = st[24 + 15 * c + r];
  const bool ok = est.lio_state_estimation(x);
  std::vector<pointVar> wv; std::vector<Vec3> pwld;
  est.pvec_update(x, wv, pwld);
  FILE* fo = std::fopen(argv[2], "wb");
  if (!fo) return 2;
  std::vector<double> out;
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) out.push_back(x.R(r, c));
  for (int k = 0; k < 3; k++) out.push_back(x.p[k]);
  for (int k = 0; k < 3; k++) out.push_back(x.v[k]);
  for (int k = 0; k < 3; k++) out.push_back(x.bg[k]);
  for (int k = 0; k < 3; k++) out.push_back(x.ba[k]);
  for (int c = 0; c < 15; c++) for (int r = 0; r < 15; r++) out.push_back(x.cov(r, c));
  out.push_back(ok ? 1.0 : 0.0); out.push_back(est.match_num); out.push_back(est.iterations);
  for (int k = 0; k < 3; k++) out.push_back(pwld[0][k]);
  out.push_back(wv[n_pts - 1].var(2, 1));
  // cut_voxel's increments for two leaves holding scan points {4, 2, 9} and {0 .. 6}, covariances left on the device
  std::vector<Vec3> pw2;
  est.pvec_update_points(x, pw2);
  std::vector<int64_t> cell_ptr = {0, 3, 10};
  std::vector<int32_t> order = {4, 2, 9, 0, 1, 2, 3, 4, 5, 6};
  std::vector<double> cl, ca;
  est.leaf_stats(cell_ptr, order, cl, ca);
  out.push_back(pw2[0][1] == pwld[0][1] ? 1.0 : 0.0);
  for (int k = 0; k < 20; k++) out.push_back(cl[k]);
  out.push_back(ca[0]); out.push_back(ca[81 + 9 * 7 + 2]);
  std::fwrite(out.data(), 8, out.size(), fo);
  std::fclose(fo);
  return 0;
}
