// Drives include/vxba_lidar_factor.hpp exactly the way voxel_map.hpp drives `LidarFactor`, with small stand-ins for the
// Eigen / reference types (Eigen is not installed in this image; the adapter only needs operator()(r,c), operator[],
// .data() column-major and .resize()).  Input: a flat binary scene written by the Python test; output: poses + resis.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vxba_lidar_factor.hpp"

struct Vec3 { double d[3] = {0, 0, 0}; double& operator[](int i) { return d[i]; } const double& operator[](int i) const { return d[i]; } };
struct Mat3 {
  double d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // column-major like Eigen::Matrix3d
  double& operator()(int r, int c) { return d[3 * c + r]; }
  const double& operator()(int r, int c) const { return d[3 * c + r]; }
};
struct MatX {
  int n = 0; std::vector<double> a;
  void resize(int r, int) { n = r; a.assign((size_t)r * r, 0.0); }
  double* data() { return a.data(); }
  double& operator()(int r, int c) { return a[(size_t)c * n + r]; }
};
struct VecX { std::vector<double> a; double* data() { return a.data(); } };
struct PointCluster { Mat3 P; Vec3 v; int N = 0; };                 // tools.hpp:304-365
struct IMUST { double t = 0; Mat3 R; Vec3 p, vel, bg, ba, g; };      // tools.hpp:135-199 (R, p are what the factor reads)

using LidarFactor = vxba::LidarFactorT<PointCluster, IMUST, Vec3, Mat3, MatX, VecX>;

static PointCluster unpack(const double* c) {
  PointCluster pc;
  pc.P(0, 0) = c[0]; pc.P(0, 1) = pc.P(1, 0) = c[1]; pc.P(0, 2) = pc.P(2, 0) = c[2];
  pc.P(1, 1) = c[3]; pc.P(1, 2) = pc.P(2, 1) = c[4]; pc.P(2, 2) = c[5];
  pc.v[0] = c[6]; pc.v[1] = c[7]; pc.v[2] = c[8]; pc.N = (int)c[9];
  return pc;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: adapter_demo scene.bin out.bin\n"); return 2; }
  FILE* fi = std::fopen(argv[1], "rb");
  if (!fi) return 2;
  double hdr[3];
  if (std::fread(hdr, 8, 3, fi) != 3) return 2;
  const int W = (int)hdr[0], V = (int)hdr[1], max_iter = (int)hdr[2];
  std::vector<double> clusters((size_t)V * W * 10), fix((size_t)V * 10), coe(V), rp((size_t)W * 12);
  if (std::fread(clusters.data(), 8, clusters.size(), fi) != clusters.size() || std::fread(fix.data(), 8, fix.size(), fi) != fix.size() ||
      std::fread(coe.data(), 8, coe.size(), fi) != coe.size() || std::fread(rp.data(), 8, rp.size(), fi) != rp.size()) return 2;
  std::fclose(fi);

  try {
    LidarFactor voxhess(W);
    // OctoTree::tras_opt (voxel_map.hpp:1308-1323): one push_voxel per plane voxel
    for (int a = 0; a < V; a++) {
      std::vector<PointCluster> pcrs(W);
      for (int i = 0; i < W; i++) pcrs[i] = unpack(&clusters[((size_t)a * W + i) * 10]);
      PointCluster pcr_fix = unpack(&fix[(size_t)a * 10]), pcr_add;
      Vec3 eig_value; Mat3 eig_vector;
      voxhess.push_voxel(pcrs, pcr_fix, coe[a], eig_value, eig_vector, pcr_add);
    }
    std::vector<IMUST> x_buf(W);
    for (int i = 0; i < W; i++) {
      for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) x_buf[i].R(r, c) = rp[12 * i + 3 * c + r];
      for (int k = 0; k < 3; k++) x_buf[i].p[k] = rp[12 * i + 9 + k];
    }
    if ((int)voxhess.plvec_voxels.size() != 0) {}  // staged voxels are not uploaded yet; size() reflects the device factor
    double r0 = 0;
    voxhess.evaluate_only_residual(x_buf, 0, V, r0);     // seeds the cache (what recut's eig does upstream)
    if ((int)voxhess.plvec_voxels.size() != V) { std::fprintf(stderr, "size mismatch\n"); return 3; }
    MatX hess;
    std::vector<double> resis;
    const bool conv = voxhess.damping_iter(x_buf, &hess, resis, max_iter);
    const double lam0_first = voxhess.eig_values[0][0];  // OctoTree::margi reads these (voxel_map.hpp:1217-1222)
    const int n_first = voxhess.pcr_adds[0].N;

    FILE* fo = std::fopen(argv[2], "wb");
    std::vector<double> out;
    for (int i = 0; i < W; i++) {
      for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) out.push_back(x_buf[i].R(r, c));
      for (int k = 0; k < 3; k++) out.push_back(x_buf[i].p[k]);
    }
    out.push_back(r0); out.push_back(resis[0]); out.push_back(resis[1]); out.push_back(conv ? 1.0 : 0.0);
    out.push_back(lam0_first); out.push_back((double)n_first); out.push_back(hess(6, 6));
    std::fwrite(out.data(), 8, out.size(), fo);
    std::fclose(fo);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
