// Drives include/vxba_li_optimizer.hpp the way voxelslam.cpp:1645-1653 drives `LI_BA_Optimizer`, with stand-ins for the
// Eigen / reference types.  Input: flat binary window (scene + states + preintegrated IMU factors); output: states + dbg/dba.
#include <cstdio>
#include <deque>
#include <vector>

#include "vxba_lidar_factor.hpp"
#include "vxba_li_optimizer.hpp"

struct Vec3 { double d[3] = {0, 0, 0}; double& operator[](int i) { return d[i]; } const double& operator[](int i) const { return d[i]; } };
struct Mat3 {
  double d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double& operator()(int r, int c) { return d[3 * c + r]; }
  const double& operator()(int r, int c) const { return d[3 * c + r]; }
};
struct Mat15 {
  double d[225] = {0};
  double& operator()(int r, int c) { return d[15 * c + r]; }
  const double& operator()(int r, int c) const { return d[15 * c + r]; }
};
struct MatX {
  int n = 0; std::vector<double> a;
  void resize(int r, int) { n = r; a.assign((size_t)r * r, 0.0); }
  double* data() { return a.data(); }
  double& operator()(int r, int c) { return a[(size_t)c * n + r]; }
};
struct VecX { std::vector<double> a; double* data() { return a.data(); } };
struct PointCluster { Mat3 P; Vec3 v; int N = 0; };
struct IMUST { double t = 0; Mat3 R; Vec3 p, v, bg, ba, g; };                       // tools.hpp:135-199
struct IMU_PRE {                                                                    // preintegration.hpp:11-30
  Mat3 R_delta; Vec3 p_delta, v_delta, bg, ba;
  Mat3 R_bg, p_bg, p_ba, v_bg, v_ba;
  double dtime = 0;
  Vec3 dbg, dba, dbg_buf, dba_buf;
  Mat15 cov;
};

using LidarFactor = vxba::LidarFactorT<PointCluster, IMUST, Vec3, Mat3, MatX, VecX>;
using LI_BA_Optimizer = vxba::LI_BA_OptimizerT<IMUST, IMU_PRE, MatX, VecX, LidarFactor>;

static PointCluster unpack(const double* c) {
  PointCluster pc;
  pc.P(0, 0) = c[0]; pc.P(0, 1) = pc.P(1, 0) = c[1]; pc.P(0, 2) = pc.P(2, 0) = c[2];
  pc.P(1, 1) = c[3]; pc.P(1, 2) = pc.P(2, 1) = c[4]; pc.P(2, 2) = c[5];
  pc.v[0] = c[6]; pc.v[1] = c[7]; pc.v[2] = c[8]; pc.N = (int)c[9];
  return pc;
}
static void m33(const double* o, Mat3& m) { for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) m(r, c) = o[3 * c + r]; }
static void v3(const double* o, Vec3& v) { for (int k = 0; k < 3; k++) v[k] = o[k]; }

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* fi = std::fopen(argv[1], "rb");
  if (!fi) return 2;
  double hdr[2];
  if (std::fread(hdr, 8, 2, fi) != 2) return 2;
  const int W = (int)hdr[0], V = (int)hdr[1];
  std::vector<double> clusters((size_t)V * W * 10), fix((size_t)V * 10), coe(V), st((size_t)W * 24), im((size_t)(W - 1) * 304);
  if (std::fread(clusters.data(), 8, clusters.size(), fi) != clusters.size() || std::fread(fix.data(), 8, fix.size(), fi) != fix.size() ||
      std::fread(coe.data(), 8, coe.size(), fi) != coe.size() || std::fread(st.data(), 8, st.size(), fi) != st.size() ||
      std::fread(im.data(), 8, im.size(), fi) != im.size()) return 2;
  std::fclose(fi);
  try {
    LidarFactor voxhess(W);
    for (int a = 0; a < V; a++) {
      std::vector<PointCluster> pcrs(W);
      for (int i = 0; i < W; i++) pcrs[i] = unpack(&clusters[((size_t)a * W + i) * 10]);
      PointCluster pcr_fix = unpack(&fix[(size_t)a * 10]), pcr_add;
      Vec3 ev; Mat3 evec;
      voxhess.push_voxel(pcrs, pcr_fix, coe[a], ev, evec, pcr_add);
    }
    std::vector<IMUST> x_buf(W);
    for (int i = 0; i < W; i++) {
      const double* s = &st[(size_t)24 * i];
      m33(s, x_buf[i].R); v3(s + 9, x_buf[i].p); v3(s + 12, x_buf[i].v); v3(s + 15, x_buf[i].bg); v3(s + 18, x_buf[i].ba); v3(s + 21, x_buf[i].g);
    }
    std::vector<IMU_PRE> store(W - 1);
    std::deque<IMU_PRE*> imu_pre_buf;
    for (int i = 0; i < W - 1; i++) {
      const double* b = &im[(size_t)304 * i];
      IMU_PRE& f = store[i];
      m33(b, f.R_delta); v3(b + 9, f.p_delta); v3(b + 12, f.v_delta); v3(b + 15, f.bg); v3(b + 18, f.ba);
      m33(b + 21, f.R_bg); m33(b + 30, f.p_bg); m33(b + 39, f.p_ba); m33(b + 48, f.v_bg); m33(b + 57, f.v_ba);
      f.dtime = b[66];
      for (int k = 0; k < 225; k++) f.cov.d[k] = b[79 + k];
      imu_pre_buf.push_back(&f);
    }
    double r0 = 0;
    voxhess.evaluate_only_residual(x_buf, 0, V, r0);     // seeds the cache, as recut's eig does upstream
    MatX hess;
    LI_BA_Optimizer opt_lsv;
    opt_lsv.damping_iter(x_buf, voxhess, imu_pre_buf, &hess);

    std::vector<double> out;
    for (int i = 0; i < W; i++) {
      for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) out.push_back(x_buf[i].R(r, c));
      for (int k = 0; k < 3; k++) out.push_back(x_buf[i].p[k]);
      for (int k = 0; k < 3; k++) out.push_back(x_buf[i].v[k]);
      for (int k = 0; k < 3; k++) out.push_back(x_buf[i].bg[k]);
      for (int k = 0; k < 3; k++) out.push_back(x_buf[i].ba[k]);
    }
    for (int i = 0; i < W - 1; i++) for (int k = 0; k < 3; k++) out.push_back(store[i].dbg[k]);
    out.push_back(hess(20, 20)); out.push_back((double)opt_lsv.imu_leng);
    FILE* fo = std::fopen(argv[2], "wb");
    std::fwrite(out.data(), 8, out.size(), fo);
    std::fclose(fo);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
