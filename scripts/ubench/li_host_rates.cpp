// Micro-benchmark (development, round 2): host-side pieces of one LiDAR-inertial LM iteration at W = 10 -- IMU blocks, band half and
// finish of the structured solve, hess_plus -- on synthetic factors.  Build: g++ -O3 -std=c++17 -o li_host_rates li_host_rates.cpp
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../../voxel-slam_amd/csrc/vxba_imu.hpp"
#include "../../voxel-slam_amd/csrc/vxba_host.hpp"
using namespace std;
int main() {
  const int W = 10, n = 15 * W;
  mt19937 rng(1); normal_distribution<double> nd(0, 1);
  vector<double> states(vxi::STATE_LEN * W, 0.0), imus((size_t)vxi::IMU_LEN * (W - 1));
  for (int i = 0; i < W; i++) {
    double* s = &states[vxi::STATE_LEN * i];
    double w[3] = {0.01 * nd(rng), 0.01 * nd(rng), 0.02 * i};
    vxi::so3_exp(w, s);
    for (int k = 0; k < 3; k++) { s[9 + k] = 0.5 * i + 0.01 * nd(rng); s[12 + k] = 1 + 0.01 * nd(rng); s[15 + k] = 0.001 * nd(rng); s[18 + k] = 0.001 * nd(rng); }
    s[21] = 0; s[22] = 0; s[23] = -9.8;
  }
  double nm[36] = {0}, nw[36] = {0}; for (int k = 0; k < 6; k++) { nm[7*k] = k < 3 ? 1e-2 : 1e-1; nw[7*k] = k < 3 ? 1e-4 : 1e-3; }
  for (int i = 0; i < W - 1; i++) {
    double* f = &imus[(size_t)vxi::IMU_LEN * i];
    vxi::imu_init(f, &states[15], &states[18]);
    for (int k = 0; k < 20; k++) { double g[3] = {0.01*nd(rng),0.01*nd(rng),0.2}, a[3] = {0.1*nd(rng),0.1*nd(rng),9.8}; vxi::imu_add(f, g, a, 0.005, nm, nw); }
  }
  vector<double> cov((size_t)225 * (W - 1));
  printf("inv ok %d\n", (int)vxi::li_invert_covariances(W, imus.data(), cov.data()));
  vector<double> Hess((size_t)n * n), JacT(n), hs(3600, 0.0), js(60, 0.0);
  for (int i = 0; i < 60; i++) { hs[i * 60 + i] = 1e4; js[i] = nd(rng); }
  auto now = [] { return chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return chrono::duration<double, micro>(b - a).count(); };
  vxi::ImuWork w; bool ok;
  double tm = 0, ti = 0, th = 0, tp = 0, tf = 0, tr = 0; const int R = 2000;
  vxh::LiIndexSets sets = vxh::li_index_sets(W - 1, 0, 0); vxh::BandSchurWork ws;
  vector<double> rhs(n), work(n), dxi(n);
  double chk = 0;
  for (int rep = 0; rep < R; rep++) {
    auto t0 = now();
    memset(Hess.data(), 0, sizeof(double) * n * n); memset(JacT.data(), 0, sizeof(double) * n);
    auto t1 = now();
    double res = vxi::li_add_imu_blocks(W, states.data(), imus.data(), 1e-4, true, Hess.data(), JacT.data(), w, &ok, false, cov.data());
    auto t2 = now();
    const int g = 15, m = n - g;
    for (int y : sets.Y) { rhs[y] = -JacT[y + g]; work[y] = 0.01 * Hess[(size_t)(y + g) * n + y + g]; }
    bool p = vxh::band_schur_prepare(&Hess[(size_t)g * n + g], n, work.data(), rhs.data(), sets.Y.data(), (int)sets.Y.size(), sets.bw, sets.X.data(), (int)sets.X.size(), sets.xlo.data(), ws);
    auto t3 = now();
    vxi::li_hess_plus(W, Hess.data(), JacT.data(), hs.data(), js.data(), n);
    auto t4 = now();
    for (int r = 0; r < m; r++) { rhs[r] = -JacT[r + g]; work[r] = 0.01 * Hess[(size_t)(r + g) * n + r + g]; }
    vxh::band_schur_finish(&Hess[(size_t)g * n + g], n, work.data(), rhs.data(), sets.Y.data(), (int)sets.Y.size(), sets.bw, sets.X.data(), (int)sets.X.size(), dxi.data() + g, ws);
    auto t5 = now();
    double r2 = vxi::li_add_imu_blocks(W, states.data(), imus.data(), 1e-4, false, nullptr, nullptr, w, &ok, false, cov.data());
    auto t6 = now();
    chk += res + dxi[20] + p + r2;
    tm += us(t0, t1); ti += us(t1, t2); tp += us(t2, t3); th += us(t3, t4); tf += us(t4, t5); tr += us(t5, t6);
  }
  printf("memset %.1f  imu blocks(jac) %.1f  band prepare %.1f  hess_plus %.1f  finish %.1f  imu residual only %.1f us   (chk %g)\n", tm / R, ti / R, tp / R, th / R, tf / R, tr / R, chk);
}
