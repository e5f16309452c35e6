// Host-side (CPU, O(W^3)) part of the LM shell: what stays on the host between the two GPU sweeps of
// Lidar_BA_Optimizer::damping_iter (voxel_map.hpp:367-442): gauge fix, damped solve, trial-state update,
// gain ratio and damping schedule.  Dense math on (6W)^2 systems, W <= 10 -> 60x60.
#pragma once
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

namespace vxh {

// Dense solve of the symmetric system A x = b by LDL^T with symmetric diagonal pivoting (the reference uses Eigen's LDLT,
// voxel_map.hpp:403, 597, 811).  A must hold the FULL symmetric matrix (both triangles); it is overwritten.  The factor is
// built in the triangle whose rows are contiguous in memory (for a symmetric column-major matrix the upper triangle read
// row-wise IS the lower triangle), so every inner product below walks unit-stride memory; four independent partial sums
// break the FMA dependency chain (fixed order: deterministic).
inline double dot4(const double* a, const double* b, int n) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int j = 0;
  for (; j + 3 < n; j += 4) { s0 += a[j] * b[j]; s1 += a[j + 1] * b[j + 1]; s2 += a[j + 2] * b[j + 2]; s3 += a[j + 3] * b[j + 3]; }
  for (; j < n; ++j) s0 += a[j] * b[j];
  return (s0 + s1) + (s2 + s3);
}
#if defined(__x86_64__)
// The library is built for baseline x86-64 (like the reference, VoxelSLAM/CMakeLists.txt: no -march); the one hot host loop
// -- the inner product of the 15W-dimensional LDL^T -- gets an AVX2/FMA body chosen at run time.
__attribute__((target("avx2,fma"))) inline double dot_avx2(const double* a, const double* b, int n) {
  __m256d s0 = _mm256_setzero_pd(), s1 = _mm256_setzero_pd();
  int j = 0;
  for (; j + 7 < n; j += 8) {
    s0 = _mm256_fmadd_pd(_mm256_loadu_pd(a + j), _mm256_loadu_pd(b + j), s0);
    s1 = _mm256_fmadd_pd(_mm256_loadu_pd(a + j + 4), _mm256_loadu_pd(b + j + 4), s1);
  }
  for (; j + 3 < n; j += 4) s0 = _mm256_fmadd_pd(_mm256_loadu_pd(a + j), _mm256_loadu_pd(b + j), s0);
  s0 = _mm256_add_pd(s0, s1);
  double t[4];
  _mm256_storeu_pd(t, s0);
  double s = (t[0] + t[1]) + (t[2] + t[3]);
  for (; j < n; ++j) s += a[j] * b[j];
  return s;
}
inline bool cpu_has_avx2_fma() {
  static const bool has = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
  return has;
}
// four inner products against the same vector (rows r0 .. r0+3 of a row-major matrix with leading dimension ld)
__attribute__((target("avx2,fma"))) inline void dot4rows_avx2(const double* r0, size_t ld, const double* w, int n, double out[4]) {
  __m256d s0 = _mm256_setzero_pd(), s1 = _mm256_setzero_pd(), s2 = _mm256_setzero_pd(), s3 = _mm256_setzero_pd();
  const double *a0 = r0, *a1 = r0 + ld, *a2 = r0 + 2 * ld, *a3 = r0 + 3 * ld;
  int j = 0;
  for (; j + 3 < n; j += 4) {
    const __m256d wv = _mm256_loadu_pd(w + j);
    s0 = _mm256_fmadd_pd(_mm256_loadu_pd(a0 + j), wv, s0);
    s1 = _mm256_fmadd_pd(_mm256_loadu_pd(a1 + j), wv, s1);
    s2 = _mm256_fmadd_pd(_mm256_loadu_pd(a2 + j), wv, s2);
    s3 = _mm256_fmadd_pd(_mm256_loadu_pd(a3 + j), wv, s3);
  }
  double t[4];
  _mm256_storeu_pd(t, s0); out[0] = (t[0] + t[1]) + (t[2] + t[3]);
  _mm256_storeu_pd(t, s1); out[1] = (t[0] + t[1]) + (t[2] + t[3]);
  _mm256_storeu_pd(t, s2); out[2] = (t[0] + t[1]) + (t[2] + t[3]);
  _mm256_storeu_pd(t, s3); out[3] = (t[0] + t[1]) + (t[2] + t[3]);
  for (; j < n; ++j) { out[0] += a0[j] * w[j]; out[1] += a1[j] * w[j]; out[2] += a2[j] * w[j]; out[3] += a3[j] * w[j]; }
}
inline double dotp(const double* a, const double* b, int n) { return cpu_has_avx2_fma() ? dot_avx2(a, b, n) : dot4(a, b, n); }
inline void dot4rows(const double* r0, size_t ld, const double* w, int n, double out[4]) {
  if (cpu_has_avx2_fma()) { dot4rows_avx2(r0, ld, w, n, out); return; }
  for (int q = 0; q < 4; q++) out[q] = dot4(r0 + q * ld, w, n);
}
#else
inline double dotp(const double* a, const double* b, int n) { return dot4(a, b, n); }
inline void dot4rows(const double* r0, size_t ld, const double* w, int n, double out[4]) {
  for (int q = 0; q < 4; q++) out[q] = dot4(r0 + q * ld, w, n);
}
#endif
inline void ldlt_solve_inplace(int n, double* A, const double* b, double* x, int* perm, double* work) {
  auto at = [&](int r, int c) -> double& { return A[(size_t)r * n + c]; };   // (r, c) with r >= c; row r is contiguous
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = std::fabs(at(k, k));
    for (int i = k + 1; i < n; ++i) {
      const double v = std::fabs(at(i, i));
      if (v > best) { best = v; piv = i; }
    }
    perm[k] = piv;
    if (piv != k) {
      for (int j = 0; j < k; ++j) std::swap(at(k, j), at(piv, j));
      for (int i = piv + 1; i < n; ++i) std::swap(at(i, k), at(i, piv));
      std::swap(at(k, k), at(piv, piv));
      for (int i = k + 1; i < piv; ++i) std::swap(at(i, k), at(piv, i));
    }
    const double* rk = &at(k, 0);
    for (int j = 0; j < k; ++j) work[j] = at(j, j) * rk[j];
    const double d = at(k, k) - dotp(rk, work, k);
    at(k, k) = d;
    const bool nz = std::fabs(d) > 0.0;
    int i = k + 1;
    for (; i + 3 < n; i += 4) {   // four rows share the loads of `work`
      double dots[4];
      dot4rows(&at(i, 0), (size_t)n, work, k, dots);
      for (int q = 0; q < 4; q++) { const double s = at(i + q, k) - dots[q]; at(i + q, k) = nz ? s / d : s; }
    }
    for (; i < n; ++i) {
      const double s = at(i, k) - dotp(&at(i, 0), work, k);
      at(i, k) = nz ? s / d : s;
    }
  }
  for (int i = 0; i < n; ++i) x[i] = b[i];
  for (int k = 0; k < n; ++k) std::swap(x[k], x[perm[k]]);
  for (int i = 0; i < n; ++i) x[i] -= dotp(&at(i, 0), x, i);
  const double tol = 1.0 / std::numeric_limits<double>::max();
  for (int i = 0; i < n; ++i) x[i] = (std::fabs(at(i, i)) > tol) ? x[i] / at(i, i) : 0.0;
  for (int i = n - 1; i >= 0; --i) {          // L^T x = y, column-oriented so that row i of L is read contiguously
    const double xi = x[i];
    const double* ri = &at(i, 0);
    for (int j = 0; j < i; ++j) x[j] -= ri[j] * xi;
  }
  for (int k = n - 1; k >= 0; --k) std::swap(x[k], x[perm[k]]);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Structured solve of the damped LiDAR-inertial system (LI_BA_Optimizer / LI_BA_OptimizerGravity, voxel_map.hpp:597, 811).
// Upstream hands the whole 15W (+3) system to a dense pivoted LDL^T.  Its structure is much thinner than that: the LiDAR factor only
// fills the pose-pose blocks; velocities and biases (9 unknowns per frame: the set Y) meet each other and the poses only through the
// IMU factors of CONSECUTIVE frames.  Listed frame by frame, A[Y][Y] is therefore banded (half-bandwidth 17) and a pose column
// couples to the Y rows of three frames only.  So:  band Cholesky A[Y][Y] = L L^T;  W = L^-1 A[Y][X] (forward substitution, a column
// starts at its first structural non-zero, and row a of W is non-zero on a PREFIX of the columns when X is listed by first row);
// S = A[X][X] - W^T W;  the dense part S (poses [+ gravity]: 54 or 57 unknowns at W = 10) goes to the pivoted LDL^T as before;
// y = L^-T (L^-1 b_Y - W x).  ~180k multiply-adds instead of 820k for the 135-unknown system.  The damped system is positive
// definite wherever LM accepts steps; a non-positive band pivot returns false and the caller falls back to the dense solve, whose
// pivoting and null-pivot rule are the reference's.
//   A: m x m, full symmetric storage (row- == column-major), not modified.   Y[ny], X[nx]: the two index sets (disjoint, together
//   all of 0..m-1), X sorted by xlo.   bw: half-bandwidth of A[Y][Y] in Y order.   xlo[p]: first Y position column X[p] couples to.
struct BandSchurWork {
  std::vector<double> L, Lt, invd, Wm, wrow, S, S0, wb, yb, rx, rx0, xs, work;
  std::vector<int> perm, qmax;
};
//   lda: row stride of A (>= m: A may be a window into a larger matrix).   dadd (nullable): added to the diagonal, dadd[i] for A(i, i)
//   -- the LM damping u * diag(H), so that the caller need not build H + u D.
// The solve comes in two halves.  `prepare` touches A[Y][Y], A[Y][X], b[Y] and dadd[Y] only -- in the LiDAR-inertial system those
// hold nothing but IMU terms, so the LI shells run it while the GPU is still sweeping the LiDAR Hessian -- and leaves L, W, L^-1 b_Y,
// -W^T W and -W^T L^-1 b_Y in the workspace; `finish` adds A[X][X] (+ damping) and b[X], takes the dense step and substitutes back.
// Layout of the workspace: L in row-band storage (row a holds columns a-bw .. a at [0 .. bw]) and, transposed, in column-band storage
// (Lt: column c holds rows c .. c+bw) so that both sweeps of the factorisation run down contiguous memory; W and S0 with rows padded to
// a multiple of 16 doubles (the Schur-complement loop works on 16-wide register blocks and may read / write the padding).
constexpr int BS_MAXBW = 63;
inline int band_schur_stride(int nx) { return (nx + 15) & ~15; }
template <bool AVX>
__attribute__((always_inline)) inline bool band_schur_prepare_impl(const double* A, int lda, const double* dadd, const double* b, const int* Y, int ny, int bw, const int* X, int nx,
                                  const int* xlo, BandSchurWork& ws) {
  if (bw > BS_MAXBW) return false;
  const int ld = bw + 1;
  const int nc = band_schur_stride(nx);
  ws.L.resize((size_t)ny * ld);
  ws.Lt.resize((size_t)ny * ld);
  ws.invd.resize(ny);
  ws.Wm.resize((size_t)ny * nc);            // new elements are zero; whatever an earlier call left behind is finite
  ws.S0.resize((size_t)nx * nc);
  ws.wb.resize(ny);
  ws.rx0.resize(nx);
  ws.qmax.resize(ny);
  ws.wrow.resize(nc);
  double* __restrict L = ws.L.data();
  double* __restrict Lt = ws.Lt.data();
  double* __restrict invd = ws.invd.data();
  // Band Cholesky, row by row.  Inside a row the substitution is right-looking: once L(a, c) is known it is taken out of the
  // entries to its right with column c of L (contiguous in Lt) -- independent updates instead of one dependent sum per entry; every
  // entry still receives its terms in ascending order of c.  Divisions by the pivots are multiplications by their reciprocals.
  for (int a = 0; a < ny; a++) {
    const int c0 = a - bw > 0 ? a - bw : 0, len = a - c0 + 1;
    const double* Arow = A + (size_t)Y[a] * lda;
    double v[BS_MAXBW + 1];
    for (int t = 0; t < len; t++) v[t] = Arow[Y[c0 + t]];
    if (dadd) v[len - 1] += dadd[Y[a]];
    for (int c = c0; c < a; c++) {
      const double x = v[c - c0] * invd[c];
      v[c - c0] = x;
      const double* __restrict lt = Lt + (size_t)c * ld;   // lt[t] = L(c + t, c)
      double* __restrict vr = v + (c - c0);
      const int m = a - c;
      for (int t = 1; t < m; t++) vr[t] -= x * lt[t];
      vr[m] -= x * x;                                      // the diagonal entry: L(a, c)^2
    }
    const double d = v[len - 1];
    if (!(d > 0.0)) return false;
    const double sd = std::sqrt(d);
    v[len - 1] = sd;
    invd[a] = 1.0 / sd;
    double* la = L + (size_t)a * ld + (c0 - (a - bw));
    for (int t = 0; t < len; t++) { la[t] = v[t]; Lt[(size_t)(c0 + t) * ld + (a - c0 - t)] = v[t]; }
  }
  // W = L^-1 A[Y][X] and wb = L^-1 b_Y, row by row; row a is non-zero on the columns whose first row is <= a
  {
    int q = 0;
    for (int a = 0; a < ny; a++) { while (q < nx && xlo[q] <= a) q++; ws.qmax[a] = q; }
  }
  double* __restrict Wm = ws.Wm.data();
  double* __restrict wbv = ws.wb.data();
  for (int a = 0; a < ny; a++) {
    const int c0 = a - bw > 0 ? a - bw : 0;
    const int qa = ws.qmax[a];
    double* __restrict wa = ws.wrow.data();
    const double* Arow = A + (size_t)Y[a] * lda;
    for (int q = 0; q < qa; q++) wa[q] = Arow[X[q]];
    double sb = b[Y[a]];
    const double* la = L + (size_t)a * ld - (a - bw);      // la[k] = L(a, k)
    for (int k = c0; k < a; k++) {
      const double l = la[k];
      const double* __restrict wk = Wm + (size_t)k * nc;
      const int qk = ws.qmax[k];              // qk <= qa
      for (int q = 0; q < qk; q++) wa[q] -= l * wk[q];
      sb -= l * wbv[k];
    }
    const double inv = invd[a];
    double* __restrict wout = Wm + (size_t)a * nc;
    for (int q = 0; q < qa; q++) wout[q] = wa[q] * inv;
    wbv[a] = sb * inv;
  }
  // S0 = -W^T W (lower triangle), 16 columns of one row at a time in registers, summed over the rows of W in ascending order;
  // rx0 = -W^T wb
  double* __restrict S0 = ws.S0.data();
  for (int p = 0; p < nx; p++) {
    const int lo = xlo[p] < ny ? xlo[p] : ny;
    for (int q0 = 0; q0 <= p; q0 += 16) {
      double acc[16];
      for (int t = 0; t < 16; t++) acc[t] = 0.0;
      for (int a = lo; a < ny; a++) {
        const double* __restrict wa = Wm + (size_t)a * nc;
        const double wp = wa[p];
        for (int t = 0; t < 16; t++) acc[t] -= wp * wa[q0 + t];
      }
      double* sp = S0 + (size_t)p * nc + q0;
      for (int t = 0; t < 16; t++) sp[t] = acc[t];
    }
  }
  double* __restrict rx0 = ws.rx0.data();
  for (int p = 0; p < nx; p++) rx0[p] = 0.0;
  for (int a = 0; a < ny; a++) {
    const double* __restrict wa = Wm + (size_t)a * nc;
    const int qa = ws.qmax[a];
    const double wba = wbv[a];
    for (int p = 0; p < qa; p++) rx0[p] -= wa[p] * wba;
  }
  return true;
}
template <bool AVX>
__attribute__((always_inline)) inline void band_schur_finish_impl(const double* A, int lda, const double* dadd, const double* b, const int* Y, int ny, int bw, const int* X, int nx,
                                  double* x, BandSchurWork& ws) {
  const int ld = bw + 1;
  const int nc = band_schur_stride(nx);
  double* L = ws.L.data();
  auto Lat = [&](int a, int c) -> double& { return L[(size_t)a * ld + (c - (a - bw))]; };
  // S = A[X][X] (+ damping) - W^T W, mirrored; rx = b_X - W^T wb
  ws.S.resize((size_t)nx * nx);
  ws.rx.resize(nx);
  double* S = ws.S.data();
  const double* S0 = ws.S0.data();
  for (int p = 0; p < nx; p++) {
    const double* Arow = A + (size_t)X[p] * lda;
    for (int q = 0; q <= p; q++) S[(size_t)p * nx + q] = Arow[X[q]] + S0[(size_t)p * nc + q];
    if (dadd) S[(size_t)p * nx + p] = (Arow[X[p]] + dadd[X[p]]) + S0[(size_t)p * nc + p];
    ws.rx[p] = b[X[p]] + ws.rx0[p];
  }
  for (int p = 0; p < nx; p++)
    for (int q = 0; q < p; q++) S[(size_t)q * nx + p] = S[(size_t)p * nx + q];
  // dense part: the reference's pivoted LDL^T
  ws.xs.assign(nx, 0.0); ws.work.assign(nx, 0.0); ws.perm.assign(nx, 0);
  ldlt_solve_inplace(nx, S, ws.rx.data(), ws.xs.data(), ws.perm.data(), ws.work.data());
  for (int p = 0; p < nx; p++) x[X[p]] = ws.xs[p];
  // y = L^-T (wb - W x)
  ws.yb.resize(ny);
  for (int a = 0; a < ny; a++) {
    const double* wa = ws.Wm.data() + (size_t)a * nc;
    double sacc = ws.wb[a];
    for (int q = 0; q < ws.qmax[a]; q++) sacc -= wa[q] * ws.xs[q];
    ws.yb[a] = sacc;
  }
  for (int a = ny - 1; a >= 0; a--) {
    const double ya = ws.yb[a] / Lat(a, a);
    ws.yb[a] = ya;
    const int c0 = a - bw > 0 ? a - bw : 0;
    for (int k = c0; k < a; k++) ws.yb[k] -= Lat(a, k) * ya;
  }
  for (int a = 0; a < ny; a++) x[Y[a]] = ws.yb[a];
}
// The second half of `finish` alone: given the solution xs of the dense part (X order) -- the LiDAR-inertial shells take it from the
// device, which solves the reduced pose system itself (vxba_solve4.hpp) -- scatter it and substitute the banded unknowns back:
// y = L^-T (L^-1 b_Y - W x).  After a successful `prepare` on the same system.
inline void band_schur_finish_y(const int* Y, int ny, int bw, const int* X, int nx, const double* xs, double* x, BandSchurWork& ws) {
  const int ld = bw + 1;
  const int nc = band_schur_stride(nx);
  double* L = ws.L.data();
  auto Lat = [&](int a, int c) -> double& { return L[(size_t)a * ld + (c - (a - bw))]; };
  for (int p = 0; p < nx; p++) x[X[p]] = xs[p];
  ws.yb.resize(ny);
  for (int a = 0; a < ny; a++) {
    const double* wa = ws.Wm.data() + (size_t)a * nc;
    double sacc = ws.wb[a];
    for (int q = 0; q < ws.qmax[a]; q++) sacc -= wa[q] * xs[q];
    ws.yb[a] = sacc;
  }
  for (int a = ny - 1; a >= 0; a--) {
    const double ya = ws.yb[a] / Lat(a, a);
    ws.yb[a] = ya;
    const int c0 = a - bw > 0 ? a - bw : 0;
    for (int k = c0; k < a; k++) ws.yb[k] -= Lat(a, k) * ya;
  }
  for (int a = 0; a < ny; a++) x[Y[a]] = ws.yb[a];
}
#if defined(__x86_64__)
__attribute__((target("avx2,fma"))) inline bool band_schur_prepare_avx2(const double* A, int lda, const double* dadd, const double* b, const int* Y, int ny, int bw,
                                                                        const int* X, int nx, const int* xlo, BandSchurWork& ws) {
  return band_schur_prepare_impl<true>(A, lda, dadd, b, Y, ny, bw, X, nx, xlo, ws);
}
__attribute__((target("avx2,fma"))) inline void band_schur_finish_avx2(const double* A, int lda, const double* dadd, const double* b, const int* Y, int ny, int bw,
                                                                       const int* X, int nx, double* x, BandSchurWork& ws) {
  band_schur_finish_impl<true>(A, lda, dadd, b, Y, ny, bw, X, nx, x, ws);
}
#endif
// false: a band pivot was not positive (nothing usable in the workspace; the caller takes the dense solve)
inline bool band_schur_prepare(const double* A, int lda, const double* dadd, const double* b, const int* Y, int ny, int bw, const int* X, int nx, const int* xlo,
                               BandSchurWork& ws) {
#if defined(__x86_64__)
  if (cpu_has_avx2_fma()) return band_schur_prepare_avx2(A, lda, dadd, b, Y, ny, bw, X, nx, xlo, ws);
#endif
  return band_schur_prepare_impl<false>(A, lda, dadd, b, Y, ny, bw, X, nx, xlo, ws);
}
// after a successful prepare on the same A[Y][.], b[Y], dadd[Y]
inline void band_schur_finish(const double* A, int lda, const double* dadd, const double* b, const int* Y, int ny, int bw, const int* X, int nx, double* x,
                              BandSchurWork& ws) {
#if defined(__x86_64__)
  if (cpu_has_avx2_fma()) return band_schur_finish_avx2(A, lda, dadd, b, Y, ny, bw, X, nx, x, ws);
#endif
  band_schur_finish_impl<false>(A, lda, dadd, b, Y, ny, bw, X, nx, x, ws);
}
inline bool band_schur_solve(int m, const double* A, int lda, const double* dadd, const double* b, const int* Y, int ny, int bw, const int* X, int nx,
                             const int* xlo, double* x, BandSchurWork& ws) {
  (void)m;
  if (!band_schur_prepare(A, lda, dadd, b, Y, ny, bw, X, nx, xlo, ws)) return false;
  band_schur_finish(A, lda, dadd, b, Y, ny, bw, X, nx, x, ws);
  return true;
}

// Index sets of the LiDAR-inertial system for band_schur_solve.  The system handed in holds frames f0 .. W-1 with 15 unknowns each
// (rotation 3, position 3 | velocity 3, gyro bias 3, accelerometer bias 3), optionally preceded by `lead_y` loose velocity/bias unknowns
// of frame f0 - 1 (the gravity variant fixes only frame 0's POSE) and followed by `tail_x` dense unknowns (gravity).
struct LiIndexSets {
  std::vector<int> Y, X, xlo;
  int bw = 17;
};
inline LiIndexSets li_index_sets(int nframes, int lead_y, int tail_x) {
  LiIndexSets s;
  const int yoff = lead_y;   // Y positions of the lead block come first
  for (int k = 0; k < lead_y; k++) s.Y.push_back(k);
  for (int j = 0; j < nframes; j++)
    for (int k = 6; k < 15; k++) s.Y.push_back(lead_y + 15 * j + k);
  const int ny = (int)s.Y.size();
  // dense unknowns sorted by their first Y row: the tail block (couples to everything) first, then the poses frame by frame
  for (int k = 0; k < tail_x; k++) { s.X.push_back(lead_y + 15 * nframes + k); s.xlo.push_back(0); }
  for (int j = 0; j < nframes; j++)
    for (int k = 0; k < 6; k++) {
      s.X.push_back(lead_y + 15 * j + k);
      // pose of frame j couples to the velocity/bias unknowns of frames j-1 .. j+1 (the lead block counts as frame -1)
      int lo = lead_y > 0 ? (j == 0 ? 0 : yoff + 9 * (j - 1)) : 9 * (j - 1);
      s.xlo.push_back(lo < 0 ? 0 : (lo > ny ? ny : lo));
    }
  return s;
}

// R <- R * Exp(dphi), column-major 3x3 in/out; Rodrigues with the reference's 1e-11 cut-off (tools.hpp:51-66).
inline void right_multiply_exp(const double* Rin, const double* dphi, double* Rout) {
  const double th = std::sqrt(dphi[0] * dphi[0] + dphi[1] * dphi[1] + dphi[2] * dphi[2]);
  double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major Exp(dphi)
  if (th >= 1e-11) {
    const double k0 = dphi[0] / th, k1 = dphi[1] / th, k2 = dphi[2] / th;
    const double K[9] = {0, -k2, k1, k2, 0, -k0, -k1, k0, 0};
    double KK[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) KK[3 * r + c] = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
    const double s = std::sin(th), c1 = 1.0 - std::cos(th);
    for (int q = 0; q < 9; q++) E[q] += s * K[q] + c1 * KK[q];
  }
  double out[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      // (R E)(r,c) with R column-major: R(r,k) = Rin[3k + r]
      out[3 * c + r] = Rin[r] * E[c] + Rin[3 + r] * E[3 + c] + Rin[6 + r] * E[6 + c];
    }
  std::memcpy(Rout, out, sizeof out);
}

// One damped step of voxel_map.hpp:397-410 on the LiDAR-only (6W) system.
//   Hess (n x n col-major) and JacT (n) are modified in place (gauge fix), exactly like the reference's locals.
//   Rp -> Rp_trial = Rp (+) dxi;  returns q1 = 0.5 dxi . (u D dxi - JacT).
struct LMWorkspace {
  std::vector<double> A, rhs, dxi, work;
  std::vector<int> perm;
  void resize(int n) { A.resize((size_t)n * n); rhs.resize(n); dxi.resize(n); work.resize(n); perm.resize(n); }
};

inline double lm_damped_step(int W, double* Hess, double* JacT, double u, const double* Rp, double* Rp_trial, LMWorkspace& ws) {
  const int n = 6 * W;
  ws.resize(n);
  // gauge: frame 0 fixed (voxel_map.hpp:397-400)
  for (int c = 0; c < n; c++)
    for (int r = 0; r < 6; r++) { Hess[(size_t)c * n + r] = 0.0; Hess[(size_t)r * n + c] = 0.0; }
  for (int r = 0; r < 6; r++) { Hess[(size_t)r * n + r] = 1.0; JacT[r] = 0.0; }
  // (Hess + u D) dxi = -JacT,  D = diag(Hess)  (voxel_map.hpp:402-403)
  std::memcpy(ws.A.data(), Hess, sizeof(double) * n * n);
  for (int r = 0; r < n; r++) ws.A[(size_t)r * n + r] = Hess[(size_t)r * n + r] + u * Hess[(size_t)r * n + r];
  for (int r = 0; r < n; r++) ws.rhs[r] = -JacT[r];
  ldlt_solve_inplace(n, ws.A.data(), ws.rhs.data(), ws.dxi.data(), ws.perm.data(), ws.work.data());
  // trial state (voxel_map.hpp:405-409)
  for (int j = 0; j < W; j++) {
    right_multiply_exp(Rp + 12 * j, ws.dxi.data() + 6 * j, Rp_trial + 12 * j);
    for (int k = 0; k < 3; k++) Rp_trial[12 * j + 9 + k] = Rp[12 * j + 9 + k] + ws.dxi[6 * j + 3 + k];
  }
  double q1 = 0.0;
  for (int r = 0; r < n; r++) q1 += ws.dxi[r] * (u * Hess[(size_t)r * n + r] * ws.dxi[r] - JacT[r]);
  return 0.5 * q1;
}

// Damping schedule of voxel_map.hpp:418-435.  Returns true if the step is accepted.
inline bool lm_update_damping(double residual1, double residual2, double q1, double& u, double& v) {
  double q = residual1 - residual2;
  if (q > 0) {
    const double one_three = 1.0 / 3;
    q = q / q1;
    v = 2;
    q = 1 - std::pow(2 * q - 1, 3);
    u *= (q < one_three ? one_three : q);
    return true;
  }
  u = u * v;
  v = 2 * v;
  return false;
}

}  // namespace vxh
