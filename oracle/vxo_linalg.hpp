// TEST INFRASTRUCTURE ONLY -- CPU oracle for the local-BA hot path.
// Nothing under oracle/ may be linked, imported or executed by the product path
// (voxel-slam_amd/); only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, as the checker / reported baseline.
//
// Parity status.  The reference (hku-mars/Voxel-SLAM) ships no tests, golden vectors or fixtures for this path, and its build (catkin, Eigen
// 3.3.7, PCL, ROS) does not exist here.  What does: its hot-path HEADERS compile unmodified over an API shim (oracle/shim, `make -C oracle ref`
// -> oracle/_ref/libref.so), and tests/test_ref_pin.py pins every restatement in this directory to them -- see the header of each vxo_*.hpp.
// NOT pinned that way, because Eigen itself is absent and the shim forwards to THIS file: SelfAdjointEigenSolver<Matrix3d> (eig_sym3 below),
// LDLT (ldlt_solve) and Matrix<15,15>::inverse() (vxo_imu.hpp).  Those three are pinned through mathematics (tests/test_oracle_*.py: residuals
// of the decomposition, lambda_min of raw points via numpy.linalg.eigh, finite-difference derivative checks); every on-path use of an
// eigenvector is quadratic in it and the solved step is algorithm-independent to round-off, nine orders below the 1e-4 contract.
//
// Small fixed-size linear algebra with no third-party dependency.  The reference's
// arithmetic lives in Eigen 3.3.7 (README.md:26, VoxelSLAM/CMakeLists.txt:26), which
// is not vendored.  The two non-trivial Eigen algorithms on the path are restated
// here from their published form:
//   * SelfAdjointEigenSolver<Matrix3d>(A)  -> compute(): scale by max|a_ij|, 3x3
//     tridiagonalisation by one reflector, implicit symmetric QR steps with
//     Wilkinson shift, ascending sort      (call sites voxel_map.hpp:267,1161,1242)
//   * LDLT<MatrixXd>::compute/solve: diagonal-pivoted LDL^T  (voxel_map.hpp:403,597,811)
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

namespace vxo {

struct V3 {
  double x[3];
  double& operator[](int i) { return x[i]; }
  const double& operator[](int i) const { return x[i]; }
};

// row-major 3x3: m[r][c]
struct M3 {
  double m[3][3];
  double& operator()(int r, int c) { return m[r][c]; }
  const double& operator()(int r, int c) const { return m[r][c]; }
};

inline V3 v3(double a, double b, double c) { return V3{{a, b, c}}; }
inline V3 zero3() { return V3{{0, 0, 0}}; }
inline M3 zero33() { M3 r; std::memset(&r, 0, sizeof r); return r; }
inline M3 eye33() { M3 r = zero33(); r(0,0) = r(1,1) = r(2,2) = 1.0; return r; }

inline V3 operator+(const V3& a, const V3& b) { return v3(a[0]+b[0], a[1]+b[1], a[2]+b[2]); }
inline V3 operator-(const V3& a, const V3& b) { return v3(a[0]-b[0], a[1]-b[1], a[2]-b[2]); }
inline V3 operator*(double s, const V3& a) { return v3(s*a[0], s*a[1], s*a[2]); }
inline V3 operator*(const V3& a, double s) { return v3(s*a[0], s*a[1], s*a[2]); }
inline V3 operator/(const V3& a, double s) { return v3(a[0]/s, a[1]/s, a[2]/s); }
inline double dot(const V3& a, const V3& b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
inline double norm(const V3& a) { return std::sqrt(dot(a, a)); }

inline M3 operator+(const M3& a, const M3& b) { M3 r; for (int i=0;i<3;i++) for (int j=0;j<3;j++) r(i,j)=a(i,j)+b(i,j); return r; }
inline M3 operator-(const M3& a, const M3& b) { M3 r; for (int i=0;i<3;i++) for (int j=0;j<3;j++) r(i,j)=a(i,j)-b(i,j); return r; }
inline M3 operator*(double s, const M3& a) { M3 r; for (int i=0;i<3;i++) for (int j=0;j<3;j++) r(i,j)=s*a(i,j); return r; }
inline M3 operator*(const M3& a, double s) { return s * a; }
inline M3 operator/(const M3& a, double s) { M3 r; for (int i=0;i<3;i++) for (int j=0;j<3;j++) r(i,j)=a(i,j)/s; return r; }
inline M3 operator*(const M3& a, const M3& b) {
  M3 r;
  for (int i=0;i<3;i++) for (int j=0;j<3;j++) r(i,j) = a(i,0)*b(0,j) + a(i,1)*b(1,j) + a(i,2)*b(2,j);
  return r;
}
inline V3 operator*(const M3& a, const V3& b) {
  return v3(a(0,0)*b[0]+a(0,1)*b[1]+a(0,2)*b[2], a(1,0)*b[0]+a(1,1)*b[1]+a(1,2)*b[2], a(2,0)*b[0]+a(2,1)*b[1]+a(2,2)*b[2]);
}
inline M3 transpose(const M3& a) { M3 r; for (int i=0;i<3;i++) for (int j=0;j<3;j++) r(i,j)=a(j,i); return r; }
inline M3 outer(const V3& a, const V3& b) { M3 r; for (int i=0;i<3;i++) for (int j=0;j<3;j++) r(i,j)=a[i]*b[j]; return r; }
inline V3 col(const M3& a, int c) { return v3(a(0,c), a(1,c), a(2,c)); }

// hat(v) b = v x b     (tools.hpp:93-100; SKEW_SYM_MATRX tools.hpp:11)
inline M3 hat(const V3& v) {
  M3 r = zero33();
  r(0,1) = -v[2]; r(0,2) =  v[1];
  r(1,0) =  v[2]; r(1,2) = -v[0];
  r(2,0) = -v[1]; r(2,1) =  v[0];
  return r;
}

// Rodrigues with the reference's 1e-11 cut-off   (tools.hpp:51-66)
inline M3 Exp(const V3& ang) {
  double ang_norm = norm(ang);
  if (ang_norm >= 1e-11) {
    V3 axis = ang / ang_norm;
    M3 K = hat(axis);
    return eye33() + std::sin(ang_norm) * K + ((1.0 - std::cos(ang_norm)) * K) * K;   // the reference's association: (s*K)*K
  }
  return eye33();
}

// Log map (tools.hpp:86-91) -- used by tests for rotation error only.
inline V3 Log(const M3& R) {
  double tr = R(0,0) + R(1,1) + R(2,2);
  double theta = (tr > 3.0 - 1e-6) ? 0.0 : std::acos(0.5 * (tr - 1));
  V3 K = v3(R(2,1) - R(1,2), R(0,2) - R(2,0), R(1,0) - R(0,1));
  return (std::fabs(theta) < 0.001) ? (0.5 * K) : ((0.5 * theta / std::sin(theta)) * K);
}

// ---------------------------------------------------------------------------
// Symmetric 3x3 eigen-decomposition, ascending eigenvalues, columns of `vec`
// are the unit eigenvectors.  Restates Eigen 3.3.7
// SelfAdjointEigenSolver<Matrix3d>::compute (the constructor path the reference
// uses): scaling, 3x3 tridiagonalisation, implicit QR with Wilkinson shift,
// selection sort.  Only the lower triangle of A is read, as in Eigen.
// ---------------------------------------------------------------------------
namespace detail {
struct Givens { double c, s; };
inline Givens make_givens(double p, double q) {
  Givens g;
  if (q == 0.0) { g.c = p < 0 ? -1.0 : 1.0; g.s = 0.0; }
  else if (p == 0.0) { g.c = 0.0; g.s = q < 0 ? 1.0 : -1.0; }
  else if (std::fabs(p) > std::fabs(q)) {
    double t = q / p; double u = std::sqrt(1.0 + t * t); if (p < 0) u = -u;
    g.c = 1.0 / u; g.s = -t * g.c;
  } else {
    double t = p / q; double u = std::sqrt(1.0 + t * t); if (q < 0) u = -u;
    g.s = -1.0 / u; g.c = -t * g.s;
  }
  return g;
}
}  // namespace detail

inline void eig_sym3(const M3& A, V3& val, M3& vec) {
  using detail::Givens;
  double a00 = A(0,0), a10 = A(1,0), a11 = A(1,1), a20 = A(2,0), a21 = A(2,1), a22 = A(2,2);
  double scale = 0.0;
  for (double v : {a00, a10, a11, a20, a21, a22}) scale = std::fmax(scale, std::fabs(v));
  if (scale == 0.0) scale = 1.0;
  a00 /= scale; a10 /= scale; a11 /= scale; a20 /= scale; a21 /= scale; a22 /= scale;

  double diag[3], sub[2];
  M3 Q;
  const double tiny = std::numeric_limits<double>::min();
  diag[0] = a00;
  double v1norm2 = a20 * a20;
  if (v1norm2 <= tiny) {
    diag[1] = a11; diag[2] = a22; sub[0] = a10; sub[1] = a21;
    Q = eye33();
  } else {
    double beta = std::sqrt(a10 * a10 + v1norm2);
    double invBeta = 1.0 / beta;
    double m01 = a10 * invBeta, m02 = a20 * invBeta;
    double q = 2.0 * m01 * a21 + m02 * (a22 - a11);
    diag[1] = a11 + m02 * q;
    diag[2] = a22 - m02 * q;
    sub[0] = beta;
    sub[1] = a21 - m01 * q;
    Q = zero33();
    Q(0,0) = 1.0; Q(1,1) = m01; Q(1,2) = m02; Q(2,1) = m02; Q(2,2) = -m01;
  }

  const int n = 3;
  int end = n - 1, start = 0, iter = 0;
  const int max_iter = 30;
  const double precision = 2.0 * std::numeric_limits<double>::epsilon();
  while (end > 0) {
    for (int i = start; i < end; ++i)
      if (std::fabs(sub[i]) <= (std::fabs(diag[i]) + std::fabs(diag[i+1])) * precision || std::fabs(sub[i]) <= tiny)
        sub[i] = 0.0;
    while (end > 0 && sub[end-1] == 0.0) end--;
    if (end <= 0) break;
    iter++;
    if (iter > max_iter * n) break;
    start = end - 1;
    while (start > 0 && sub[start-1] != 0.0) start--;

    // one implicit symmetric QR step on [start, end] with Wilkinson shift
    double td = (diag[end-1] - diag[end]) * 0.5;
    double e = sub[end-1];
    double mu = diag[end];
    if (td == 0.0) mu -= std::fabs(e);
    else {
      double e2 = e * e;
      double h = std::hypot(td, e);
      if (e2 == 0.0) mu -= (e / (td + (td > 0 ? 1.0 : -1.0))) * (e / h);
      else mu -= e2 / (td + (td > 0 ? h : -h));
    }
    double x = diag[start] - mu;
    double z = sub[start];
    for (int k = start; k < end; ++k) {
      Givens rot = detail::make_givens(x, z);
      double sdk = rot.s * diag[k] + rot.c * sub[k];
      double dkp1 = rot.s * sub[k] + rot.c * diag[k+1];
      diag[k] = rot.c * (rot.c * diag[k] - rot.s * sub[k]) - rot.s * (rot.c * sub[k] - rot.s * diag[k+1]);
      diag[k+1] = rot.s * sdk + rot.c * dkp1;
      sub[k] = rot.c * sdk - rot.s * dkp1;
      if (k > start) sub[k-1] = rot.c * sub[k-1] - rot.s * z;
      x = sub[k];
      if (k < end - 1) { z = -rot.s * sub[k+1]; sub[k+1] = rot.c * sub[k+1]; }
      // Q = Q * G(k, k+1)
      for (int r = 0; r < 3; ++r) {
        double xi = Q(r,k), yi = Q(r,k+1);
        Q(r,k)   = rot.c * xi - rot.s * yi;
        Q(r,k+1) = rot.s * xi + rot.c * yi;
      }
    }
  }
  // ascending selection sort, swapping eigenvector columns
  for (int i = 0; i < n - 1; ++i) {
    int k = i;
    for (int j = i + 1; j < n; ++j) if (diag[j] < diag[k]) k = j;
    if (k != i) {
      std::swap(diag[i], diag[k]);
      for (int r = 0; r < 3; ++r) std::swap(Q(r,i), Q(r,k));
    }
  }
  for (int i = 0; i < 3; ++i) val[i] = diag[i] * scale;
  vec = Q;
}

// ---------------------------------------------------------------------------
// Dense column-major helpers for the (6W)^2 / (15W)^2 LM systems.
// ---------------------------------------------------------------------------
struct MatX {
  int rows = 0, cols = 0;
  std::vector<double> a;  // column-major, like Eigen::MatrixXd
  MatX() {}
  MatX(int r, int c) : rows(r), cols(c), a((size_t)r * c, 0.0) {}
  void resize(int r, int c) { rows = r; cols = c; a.assign((size_t)r * c, 0.0); }
  void setZero() { std::fill(a.begin(), a.end(), 0.0); }
  double& operator()(int r, int c) { return a[(size_t)c * rows + r]; }
  const double& operator()(int r, int c) const { return a[(size_t)c * rows + r]; }
};

// x = (L D L^T with symmetric diagonal pivoting)^{-1} b.  Restates Eigen 3.3.7
// LDLT<MatrixXd, Lower>: at step k the remaining diagonal entry of largest
// magnitude is swapped into place, then the unblocked right-looking update is
// applied; solve() zeroes components whose pivot is below 1/highest().
inline std::vector<double> ldlt_solve(const MatX& Ain, const std::vector<double>& b) {
  const int n = Ain.rows;
  MatX A = Ain;
  std::vector<int> transp(n);
  for (int k = 0; k < n; ++k) {
    int piv = k; double big = std::fabs(A(k,k));
    for (int i = k + 1; i < n; ++i) if (std::fabs(A(i,i)) > big) { big = std::fabs(A(i,i)); piv = i; }
    transp[k] = piv;
    if (piv != k) {
      // symmetric swap of rows/cols k and piv, touching the lower triangle only
      for (int j = 0; j < k; ++j) std::swap(A(k,j), A(piv,j));
      for (int i = piv + 1; i < n; ++i) std::swap(A(i,k), A(i,piv));
      std::swap(A(k,k), A(piv,piv));
      for (int i = k + 1; i < piv; ++i) std::swap(A(i,k), A(piv,i));
    }
    int rs = n - k - 1;
    if (k > 0) {
      // temp = A(k,0:k) .* D(0:k);  A(k,k) -= temp . A(k,0:k);  A(k+1:,k) -= A(k+1:,0:k) temp
      std::vector<double> temp(k);
      for (int j = 0; j < k; ++j) temp[j] = A(j,j) * A(k,j);
      double dkk = A(k,k);
      for (int j = 0; j < k; ++j) dkk -= A(k,j) * temp[j];
      A(k,k) = dkk;
      if (rs > 0)
        for (int i = k + 1; i < n; ++i) {
          double s = 0.0;
          for (int j = 0; j < k; ++j) s += A(i,j) * temp[j];
          A(i,k) -= s;
        }
    }
    double pivot = A(k,k);
    if (rs > 0 && std::fabs(pivot) > 0.0)
      for (int i = k + 1; i < n; ++i) A(i,k) /= pivot;
  }
  std::vector<double> x = b;
  for (int k = 0; k < n; ++k) std::swap(x[k], x[transp[k]]);            // P b
  for (int i = 0; i < n; ++i) { double s = x[i]; for (int j = 0; j < i; ++j) s -= A(i,j) * x[j]; x[i] = s; }  // L^-1
  const double tol = 1.0 / std::numeric_limits<double>::max();
  for (int i = 0; i < n; ++i) x[i] = (std::fabs(A(i,i)) > tol) ? x[i] / A(i,i) : 0.0;                       // D^-1
  for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int j = i + 1; j < n; ++j) s -= A(j,i) * x[j]; x[i] = s; }  // L^-T
  for (int k = n - 1; k >= 0; --k) std::swap(x[k], x[transp[k]]);       // P^T
  return x;
}

}  // namespace vxo
