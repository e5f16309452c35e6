// EXPERIMENT (round 2, measured and rejected -- not part of the library): the damped 6W-dimensional solve by 6x6 blocks instead of
// the scalar elimination with look-ahead of csrc/vxba_solve.hpp (dense_solve_rows).  To try it again: paste into vxba_solve.hpp
// (namespace vxk, after dense_solve_rows) and call dense_solve_rows_blocked<n>(A, b, lds, lane) from lm_solve_body with BSOLVE_LDS
// doubles of LDS.
//
// Same-box A/B at cfg2 (W = 10, 54 unknowns), launch = solve + residual sweep, scalar version 28.5-29.5 us:
//   v1  factor at the head of every step, trailing update straight from LDS                 44.1 us  (parity suite green)
//   v2  + register double buffer for the trailing reads (next four columns requested first)  33.2 us  (parity suite green)
//   v3  + next block read back and factored inside the first trailing groups (this file)     31.4 us  (one parity case fails: bug not found)
// Two compiler traps on the way, both worth remembering: (1) nothing needs A[j] before the next step, so LLVM keeps the broadcast READS
// of the trailing update where they are and sinks the multiply-adds to their first use -- 576 read results live at K = 1, ~1000
// registers spilled; an empty asm volatile("" : "+v"(A[j])) after each column pins the arithmetic.  (2) volatile LDS pointers turn the
// reads into flat loads with a full wait each.  Why it loses: the instruction counts are equal (6.3k vs 6.2k), but the scalar version's
// look-ahead hides every publish -> fetch -> reciprocal chain behind the previous rank-1 update, while here each step still exposes
// panel publish -> 27 broadcast reads -> 6 dependent reciprocals -> two triangular solves before its (well pipelined) update starts.
#pragma once
namespace vxk {
// ------------------------------------------------------------------------------------------------------------------------------
// The same solve by 6x6 BLOCKS (one block = one frame's pose): N/6 - 1 dependent steps instead of N - 6.
// Step K: every lane publishes its six entries of block column K (and its right-hand side) to LDS; every lane reads the 6x6 diagonal
// block D_K back (broadcast reads) and factors it for itself, D_K = L d L^T, in registers (21 values: redundant work costs nothing on
// a one-wave problem, a hand-over would cost an LDS round trip per pivot); a row below the block forms its multipliers
// M = a D_K^-1 from its own six panel entries and the factors (two triangular solves, 36 operations) and takes
// sum_r M_r * (row 6K+r) out of its trailing columns -- by symmetry "row 6K+r at column j" is row j's published panel entry, so the
// update reads the panel, not the pivot rows, and the pivot rows themselves need no elimination inside the block: they simply stop
// being updated (block Gaussian elimination), and back substitution solves D_K x_K = b_K - sum_{j > K} A_Kj x_j with the kept factors.
// Pivots follow pivot_rcp_f64's rule (a frame without observations: all-zero block, x = 0).
// LDS: BSOLVE_LDS doubles.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int BSOLVE_PANEL = 7 * 64;                       // six panel columns + the right-hand side, one value per lane each
constexpr int BSOLVE_FAC = 27;                             // L (15, strictly lower, row-major) | 1/d (6) | x of the block (6)
constexpr int BSOLVE_LDS = 2 * BSOLVE_PANEL + 64 + 16 * BSOLVE_FAC;

// D (lower triangle, d[q][p] at q*(q+1)/2 + p) -> L (strictly lower, same indexing; diagonal slots hold d_p) and inv[p] = 1/d_p
__device__ __forceinline__ void ldl6(double (&D)[21], double (&inv)[6]) {
#pragma unroll
  for (int p = 0; p < 6; p++) {
    inv[p] = pivot_rcp_f64(D[p * (p + 1) / 2 + p]);
    double col[6];                       // column p below the pivot, unscaled (= d_p l_rp)
#pragma unroll
    for (int q = p + 1; q < 6; q++) col[q] = D[q * (q + 1) / 2 + p];
#pragma unroll
    for (int q = p + 1; q < 6; q++) {
      const double l = col[q] * inv[p];
#pragma unroll
      for (int r = p + 1; r <= q; r++) D[q * (q + 1) / 2 + r] -= l * col[r];
      D[q * (q + 1) / 2 + p] = l;
    }
  }
}
// y (row vector) <- y D^-1 = ((y L^-T) diag(1/d)) L^-1 with the factors of ldl6
__device__ __forceinline__ void ldl6_solve(const double (&L)[21], const double (&inv)[6], double (&y)[6]) {
#pragma unroll
  for (int p = 0; p < 6; p++)            // z = y L^-T  (forward: z_p = y_p - sum_{r<p} z_r L_pr)
#pragma unroll
    for (int r = 0; r < p; r++) y[p] -= y[r] * L[p * (p + 1) / 2 + r];
#pragma unroll
  for (int p = 0; p < 6; p++) y[p] *= inv[p];
#pragma unroll
  for (int p = 5; p >= 0; p--)           // m = w L^-1  (backward: m_p = w_p - sum_{q>p} m_q L_qp)
#pragma unroll
    for (int q = p + 1; q < 6; q++) y[p] -= y[q] * L[q * (q + 1) / 2 + p];
}

template <int K, int N>
struct BlockElim {
  static constexpr int NB = N / 6;
  // lds: [panel buffers 2 x (6 x 64 + 64)] [unused 64] [factors NB x BSOLVE_FAC]
  static __device__ __forceinline__ double* panel(double* lds, int k) { return lds + (k & 1) * BSOLVE_PANEL; }
  static __device__ __forceinline__ double* fac(double* lds, int k) { return lds + 2 * BSOLVE_PANEL + 64 + k * BSOLVE_FAC; }

  // D (factored in place by ldl6), inv, bk: block K's diagonal block and right-hand side, read and factored by the PREVIOUS step (in the
  // shadow of its trailing update) or by the caller for the first block.
  static __device__ __forceinline__ void read_block(const double* P, int k, double (&D)[21], double (&bk)[6]) {
#pragma unroll
    for (int q = 0; q < 6; q++)
#pragma unroll
      for (int p = 0; p <= q; p++) D[q * (q + 1) / 2 + p] = P[p * 64 + 6 * k + q];     // row 6k+q, column 6k+p
#pragma unroll
    for (int r = 0; r < 6; r++) bk[r] = P[6 * 64 + 6 * k + r];
  }
  static __device__ __forceinline__ void forward(double (&A)[N > 6 ? N : 7], double& b, double* lds, int lane, double (&D)[21], double (&inv)[6], double (&bk)[6]) {
    if constexpr (K < NB) {
      double* P = panel(lds, K);
      if (lane == 0) {
        double* F = fac(lds, K);
#pragma unroll
        for (int q = 1; q < 6; q++)
#pragma unroll
          for (int p = 0; p < q; p++) F[q * (q - 1) / 2 + p] = D[q * (q + 1) / 2 + p];
#pragma unroll
        for (int p = 0; p < 6; p++) F[15 + p] = inv[p];
      }
      constexpr int J0 = 6 * (K + 1);     // first trailing column
      double Dn[21], invn[6], bn[6];
#pragma unroll
      for (int k = 0; k < 21; k++) Dn[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 6; k++) { invn[k] = 0.0; bn[k] = 0.0; }
      if constexpr (J0 < N) {
        const bool below = lane >= J0 && lane < N;
        double M[6];
#pragma unroll
        for (int r = 0; r < 6; r++) M[r] = below ? A[6 * K + r] : 0.0;
        ldl6_solve(D, inv, M);
        // the NEXT panel first: update its six columns, publish them with the updated right-hand side ...
        double* Pn = panel(lds, K + 1);
#pragma unroll
        for (int r = 0; r < 6; r++) b -= M[r] * bk[r];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = J0; j < J0 + 6; j++) {
#pragma unroll
          for (int r = 0; r < 6; r++) A[j] -= M[r] * P[r * 64 + j];
          Pn[(j - J0) * 64 + lane] = A[j];
        }
        Pn[6 * 64 + lane] = b;
        __builtin_amdgcn_wave_barrier();
        // ... read the next diagonal block back and factor it NOW: its dependent chain (six reciprocals) runs interleaved with the
        // first groups of the trailing update below instead of heading the next step
        read_block(Pn, K + 1, Dn, bn);
        constexpr int G = 4, JT = J0 + 6, NG = (N - JT + G - 1) / G;
        constexpr int MIX = 3;            // trailing groups that share a scheduling region with the factorisation
        // Four columns (24 broadcast reads) at a time, the reads of the next four requested before the multiply-adds of the current
        // ones (register double buffer), and every column's result pinned where it is computed: nothing needs A[j] before the next
        // step, so the compiler otherwise keeps the READS in place and sinks the multiply-adds to their first use -- 576 read results
        // waiting in registers at K = 1, a thousand of them spilled.
        double buf[2][6 * G];
        if constexpr (NG > 0) {
#pragma unroll
          for (int c = 0; c < G; c++)
#pragma unroll
            for (int r = 0; r < 6; r++) buf[0][6 * c + r] = (JT + c < N) ? P[r * 64 + JT + c] : 0.0;
        }
        ldl6(Dn, invn);
#pragma unroll
        for (int g = 0; g < NG; g++) {
          if (g + 1 < NG) {
#pragma unroll
            for (int c = 0; c < G; c++)
#pragma unroll
              for (int r = 0; r < 6; r++) buf[(g + 1) & 1][6 * c + r] = (JT + G * (g + 1) + c < N) ? P[r * 64 + JT + G * (g + 1) + c] : 0.0;
          }
          if (g >= MIX) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < G; c++) {
            const int j = JT + G * g + c;
            if (j < N) {
#pragma unroll
              for (int r = 0; r < 6; r++) A[j] -= M[r] * buf[g & 1][6 * c + r];
              asm volatile("" : "+v"(A[j]));
            }
          }
          if (g >= MIX - 1) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NG == 0) ldl6(Dn, invn);
      }
      __builtin_amdgcn_sched_barrier(0);
      BlockElim<K + 1, N>::forward(A, b, lds, lane, Dn, invn, bn);
    }
  }
  // x of blocks > K is known and already taken out of b; solves block K and takes it out of the rows above
  static __device__ __forceinline__ void backward(double (&A)[N > 6 ? N : 7], double& b, double& x, double* lds, int lane) {
    if constexpr (K >= 1) {
      double* sv = lds + 2 * BSOLVE_PANEL;
      sv[lane] = b;
      __builtin_amdgcn_wave_barrier();
      const double* F = fac(lds, K);
      double L[21], inv[6], y[6];
#pragma unroll
      for (int q = 1; q < 6; q++)
#pragma unroll
        for (int p = 0; p < q; p++) L[q * (q + 1) / 2 + p] = F[q * (q - 1) / 2 + p];
#pragma unroll
      for (int p = 0; p < 6; p++) { inv[p] = F[15 + p]; L[p * (p + 1) / 2 + p] = 0.0; y[p] = sv[6 * K + p]; }
      ldl6_solve(L, inv, y);            // D symmetric: D^-1 s as a row vector
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 6; r++) {
        x = (lane == 6 * K + r) ? y[r] : x;
        b -= (lane < 6 * K) ? A[6 * K + r] * y[r] : 0.0;
      }
      __builtin_amdgcn_sched_barrier(0);
      BlockElim<K - 1, N>::backward(A, b, x, lds, lane);
    }
  }
};

// Same contract as dense_solve_rows (rows 6 .. N-1, row i in lane i); lds: BSOLVE_LDS doubles.  N a multiple of 6.
template <int N>
__device__ __forceinline__ double dense_solve_rows_blocked(double (&A)[N > 6 ? N : 7], double b, double* lds, int lane) {
  double x = 0.0;
  if constexpr (N > 6) {
    double* P = BlockElim<1, N>::panel(lds, 1);
#pragma unroll
    for (int r = 0; r < 6; r++) P[r * 64 + lane] = A[6 + r];
    P[6 * 64 + lane] = b;
    __builtin_amdgcn_wave_barrier();
    double D[21], inv[6], bk[6];
    BlockElim<1, N>::read_block(P, 1, D, bk);
    ldl6(D, inv);
    BlockElim<1, N>::forward(A, b, lds, lane, D, inv, bk);
    BlockElim<N / 6 - 1, N>::backward(A, b, x, lds, lane);
  }
  return (lane >= 6 && lane < N) ? x : 0.0;
}

}  // namespace vxk
