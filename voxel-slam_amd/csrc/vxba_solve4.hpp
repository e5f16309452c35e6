// The damped 6W-dimensional solve of the LiDAR-only LM loop on FOUR waves (Lidar_BA_Optimizer::damping_iter, voxel_map.hpp:397-410:
// (Hess + u D) dxi = -JacT, gauge rows excluded), included by vxba_kernels.hip.  Round 3: replaces the one-wave scalar elimination
// of vxba_solve.hpp (54 dependent publish -> fetch -> reciprocal steps, 16 us at W = 10) on the critical path of every LM step.
//
// Blocked right-looking LDL^T by 6 x 6 blocks (one block = one frame's pose), lane = row, the 6-column block columns dealt to the
// waves round robin (block b belongs to wave b mod 4) and kept in LDS between updates:
//   * the OWNER of block column s publishes its rows, every lane of the owner reads the 6 x 6 diagonal block back (broadcast reads)
//     and factors it for itself in registers (no hand-over inside the six pivots), solves its own row of the panel
//     (g = a L^-T, l = g D^-1) and publishes g and l: ONE s_barrier per block;
//   * behind the barrier every wave takes  sum_p g_p[row] l_p[col]  out of the block columns it owns -- the next owner only out of
//     block s+1 (the urgent one: it then goes straight on to factor it), the others out of all of theirs, in the shadow of the next
//     owner's chain;
//   * the right-hand side is a chain of its own, one step behind: the wave that factored block s still holds L_ss, D_s and g in
//     registers when it leaves the barrier and does the forward substitution of that block (y_s = L_ss^-1 b_s, b_rows -= g z) then,
//     off the critical path;
//   * back substitution x = L^-T z on one wave: the rows of L stream out of LDS ahead of the 54 dependent (readlane, fma) steps.
// What a lane computes for rows at or above the current block is never consumed (upper triangle / finished rows), so nothing is
// masked.  Pivots follow pivot_rcp_f64's rule (vxba_solve.hpp): a frame without observations has an all-zero block, every
// multiplier of that block is 0 and its dxi comes out 0 -- Eigen's LDLT::solve gives the same (voxel_map.hpp:403).
//
// Included by vxba_kernels.hip INSIDE namespace vxk, after the LM state types, dbg_stamp and vxba_solve.hpp.
#pragma once

typedef double s4v2 __attribute__((ext_vector_type(2)));

constexpr int S4_WAVES = 4;
constexpr int S4_THREADS = 64 * S4_WAVES;
constexpr int S4_ROW = 6;                      // doubles per row of a block column (48 B: lane-strided 16-byte accesses are conflict-free)
constexpr int S4_BLK = 64 * S4_ROW + 6;        // block-column stride; the +6 spreads one ROW of L over the banks (back substitution)
constexpr int S4_GSLOT = 64 * S4_ROW;

template <int W>
struct S4 {
  static constexpr int N = 6 * W, M = N - 6, B = W - 1;
  static constexpr int TC = 0;                                   // [B][S4_BLK]  trailing block columns, overwritten by the panels of L
  static constexpr int G = TC + (B > 0 ? B : 1) * S4_BLK;       // [B][64][6]   g rows of every panel (a block column may lag several panels behind)
  static constexpr int BV = G + (B > 0 ? B : 1) * S4_GSLOT;     // [64] right-hand side
  static constexpr int ZV = BV + 64;                             // [64] z = D^-1 L^-1 b
  static constexpr int XS = ZV + 64;                             // [64] dxi (all 6W entries, gauge rows 0)
  static constexpr int DOUBLES = XS + 64;
};

// 1/d to fp64 round-off in three dependent operations after the estimate: e = 1 - d r, r (1 + e + e^2) -- the cubic step reaches
// 2^-69 from the instruction's 2^-23 where two Newton steps need four dependent operations
__device__ __forceinline__ double s4_rcp(double d) {
  const double r = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, r, 1.0);
  const double t = fma(e, e, e);
  const double q = fma(r, t, r);
  return fabs(d) > 1e-300 ? q : 0.0;
}

// D (lower triangle, entry (q, p) at q (q + 1) / 2 + p) -> unit lower L in the strictly lower slots, inv[p] = 1 / d_p
__device__ __forceinline__ void s4_ldl6(double (&D)[21], double (&inv)[6]) {
#pragma unroll
  for (int p = 0; p < 6; p++) {
    inv[p] = s4_rcp(D[p * (p + 1) / 2 + p]);
    double col[6];
#pragma unroll
    for (int q = p + 1; q < 6; q++) col[q] = D[q * (q + 1) / 2 + p];
#pragma unroll
    for (int q = p + 1; q < 6; q++) {
      const double l = col[q] * inv[p];
#pragma unroll
      for (int r = p + 1; r <= q; r++) D[q * (q + 1) / 2 + r] = fma(-l, col[r], D[q * (q + 1) / 2 + r]);
      D[q * (q + 1) / 2 + p] = l;
    }
  }
}

__device__ __forceinline__ void s4_load_row(const double* p, double (&a)[6]) {
  const s4v2* v = reinterpret_cast<const s4v2*>(p);
  const s4v2 t0 = v[0], t1 = v[1], t2 = v[2];
  a[0] = t0[0]; a[1] = t0[1]; a[2] = t1[0]; a[3] = t1[1]; a[4] = t2[0]; a[5] = t2[1];
}
__device__ __forceinline__ void s4_store_row(double* p, const double (&a)[6]) {
  s4v2* v = reinterpret_cast<s4v2*>(p);
  v[0] = (s4v2){a[0], a[1]};
  v[1] = (s4v2){a[2], a[3]};
  v[2] = (s4v2){a[4], a[5]};
}

// T(row, c) -= sum_p g_p(row) l_p(6 b + c): panel t out of block column b (T = the lane's row of it)
template <int W>
__device__ __forceinline__ void s4_apply(const double* lds, int t, int b, int lane, double (&T)[6]) {
  using C = S4<W>;
  double g[6];
  s4_load_row(lds + C::G + t * S4_GSLOT + lane * S4_ROW, g);
  const double* lr = lds + C::TC + t * S4_BLK + (6 * b) * S4_ROW;   // rows 6b .. 6b+5 of panel t: wave-uniform (broadcast) reads
#pragma unroll
  for (int c = 0; c < 6; c++) {
    double l[6];
    s4_load_row(lr + c * S4_ROW, l);
#pragma unroll
    for (int p = 0; p < 6; p++) T[c] = fma(-g[p], l[p], T[c]);
  }
}

// Called by all S4_THREADS threads of one workgroup.  lds: S4<W>::DOUBLES doubles, 16-byte aligned.  Reads the gauge-fixed system the
// Hessian sweep's reduction left in the LM state, writes dxi, the trial poses (agent-scope stores: the residual sweep's workgroups
// read them right after the sequence number) and q1.  Returns with the trial-pose stores of wave 0 issued but not waited for.
//
// LiDAR-inertial shells (li_rec != nullptr): the same solve on the REDUCED pose system of LI_BA_Optimizer's 15W-dimensional step
// (voxel_map.hpp:597).  Velocities and biases only meet the poses through the IMU factors, so the host eliminates them while the
// Hessian sweep is still running (band Cholesky + Schur complement, vxba_host.hpp) and leaves in mapped host memory what that
// elimination adds to the pose block -- li_rec = [u | current poses 12W | e (6W) | E (6W x 6W, column-major)], E = IMU pose terms
// (damped) - W^T W, e = the matching right-hand-side terms -- before this launch is issued; here
//   (H_lidar + u diag(H_lidar) + E) dx = -J_lidar + e
// is solved on the device, the trial poses go to the residual sweep's workgroups as in the LiDAR-only loop, and dx + the trial poses
// + `seq` (last) go to li_out in mapped host memory, from which the host substitutes the velocity / bias unknowns back.  No kernel
// waits for the host and the 6W-dimensional system never crosses PCIe on the critical path.
constexpr int li_rec_len(int W) { return 1 + 12 * W + 6 * W + 36 * W * W; }
constexpr int li_out_len(int W) { return 6 * W + 12 * W + 1; }
// Returns 1 when the loop had already been left (nothing computed, nothing written), else 0.
template <int W, bool DBG>
__device__ __forceinline__ int lm_solve_body4(LMState* st, int c, double* lds, const double* __restrict__ li_rec = nullptr, double* __restrict__ li_out = nullptr,
                                               unsigned li_seq = 0) {
  using C = S4<W>;
  LMCtl& ctl = st->ctl[c];
  // "loop done" is TESTED behind the loads of the system (below): one round of memory latency at the head of every step's critical path
  // instead of two (the flag, then everything else)
  const int loop_done = ctl.done;
  constexpr int n = C::N, M = C::M, B = C::B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  dbg_stamp(DBG && wave == 0, 4000, 0);
  dbg_stamp(DBG && wave == 0, 4000, 1);
  const double u = li_rec ? li_rec[0] : ctl.u;
  const double* __restrict__ li_x = li_rec ? li_rec + 1 : nullptr;
  const double* __restrict__ li_e = li_rec ? li_rec + 1 + 12 * W : nullptr;
  const double* __restrict__ li_E = li_rec ? li_rec + 1 + 18 * W : nullptr;
  const bool row_ok = lane < M;
  const int gi_row = row_ok ? 6 + lane : 0;     // global row of this lane

  // ---- load: wave w brings in the block columns it owns (damping on the diagonal, voxel_map.hpp:402), wave 1 the right-hand side.
  // No barrier behind the loads: the first step needs block column 0 only, which wave 0 loads first and stores at once -- the
  // factorisation of block 0 starts when those six loads have landed; everything else (the other block columns, the right-hand
  // side) is stored before the first step's barrier, which is the first point where another wave's data is read.
  constexpr int NQ = (B + S4_WAVES - 1) / S4_WAVES;
  double a0[NQ > 0 ? NQ : 1][6];
#pragma unroll
  for (int q = 0; q < NQ; q++) {          // all requests first (a block the wave does not have re-reads block `wave`)
    const int b = wave + S4_WAVES * q;
    const int bb = b < B ? b : 0;
#pragma unroll
    for (int cc = 0; cc < 6; cc++) {
      const int col = 6 + 6 * bb + cc;
      const double h = st->Hwork[(size_t)col * n + gi_row];
      const double add = (li_E && gi_row >= 6 + 6 * bb) ? li_E[(size_t)col * n + gi_row] : 0.0;   // rows above a block column are never consumed: half the PCIe reads
      a0[q][cc] = row_ok ? (((col == gi_row) ? h + u * h : h) + add) : 0.0;
    }
  }
  double hii = 0.0, gi = 0.0, rhs0 = 0.0;
  double xcur[12];
  if (wave == 0) {
    hii = st->Hwork[(size_t)gi_row * n + gi_row];
    gi = st->Jwork[gi_row];
    const int fl = lane < W ? lane : 0;   // current pose of frame `lane`, fetched now so that its latency hides behind the factorisation
#pragma unroll
    for (int k = 0; k < 12; k++) xcur[k] = li_x ? li_x[12 * fl + k] : ctl.x[12 * fl + k];
  }
  if (wave == 1 % S4_WAVES) rhs0 = row_ok ? (li_e ? li_e[gi_row] : 0.0) - st->Jwork[gi_row] : 0.0;
  if (loop_done) return 1;
  if (wave < B) s4_store_row(lds + C::TC + wave * S4_BLK + lane * S4_ROW, a0[0]);
  dbg_stamp(DBG && wave == 0, 4000, 2);

  // ---- factorisation
  int ap0 = -1, ap1 = -1, ap2 = -1;   // last panel applied to this wave's block slots 0 .. 2 (block = wave + 4 slot)
  for (int s = 0; s < B; s++) {
    const int owner = s & (S4_WAVES - 1);
    double Ld[21], inv[6], g[6];
    if (wave == owner) {
      const int sq = s >> 2;
      const int ap = sq == 0 ? ap0 : (sq == 1 ? ap1 : ap2);
      double* trow = lds + C::TC + s * S4_BLK + lane * S4_ROW;
      double a[6];
      dbg_stamp(DBG, 4010 + s, 0);
      s4_load_row(trow, a);
      for (int t = ap + 1; t < s; t++) s4_apply<W>(lds, t, s, lane, a);
      if (DBG) { asm volatile("" :: "v"(a[0]), "v"(a[5])); dbg_stamp(true, 4010 + s, 1); }
      // the rows of block s to LDS, the diagonal block back into every lane (one wave: LDS operations execute in order)
      s4_store_row(trow, a);
      __builtin_amdgcn_wave_barrier();
      {
        const double* dr = lds + C::TC + s * S4_BLK + (6 * s) * S4_ROW;
        const s4v2 r0 = *reinterpret_cast<const s4v2*>(dr);
        const s4v2 r1 = *reinterpret_cast<const s4v2*>(dr + 6);
        const s4v2 r2a = *reinterpret_cast<const s4v2*>(dr + 12), r2b = *reinterpret_cast<const s4v2*>(dr + 14);
        const s4v2 r3a = *reinterpret_cast<const s4v2*>(dr + 18), r3b = *reinterpret_cast<const s4v2*>(dr + 20);
        const s4v2 r4a = *reinterpret_cast<const s4v2*>(dr + 24), r4b = *reinterpret_cast<const s4v2*>(dr + 26), r4c = *reinterpret_cast<const s4v2*>(dr + 28);
        const s4v2 r5a = *reinterpret_cast<const s4v2*>(dr + 30), r5b = *reinterpret_cast<const s4v2*>(dr + 32), r5c = *reinterpret_cast<const s4v2*>(dr + 34);
        Ld[0] = r0[0];
        Ld[1] = r1[0]; Ld[2] = r1[1];
        Ld[3] = r2a[0]; Ld[4] = r2a[1]; Ld[5] = r2b[0];
        Ld[6] = r3a[0]; Ld[7] = r3a[1]; Ld[8] = r3b[0]; Ld[9] = r3b[1];
        Ld[10] = r4a[0]; Ld[11] = r4a[1]; Ld[12] = r4b[0]; Ld[13] = r4b[1]; Ld[14] = r4c[0];
        Ld[15] = r5a[0]; Ld[16] = r5a[1]; Ld[17] = r5b[0]; Ld[18] = r5b[1]; Ld[19] = r5c[0]; Ld[20] = r5c[1];
      }
      if (DBG) { asm volatile("" :: "v"(Ld[0]), "v"(Ld[20])); dbg_stamp(true, 4010 + s, 2); }
      s4_ldl6(Ld, inv);
      if (DBG) { asm volatile("" :: "v"(inv[5])); dbg_stamp(true, 4010 + s, 3); }
      // the lane's row of the panel: g = a L^-T (= l D), l = g D^-1
      double l[6];
#pragma unroll
      for (int p = 0; p < 6; p++) {
        double acc = a[p];
#pragma unroll
        for (int q = 0; q < p; q++) acc = fma(-g[q], Ld[p * (p + 1) / 2 + q], acc);
        g[p] = acc;
        l[p] = acc * inv[p];
      }
      s4_store_row(trow, l);
      s4_store_row(lds + C::G + s * S4_GSLOT + lane * S4_ROW, g);
      if (sq == 0) ap0 = s; else if (sq == 1) ap1 = s; else ap2 = s;
      dbg_stamp(DBG, 4010 + s, 4);
    }
    if (s == 0) {
#pragma unroll
      for (int q = 1; q < NQ; q++) {
        const int b = wave + S4_WAVES * q;
        if (b < B) s4_store_row(lds + C::TC + b * S4_BLK + lane * S4_ROW, a0[q]);
      }
      if (wave == 1 % S4_WAVES) lds[C::BV + lane] = rhs0;
    }
    if (s < 9) dbg_stamp(DBG, 4000 + wave, 6 + 2 * s);
    __syncthreads();
    if (s < 9) dbg_stamp(DBG, 4000 + wave, 7 + 2 * s);
    if (wave == owner) {
      // right-hand side, one step behind the factorisation: y_s = L_ss^-1 b_s, z_s = D_s^-1 y_s, rows below lose g z
      double bb[6];
      s4_load_row(lds + C::BV + 6 * s, bb);
      double br = lds[C::BV + lane];
      double z[6];
#pragma unroll
      for (int p = 0; p < 6; p++) {
        double acc = bb[p];
#pragma unroll
        for (int q = 0; q < p; q++) acc = fma(-bb[q], Ld[p * (p + 1) / 2 + q], acc);
        bb[p] = acc;            // y_p
        z[p] = acc * inv[p];
      }
#pragma unroll
      for (int p = 0; p < 6; p++) br = fma(-g[p], z[p], br);
      const int d = lane - 6 * s;
      double zsel = z[0];
#pragma unroll
      for (int p = 1; p < 6; p++) zsel = (d == p) ? z[p] : zsel;
      if (d >= 0 && d < 6) lds[C::ZV + lane] = zsel;
      else lds[C::BV + lane] = br;      // rows of the block keep their slot (read above, never again); rows above are not read either
    }
    const bool next_owner = (s + 1 < B) && wave == ((s + 1) & (S4_WAVES - 1));
    if (!next_owner) {
      // Catch up the block columns this wave owns with the panels published so far -- at most two (block, panel) updates per
      // interval, the block that becomes urgent in the NEXT interval first and completely: that keeps every wave's work between two
      // barriers below the owner's chain (the wave that has just factored a block otherwise arrives with two panels x two blocks of
      // backlog and everybody waits for it), and the next owner always finds exactly one panel left to take out of its block.
      int budget = 2;
#pragma unroll
      for (int q = 0; q < (B + S4_WAVES - 1) / S4_WAVES; q++) {
        const int b = wave + S4_WAVES * q;
        int ap = q == 0 ? ap0 : (q == 1 ? ap1 : ap2);
        const bool must = b == s + 2;
        if (b > s && b < B && ap < s && (must || budget > 0)) {
          double* trow = lds + C::TC + b * S4_BLK + lane * S4_ROW;
          double T[6];
          s4_load_row(trow, T);
          while (ap < s && (must || budget > 0)) {
            s4_apply<W>(lds, ap + 1, b, lane, T);
            ap++;
            budget--;
          }
          s4_store_row(trow, T);
          if (q == 0) ap0 = ap; else if (q == 1) ap1 = ap; else ap2 = ap;
        }
      }
    }
  }
  __syncthreads();
  dbg_stamp(DBG && wave == 0, 4000, 3);
  if (wave != 0) return 0;

  // ---- back substitution on wave 0: x = L^-T z.  Lane j (column j) takes L(r, j) x_r out of z_j for r = M-1 .. j+1; the rows of L are
  // requested a block ahead of the dependent (readlane, fma) chain.
  double x = row_ok ? lds[C::ZV + lane] : 0.0;
  if constexpr (B > 0) {
    const int jb = (row_ok ? lane : 0) / 6, jc = (row_ok ? lane : 0) % 6;
    const double* lcol = lds + C::TC + jb * S4_BLK + jc;    // L(r, lane) at lcol[6 r]
    double Lr[6], Ln[6];
#pragma unroll
    for (int k = 0; k < 6; k++) Lr[k] = lcol[(M - 1 - k) * S4_ROW];
    for (int rb = B - 1; rb >= 0; rb--) {
      if (rb > 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) Ln[k] = lcol[(6 * rb - 1 - k) * S4_ROW];
      }
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int r = 6 * rb + 5 - k;
        const double xr = readlane_f64(x, r);
        x = (lane < r) ? fma(-Lr[k], xr, x) : x;
      }
#pragma unroll
      for (int k = 0; k < 6; k++) Lr[k] = Ln[k];
    }
  }
  dbg_stamp(DBG, 4000, 4);
  // dxi, trial state (voxel_map.hpp:405-409), q1 = 0.5 dxi . (u D dxi - JacT) (:410)
  if (row_ok) st->dxi[6 + lane] = x;
  if (lane < 6) st->dxi[lane] = 0.0;
  double* xs = lds + C::XS;
  if (lane < 6) xs[lane] = 0.0;
  if (row_ok) xs[6 + lane] = x;
  __builtin_amdgcn_wave_barrier();
  if (lane < W) {
    double dl[6];
#pragma unroll
    for (int k = 0; k < 6; k++) dl[k] = xs[6 * lane + k];
    double xn[9];
    lm_right_multiply_exp(xcur, dl, xn);
#pragma unroll
    for (int k = 0; k < 9; k++) __hip_atomic_store(&ctl.xt[12 * lane + k], xn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < 3; k++) __hip_atomic_store(&ctl.xt[12 * lane + 9 + k], xcur[9 + k] + dl[3 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (li_out) {   // the host takes the trial poses from here, so that both sides evaluate at the same bits
#pragma unroll
      for (int k = 0; k < 9; k++) li_out[6 * W + 12 * lane + k] = xn[k];
#pragma unroll
      for (int k = 0; k < 3; k++) li_out[6 * W + 12 * lane + 9 + k] = xcur[9 + k] + dl[3 + k];
    }
  }
  if (li_out) {
    // the host's half of the step (velocities, biases, q1) needs dx; the sequence number goes out when everything before it has landed
    if (row_ok) li_out[6 + lane] = x;
    if (lane < 6) li_out[lane] = 0.0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) li_out[18 * W] = (double)li_seq;
  }
  double part = row_ok ? x * (u * hii * x - gi) : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
  if (lane == 0) ctl.q1 = 0.5 * part;
  if (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_stamp(true, 4000, 5); }
  return 0;
}
