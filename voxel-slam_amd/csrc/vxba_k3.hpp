// K3 -- Hessian / gradient sweep (LidarFactor::acc_evaluate2, voxel_map.hpp:132-241).  Included by vxba_kernels.hip inside
// namespace vxk, after the LM helpers (lm_decide / lm_residual2 / lm_carry / lm_persist) and dbg_stamp.
//
// Per voxel  H_a = -B_a^T B_a + blockdiag_i(D_{a,i})  with B_a the 3 x 6W matrix of SURVEY A.4, so the window Hessian is a
// tall-skinny SYRK over the 3V stacked rows  S = sum B^T B;  gradient, block-diagonal terms and the residual are 28 linear
// accumulators per frame (vxm::k3_entry_emit).
//
// Round 5: the SYRK runs on v_mfma_f64_4x4x4_4b_f64 with the instruction's four BLOCKS used as four K-SLICES of one 4 x 4 output
// block.  The instruction multiplies block t of A (4 x 4) with block t of B (4 x 4), t = 0..3, independently; operand lane
// l = 16 k + 4 t + i holds A_t[i][k] (B alike, with j for i), result lane 16 i + 4 t + j holds D_t[i][j] (probed:
// scripts/ubench/k3_blockk_probe.hip).  With the operand register of COLUMN GROUP g (columns 4g .. 4g+3 of B) loaded as
//     X_g[lane] = T[16 q + (lane >> 2)][4 g + (lane & 3)]         (a 16-row slab q of the step's row tile T in LDS)
// -- (k, t) -> row 4 k + t is a bijection onto the slab's 16 rows -- one  D = mfma(X_I, X_J, D)  adds, in block t, the products of
// the four rows {t, 4 + t, 8 + t, 12 + t} to the 4 x 4 output block (I, J); the sum over the four blocks is taken ONCE, at the end
// of the kernel.  What that buys against the 16x16x4 form of rounds 1-4 (ten 16 x 16 tile pairs on 64 padded columns):
//   * block-exact upper triangle: 120 pairs of 4-column groups at W = 10 (1920 entries for 1830 distinct ones) instead of 2560
//     -- 135 instructions of 17.5 cycles per wave and step instead of 45 of 64 (measured: 4717 against 5866 cycles per step);
//   * one operand register per column group serves as A and as B of every pair it takes part in: a wave reads 5 - 13 operand
//     registers per slab for 15 instructions (fewer LDS operand reads than before, not more);
//   * ONE f64 accumulator (two registers) per pair: a wave OWNS 15 of the 120 pairs over ALL rows of every step -- 30 accumulator
//     registers instead of 40, and no cross-wave reduction at the end: the epilogue of rounds 1-4 parked 80 KB of accumulators in
//     LDS and summed four K ranges; now every wave folds its own four blocks (two lane exchanges per pair) and stores its 240 sums.
// The block-diagonal terms Drt, Dtt can no longer ride through spare tile columns (there are none: 6W = 60 columns are 15 groups
// exactly); they are linear accumulators again (22 fp64 operations per entry -- 176 cycles per SIMD and step against the 574 a
// sixteenth column group would cost).
//
// Unchanged from round 2: ONE 8-wave workgroup per CU, two waves per SIMD (f64 MFMA and f64 VALU share one datapath: what the
// second wave buys is that everything that is NOT fp64 runs under the other wave's fp64 work);
//   phase A  every wave turns its own batch (NV voxels x W frames, one lane per entry) into 3 NV rows of B in a workgroup-shared,
//            double-buffered LDS tile of 8 x 3 NV rows (144 rows at W = 10 = nine 16-row slabs);
//   phase M  wave w multiplies ITS pairs over all slabs of the previous step's tile;
// skewed loop -- phase M of step s-1, then phase A of step s, then ONE barrier; the next batch's loads ride behind the first two
// slabs of phase M; workgroups take contiguous, evenly sized runs of batches; the run's last, partly filled step only covers the
// slabs that exist.  No float atomics: bitwise reproducible for a given launch geometry.
//
// Mixed precision (BASELINE configs[2], "fp32 Jacobian with fp64 Hessian accumulation"): the rows of B are ROUNDED TO f32 on their
// way into the tile, products and sums stay f64 (the f32 matrix instruction of rounds 2-4 needed 60 accumulator registers per
// wave in this mapping; the tolerance study is what the configuration is for, and it is the same now -- tests/test_gpu_parity.py).
#pragma once
#include <type_traits>

// ---- which (I, J) column-group pairs a wave owns -----------------------------------------------------------------------------
// NG groups, NP = NG (NG + 1) / 2 pairs I <= J, cut into 8 runs of PPW = ceil(NP / 8) along an order that keeps a run on few groups
// (= few operand reads per slab): groups in sets of five, intra-set pairs first, then the cross blocks row by row.  NG = 15
// (W = 10) uses a hand-made order: three intra-set runs (5 operand reads for 15 instructions) and five runs of three rows of a
// cross block (8 reads; one run straddles two blocks: 12).
struct K3PairTab {
  signed char I[8][16], J[8][16];      // pair j of wave w
  signed char sa[8][16], sb[8][16];    // ... as operand slots of that wave
  signed char slot_group[8][16];       // column group an operand slot reads
  signed char npair[8], nslot[8];
};
__host__ __device__ constexpr K3PairTab k3_make_pairs(int NG) {
  K3PairTab t{};
  int LI[136] = {}, LJ[136] = {};
  int n = 0;
  auto intra = [&](int s0, int s1) { for (int i = s0; i < s1; i++) for (int j = i; j < s1; j++) { LI[n] = i; LJ[n] = j; n++; } };
  auto rows = [&](int r0, int r1, int c0, int c1) { for (int i = r0; i < r1; i++) for (int j = c0; j < c1; j++) { LI[n] = i; LJ[n] = j; n++; } };
  if (NG == 15) {
    intra(0, 5); intra(5, 10); intra(10, 15);
    rows(0, 3, 5, 10);
    rows(3, 5, 5, 10); rows(3, 4, 10, 15);
    rows(0, 3, 10, 15);
    rows(4, 5, 10, 15); rows(5, 7, 10, 15);
    rows(7, 10, 10, 15);
  } else {
    for (int a = 0; a < NG; a += 5) intra(a, a + 5 < NG ? a + 5 : NG);
    for (int a = 0; a < NG; a += 5)
      for (int b = a + 5; b < NG; b += 5) rows(a, a + 5, b, b + 5 < NG ? b + 5 : NG);
  }
  const int NP = NG * (NG + 1) / 2, PPW = (NP + 7) / 8;
  for (int w = 0; w < 8; w++) {
    int ns = 0, np = 0;
    for (int p = w * PPW; p < (w + 1) * PPW && p < NP; p++, np++) {
      t.I[w][np] = (signed char)LI[p];
      t.J[w][np] = (signed char)LJ[p];
      int a = -1, b = -1;
      for (int s = 0; s < ns; s++) { if (t.slot_group[w][s] == LI[p]) a = s; if (t.slot_group[w][s] == LJ[p]) b = s; }
      if (a < 0) { a = ns; t.slot_group[w][ns++] = (signed char)LI[p]; }
      if (LJ[p] == LI[p]) b = a;
      if (b < 0) { b = ns; t.slot_group[w][ns++] = (signed char)LJ[p]; }
      t.sa[w][np] = (signed char)a;
      t.sb[w][np] = (signed char)b;
    }
    t.npair[w] = (signed char)np;
    t.nslot[w] = (signed char)ns;
  }
  return t;
}

template <int W>
struct K3Cfg {
  static constexpr int NG = k3_groups(W);               // 4-wide column groups (6W rounded up: odd W carry two zero columns)
  static constexpr int NP = NG * (NG + 1) / 2;          // group pairs I <= J
  static constexpr int NCOLS = 4 * NG;
  // Row stride of the tile in doubles, == 4 (mod 8): a half-wave of an operand read (ds_read_b64 is served 32 lanes at a time over
  // 64 banks of 4 B) covers 8 consecutive rows x 32 B, and 8 RS mod 256 is then an odd multiple of 32 -- eight distinct 32-byte
  // slots: conflict-free on a plain row-major tile (60 columns at W = 10: no padding at all).
  static constexpr int RS = k3_row_stride(W);
  static constexpr int NV = k3_nv(W);                   // voxels per wave-batch (even; sized so that two tile buffers fit the LDS)
  static constexpr int NACT = NV * W;                   // active lanes
  static constexpr int R = 3 * NV;                      // rows of B per batch
  static constexpr int WAVES = K3_BLOCK / 64;           // 8
  static constexpr int ROWS = WAVES * R;                // rows per step: 288 / 240 / 192 / 144, always a multiple of 16
  static constexpr int KC = ROWS / 16;                  // 16-row slabs per full step
  static constexpr int BUF = ROWS * RS;                 // doubles per tile buffer
  static constexpr int PPW = k3_pairs_per_wave(W);      // pairs (accumulators) per wave: 15 at W = 10
  static constexpr int PPWP = (PPW + 3) & ~3;           // ... rounded up to the store pattern (four pairs per 512-byte store)
  static constexpr int NTILE = WAVES * PPWP * 16;       // doubles of a workgroup partial that hold S: [wave][pair][4 i + j]
  static constexpr int PLEN = NTILE + W * DACC;
  static_assert(WAVES == 8 && ROWS % 16 == 0 && NACT <= 64 && NG <= 15 && PPW <= 16, "step geometry");
  static_assert(PLEN == (int)k3_partial_len(W), "partial layout");
  __host__ __device__ static constexpr K3PairTab pairs() { return k3_make_pairs(NG); }
  // pairs wave w owns (the last runs may be short or empty at small W)
  __host__ __device__ static constexpr int npair(int w) { const int left = NP - w * PPW; return left < 0 ? 0 : (left < PPW ? left : PPW); }
};

// Register image of one (voxel, frame) entry plus the voxel's cached plane parameters.
struct K3Entry {
  double c[10];      // body-frame cluster
  double u[9];       // eigenvectors, plane 3*col+row
  double s1, s2;     // gap scales
  double invN, sc;   // 1 / merged count, sqrt(coe)
  double mv[3];      // merged first moment
  double coe, lam0;
  bool ok;           // lane holds a real (voxel, frame) entry of [head,end)
};

typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Entry loads go through buffer descriptors: the address of every load is  descriptor base (SGPRs) + one 32-bit lane offset
// (VGPR)  -- no 64-bit address arithmetic on the VALU and no address register pairs to carry.
//
// The 18 plane parameters of a voxel (9 eigenvector components, s1, s2, merged first moment, 1/N, sqrt(coe), lambda_0, coe) are
// the same for the W lanes of the voxel.  Round 1 let every lane fetch them itself: 18 load instructions per wave and batch
// whose 64 lanes read 6 distinct addresses.  The CU's one texture-addresser serialises vector-memory instructions at ~16 cycles
// apiece whatever their width, and with eight waves per CU those 144 instructions per step became the longest part of
// phase A (s_memtime stamps: phase A 4.9k cycles with the loads, 2.9k without).  Now the wave fetches the NV x 17 values of the
// cache planes TRANSPOSED -- lane t takes (voxel t / 17, plane t % 17): two instructions at W = 10 -- plus one for coe, parks
// them in registers during phase M (10 registers instead of 36), and redistributes them through a wave-private corner of LDS
// at the start of phase A: 8 vector-memory instructions per wave and batch instead of 23.
struct K3Planes {
  const double *cache_ptr, *coe_ptr;   // cache = eigval(3) | eigvec(9) | merged(10) | aux(4), consecutive planes (FactorView / snapshot)
  const double* clb;
  unsigned vs8;   // plane stride in bytes
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t k3_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)0xffffffff, 0x00020000);   // raw buffer, no range limit
}
// The same descriptor with the range check as an on / off switch: with num_records = 0 every lane of a load is out of range, so
// the load returns zeros WITHOUT a memory request.  A wave-uniform "is there a next batch" becomes one s_cselect on a descriptor
// dword instead of a branch around the loads -- see the note on phi copies at phase_a in k3_hessian_kernel.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t k3_rsrc_gated(const void* p, bool on) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, __builtin_amdgcn_readfirstlane(on ? (int)0xffffffff : 0), 0x00020000);   // wave-uniform by construction; say so
}
__device__ __forceinline__ K3Planes k3_planes(const FactorView& fv) {
  K3Planes pl;
  pl.cache_ptr = fv.eigval;
  pl.coe_ptr = fv.coe;
  pl.clb = fv.clb;
  pl.vs8 = (unsigned)fv.VS * 8u;
  return pl;
}
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
// AUX: cache policy bits of the buffer instruction (0 = default; 16 = sc1: served by the XCD's L2, not by this CU's vector L1 -- what the
// fused residual + Hessian launch needs for cache planes its own workgroup has just rewritten, vxba_k23.hpp)
template <int AUX = 0>
__device__ __forceinline__ double k3_ld64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, AUX));
}

// clusters of batch b: five contiguous 1 KB rows per wave (batch-major copy) -- 83 % of an entry's bytes
__device__ __forceinline__ void k3_load_clusters(const K3Planes& pl, int b, int lane, double c[10], bool on = true) {
  const __amdgpu_buffer_rsrc_t rc = k3_rsrc_gated(pl.clb + (size_t)b * 640, on);
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const v2d t = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rc, lane * 16, j * 1024, 0));
    c[2 * j] = t[0];
    c[2 * j + 1] = t[1];
  }
}

// one of the five cluster rows of batch b (the interleaved form of k3_load_clusters: request j behind K-step j of phase M)
__device__ __forceinline__ void k3_load_cluster_row(const K3Planes& pl, int b, int lane, double c[10], bool on, int j) {
  const __amdgpu_buffer_rsrc_t rc = k3_rsrc_gated(pl.clb + (size_t)b * 640, on);
  const v2d t = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rc, lane * 16, j * 1024, 0));
  c[2 * j] = t[0];
  c[2 * j + 1] = t[1];
}

// Order of a voxel's 18 parameters in the staging record, and the cache plane (relative to eigval) each one comes from.
//   0..8 eigvec 0..8 | 9, 10 s1, s2 (aux 0, 1) | 11..13 merged first moment (merged 6..8) | 14 1/N (aux 2) | 15 sqrt(coe) (aux 3) |
//   16 lambda_0 (eigval 0) | 17 coe (its own plane)
__host__ __device__ constexpr int k3_param_plane(int k) {
  return k < 9 ? 3 + k : (k < 11 ? 22 + (k - 9) : (k < 14 ? 12 + 6 + (k - 11) : (k == 14 ? 24 : (k == 15 ? 25 : 0))));
}
template <int W>
struct K3Stage {
  static constexpr int NV = K3Cfg<W>::NV;
  static constexpr int NITEM = NV * 17;            // transposed items of the cache planes
  static constexpr int Q = (NITEM + 63) / 64;      // load instructions (2 at W = 10, 4 at NV = 12)
  static constexpr int REC = 18;                   // doubles per voxel record in LDS
  static constexpr int WAVE_DOUBLES = NV * REC;    // per-wave staging area
  double v[Q];      // item lane + 64 q
  double coe;       // lane < NV: coe of voxel slot `lane`
};
// request the plane parameters of batch b (transposed); voxels outside [head, end) read voxel `head` instead (masked later)
template <int W, int AUX = 0>
__device__ __forceinline__ void k3_load_params(const K3Planes& pl, int head, int end, int b, int lane, K3Stage<W>& st, bool on = true) {
  using S = K3Stage<W>;
  const __amdgpu_buffer_rsrc_t rcache = k3_rsrc_gated(pl.cache_ptr, on), rcoe = k3_rsrc_gated(pl.coe_ptr, on);
#pragma unroll
  for (int q = 0; q < S::Q; q++) {
    const int t = lane + 64 * q;
    const int tv = t < S::NITEM ? t : 0;             // idle lanes of the last instruction re-read item 0
    const int v = tv / 17, k = tv - 17 * v;
    int a = b * S::NV + v;
    a = (a >= head && a < end) ? a : head;
    // plane index of parameter k: a short select chain on a lane constant (hoisted out of the step loop by the compiler)
    int plane = 0;
#pragma unroll
    for (int kk = 0; kk < 17; kk++) plane = (k == kk) ? k3_param_plane(kk) : plane;
    st.v[q] = k3_ld64<AUX>(rcache, (unsigned)plane * pl.vs8 + (unsigned)a * 8u, 0);
  }
  {
    int a = b * S::NV + (lane < S::NV ? lane : 0);
    a = (a >= head && a < end) ? a : head;
    st.coe = k3_ld64<AUX>(rcoe, (unsigned)a * 8u, 0);
  }
}
// request q of the transposed plane parameters (q < Q) or the coe plane (q == Q): the interleaved form of k3_load_params
template <int W, int AUX = 0>
__device__ __forceinline__ void k3_load_param_q(const K3Planes& pl, int head, int end, int b, int lane, K3Stage<W>& st, bool on, int q) {
  using S = K3Stage<W>;
  if (q < S::Q) {
    const __amdgpu_buffer_rsrc_t rcache = k3_rsrc_gated(pl.cache_ptr, on);
    const int t = lane + 64 * q;
    const int tv = t < S::NITEM ? t : 0;
    const int v = tv / 17, k = tv - 17 * v;
    int a = b * S::NV + v;
    a = (a >= head && a < end) ? a : head;
    int plane = 0;
#pragma unroll
    for (int kk = 0; kk < 17; kk++) plane = (k == kk) ? k3_param_plane(kk) : plane;
    st.v[q] = k3_ld64<AUX>(rcache, (unsigned)plane * pl.vs8 + (unsigned)a * 8u, 0);
  } else {
    const __amdgpu_buffer_rsrc_t rcoe = k3_rsrc_gated(pl.coe_ptr, on);
    int a = b * S::NV + (lane < S::NV ? lane : 0);
    a = (a >= head && a < end) ? a : head;
    st.coe = k3_ld64<AUX>(rcoe, (unsigned)a * 8u, 0);
  }
}
// park the staged values in the wave's LDS corner and read back the lane's own voxel record
// write = false (wave-uniform): the records are in the corner already -- the residual half of a fused launch put them there (vxba_k23.hpp)
template <int W>
__device__ __forceinline__ void k3_unstage_params(const K3Stage<W>& st, double* stage_lds, int head, int end, int b, bool active, int vl, int lane,
                                                  K3Entry& e, bool write = true) {
  using S = K3Stage<W>;
  using C = K3Cfg<W>;
#pragma unroll
  for (int q = 0; q < S::Q; q++) {
    const int t = lane + 64 * q;
    if (t < S::NITEM && write) {
      const int v = t / 17, k = t - 17 * v;
      stage_lds[v * S::REC + k] = st.v[q];
    }
  }
  if (lane < S::NV && write) stage_lds[lane * S::REC + 17] = st.coe;
  __builtin_amdgcn_wave_barrier();   // same wave, LDS operations execute in order: a scheduling fence is all that is needed
  const int a = b * C::NV + vl;
  e.ok = active && a >= head && a < end;
  const v2d* rec = reinterpret_cast<const v2d*>(stage_lds + vl * S::REC);
  double r[18];
#pragma unroll
  for (int j = 0; j < 9; j++) { const v2d t = rec[j]; r[2 * j] = t[0]; r[2 * j + 1] = t[1]; }
#pragma unroll
  for (int k = 0; k < 9; k++) e.u[k] = r[k];
  e.s1 = r[9]; e.s2 = r[10];
  e.mv[0] = r[11]; e.mv[1] = r[12]; e.mv[2] = r[13];
  e.invN = r[14]; e.sc = r[15]; e.lam0 = r[16]; e.coe = r[17];
}

// Phase A of one entry: rows of B_a (3 x 6) and the per-frame linear accumulators, branch-free.
template <bool RT, class Emit>
__device__ __forceinline__ void k3_phase_a(K3Entry& e, int fi, const double* __restrict__ pose, double dacc[DACC], Emit&& emit) {
  // pose of the lane's frame from LDS (C-ABI layout: R column-major | p), transposed on the way in: 24 registers less to
  // carry through phase M
  double R[9], p[3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int cc = 0; cc < 3; cc++) R[3 * r + cc] = pose[3 * cc + r];
#pragma unroll
  for (int k = 0; k < 3; k++) p[k] = pose[9 + k];
  // N == 0: frame did not observe the voxel (voxel_map.hpp:178).  Masking the two scale factors is enough: every row entry and
  // every accumulator increment carries sqrt(coe) or coe as a factor, and the cluster a masked lane holds is finite (a real
  // cluster of a voxel outside [head, end), or the zeros of an unobserved frame), so the products are exact zeros -- 4 selects
  // instead of 24 on a VALU that is the bottleneck of this phase.
  const bool obs = e.ok && e.c[9] != 0.0;
  vxm::VoxelCache vc;
#pragma unroll
  for (int k = 0; k < 3; k++) { vc.u0[k] = e.u[k]; vc.u1[k] = e.u[3 + k]; vc.u2[k] = e.u[6 + k]; }
  vc.s1 = e.s1;
  vc.s2 = e.s2;
  vc.invN = e.invN;
#pragma unroll
  for (int k = 0; k < 3; k++) vc.vbar[k] = e.mv[k] * vc.invN;
  vc.coe = obs ? e.coe : 0.0;
  vc.sc = obs ? e.sc : 0.0;
  dacc[27] += (e.ok && fi == 0) ? e.coe * e.lam0 : 0.0;  // residual += coe * lambda_0, once per voxel (voxel_map.hpp:234)
  vxm::k3_entry_emit<RT>(e.c, e.c + 6, e.c[9], R, p, vc, dacc, emit);
}

// Where a lane's three row pieces go inside a tile buffer (element offsets, plain row-major): row 3 vl + r of the wave's R rows,
// columns 6 fi .. 6 fi + 5 -- three 16-byte stores per row.  Idle lanes (NACT .. 63) write a dump area instead of branching.
struct K3RowOfs {
  int rp[3];
};
template <int W>
__device__ __forceinline__ K3RowOfs k3_row_offsets(int wave, int vl, int fi) {
  using C = K3Cfg<W>;
  K3RowOfs ro;
#pragma unroll
  for (int r = 0; r < 3; r++) ro.rp[r] = (wave * C::R + 3 * vl + r) * C::RS + 6 * fi;
  return ro;
}
template <bool MIXED>
__device__ __forceinline__ void k3_store_row(double* buf, const K3RowOfs& ro, int r, const double row[6]) {
#pragma unroll
  for (int j = 0; j < 3; j++) {
    // mixed precision: the Jacobian row is rounded to f32 here; everything downstream (products, sums) is f64
    const double a = MIXED ? (double)(float)row[2 * j] : row[2 * j], b = MIXED ? (double)(float)row[2 * j + 1] : row[2 * j + 1];
    *reinterpret_cast<v2d*>(buf + ro.rp[r] + 2 * j) = (v2d){a, b};
  }
}

// Phase M of wave WV: its pairs over `nch` 16-row slabs of the tile at `bp` (= buffer + the lane's operand offset
// (lane >> 2) RS + (lane & 3)).  FULL: nch == KC at compile time, operands of slab q + 1 requested before the instructions of slab q
// are issued, hook(q) behind them (the next batch's global requests ride there).  Every index below is a compile-time constant.
template <int W, int WV, int J>
__device__ __forceinline__ void k3_mfma_pairs(const double* x, double* acc) {
  using C = K3Cfg<W>;
  constexpr K3PairTab T = C::pairs();
  if constexpr (J < T.npair[WV]) {
    acc[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[T.sa[WV][J]], x[T.sb[WV][J]], acc[J], 0, 0, 0);
    k3_mfma_pairs<W, WV, J + 1>(x, acc);
  }
}
template <int W, int WV, int S>
__device__ __forceinline__ void k3_read_slots(const double* bp, int slab, double* x) {
  using C = K3Cfg<W>;
  constexpr K3PairTab T = C::pairs();
  if constexpr (S < T.nslot[WV]) {
    x[S] = bp[slab * 16 * C::RS + 4 * T.slot_group[WV][S]];
    k3_read_slots<W, WV, S + 1>(bp, slab, x);
  }
}
struct K3NoHook { __device__ __forceinline__ void operator()(int) const {} };
// A wave that owns only one to three pairs (every wave at W <= 2, waves of the last run at W = 3, 5, ..) issues DEPENDENT v_mfma_f64_4x4x4_4b
// instructions one to three issue slots apart.  The hardware interlocks such a chain only when SrcC and vDst are the same registers; when the
// register allocator gives the accumulator a new home (vDst != SrcC: it does, around the software-pipelined operand registers), the distance is
// the compiler's business -- and its table for this opcode is short on gfx950 (ROCm 7.2): the consumer read the accumulator BEFORE the producer
// had written it and a whole slab's products were lost, depending on when the operand reads happened to land (round 6: W = 2 in the fused
// residual + Hessian launch, pair (0, 0), 1 / 18 of S missing in 213 of 256 workgroups; the stand-alone kernel had the same instruction pattern
// and passed by timing).  Sixteen idle issue slots behind each slab of such a wave make the chain safe whatever the allocation; waves with the
// full 5 - 15 pairs per slab have their dependent instructions that far apart anyway and pay nothing.
template <int NPAIR>
__device__ __forceinline__ void k3_mfma_gap() {
  if constexpr (NPAIR >= 1 && NPAIR <= 3) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15");
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int W, int WV, bool FULL, class Hook>
__device__ __forceinline__ void k3_mfma_wave(const double* bp, int nch, double* acc, Hook& hook) {
  using C = K3Cfg<W>;
  constexpr K3PairTab T = C::pairs();
  constexpr int NS = T.nslot[WV] > 0 ? T.nslot[WV] : 1;
  if constexpr (FULL) {
    double x[NS], xn[NS];
    k3_read_slots<W, WV, 0>(bp, 0, x);
#pragma unroll
    for (int q = 0; q < C::KC; q++) {
      if (q + 1 < C::KC) {
        k3_read_slots<W, WV, 0>(bp, q + 1, xn);
        __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the MFMAs (the scheduler sinks them to their first use otherwise)
      }
      k3_mfma_pairs<W, WV, 0>(x, acc);
      k3_mfma_gap<T.npair[WV]>();
      hook(q);
#pragma unroll
      for (int s = 0; s < NS; s++) x[s] = xn[s];
    }
  } else {
    for (int q = 0; q < nch; q++) {
      double x[NS];
      k3_read_slots<W, WV, 0>(bp, q, x);
      k3_mfma_pairs<W, WV, 0>(x, acc);
      k3_mfma_gap<T.npair[WV]>();
    }
  }
}
template <int W, bool FULL, class Hook>
__device__ __forceinline__ void k3_mfma_phase(const double* bp, int wave, int nch, double* acc, Hook& hook) {
  switch (wave) {   // wave-uniform: scalar branches, eight straight-line variants
    case 0: k3_mfma_wave<W, 0, FULL>(bp, nch, acc, hook); break;
    case 1: k3_mfma_wave<W, 1, FULL>(bp, nch, acc, hook); break;
    case 2: k3_mfma_wave<W, 2, FULL>(bp, nch, acc, hook); break;
    case 3: k3_mfma_wave<W, 3, FULL>(bp, nch, acc, hook); break;
    case 4: k3_mfma_wave<W, 4, FULL>(bp, nch, acc, hook); break;
    case 5: k3_mfma_wave<W, 5, FULL>(bp, nch, acc, hook); break;
    case 6: k3_mfma_wave<W, 6, FULL>(bp, nch, acc, hook); break;
    default: k3_mfma_wave<W, 7, FULL>(bp, nch, acc, hook); break;
  }
}

// Epilogue: the 28 linear accumulators of every lane are parked with stride 29 (odd: conflict-free column reads), then one thread per
// (frame, slot) sums the 8 NV lanes that hold it.
constexpr int K3_DS = DACC + 1;
template <int W>
__device__ __forceinline__ void k3_sum_linear(const double* park_d, double* out, int tid) {
  using C = K3Cfg<W>;
  for (int el = tid; el < W * DACC; el += K3_BLOCK) {
    const int i = el / DACC, k = el % DACC;
    double sum = 0.0;
    for (int w = 0; w < C::WAVES; w++)
#pragma unroll
      for (int v = 0; v < C::NV; v++) sum += park_d[(w * 64 + v * W + i) * K3_DS + k];
    st_out(&out[i * DACC + k], sum);
  }
}

// Kernel arguments: the first 14 dwords are PRELOADED into SGPRs when a wave is launched (-mllvm -amdgpu-kernarg-preload-count=14 in the
// Makefile; gfx950 has 16 user SGPRs, two hold the kernarg pointer) -- everything a wave needs to request its first batch and the poses,
// so that those requests do not wait for a scalar load of the argument block first (a cold miss: the block was written by the host a few
// microseconds earlier).  Structs are not preloaded and stop the sequence, hence the flat list; what is not urgent follows as before.
//   pend_flags = pending | restart << 8, nwg = nwg (otherwise a load from the hidden arguments)
constexpr int K3_LATE_PER = 4;   // requests behind each slab of phase M: all eight behind the first two of the nine slabs at W = 10 (1 and 2 per K-step measured slower in round 4: later requests land later)
// The sweep of ONE workgroup.  FUSED = false: the body of k3_hessian_kernel (g = blockIdx.x).  FUSED = true: the Hessian half of the fused
// residual + Hessian launch (k23_fused_kernel, vxba_k23.hpp): the caller -- workgroup g of nwg sweep workgroups -- has already put the
// poses to linearise at into LDS (poseA), run the residual half over exactly this workgroup's voxels (cache planes rewritten, stores
// acknowledged) and passed a workgroup barrier; there is no LM decision here (the reduction kernel behind the launch takes it), and the plane
// parameters are read past this CU's vector L1 (AUX = sc1), which may still hold lines of the OLD eigenvector planes (the residual half's warm start).
// Only the padding columns 6W .. 4 NG of the two tile buffers have to start as zeros (odd W: two columns): phase A writes columns
// 0 .. 6W of every row of a step, the ragged step clears the rows it rounds up to, nothing else is read.
template <int W>
__device__ __forceinline__ void k3_clear_pads(double* lds, int tid) {
  using C = K3Cfg<W>;
  constexpr int PADC = C::NCOLS - 6 * W;
  if constexpr (PADC > 0) {
    for (int k = tid; k < 2 * C::ROWS * PADC; k += K3_BLOCK) {
      const int b = k / (C::ROWS * PADC), rr = (k / PADC) % C::ROWS, cc = 6 * W + k % PADC;
      lds[b * C::BUF + rr * C::RS + cc] = 0.0;
    }
  }
}

// FUSED: pre_c = the wave's first batch of clusters, requested by the caller (before the residual half's barrier; nullptr never);
// the caller has cleared the padding columns and passed a barrier, so the first phase A starts when the wave's own parameters have landed.
template <int W, bool DBG, bool MIXED, bool FUSED>
__device__ __forceinline__ void k3_sweep_body(double* lds, const double* __restrict__ clb, const double* __restrict__ cache_planes, const double* __restrict__ coe_plane,
                                              LMState* __restrict__ st, int VS, int head, int end, int c_in, int pend_flags, int nwg, int g, const PoseArg& poses,
                                              const LMPending& pend, double* __restrict__ partial, const double* pre_c = nullptr) {
  const int pending = pend_flags & 0xff, restart = pend_flags >> 8;
#ifdef VXBA_K23_DBG_PAUX0
  constexpr int PAUX = 0;
#else
  constexpr int PAUX = FUSED ? 16 : 0;   // cache policy of the plane-parameter loads
#endif

  using C = K3Cfg<W>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // tell the compiler it is wave-uniform: scalar branches, descriptors in SGPRs
  const bool active = lane < C::NACT;
  const int vl = active ? lane / W : 0;
  const int fi = active ? lane % W : 0;
  const int gw = g * C::WAVES + wave;
  dbg_stamp(DBG && !FUSED, gw, 0);

  // this workgroup's run of batches (absolute: batch b = voxels [b NV, (b+1) NV), so the batch-major copy does not depend on `head`)
  const int b0 = head / C::NV, b1 = (end - 1) / C::NV;
  const int nb_all = b1 - b0 + 1, G = nwg;
  const int q = nb_all / G, rem = nb_all % G;
  const int cnt = q + (g < rem ? 1 : 0);
  const int bs = b0 + g * q + (g < rem ? g : rem);

  // The first batch is requested before anything else: it does not depend on the poses, so the LM decision below (a few
  // dependent global reads) runs in the shadow of these loads.  Wave 0 is the exception: loads return in order, so it asks for
  // the control block FIRST and for its batch afterwards -- behind 12 MB of cold first-batch requests from the whole chip its
  // decision arrived 7.6k cycles into the kernel with every other wave waiting at the barrier.
  K3Entry e;
  e.ok = false;
  K3Planes pl;
  pl.cache_ptr = cache_planes;
  pl.coe_ptr = coe_plane;
  pl.clb = clb;
  pl.vs8 = (unsigned)VS * 8u;
  K3Stage<W> stg;
#pragma unroll
  for (int q = 0; q < K3Stage<W>::Q; q++) stg.v[q] = 0.0;
  stg.coe = 0.0;
  // The prologue, by wave:
  //   wave 0    fetches the poses to linearise at and the four control scalars, requests the residual-sweep partials whose sum
  //             (residual2) decides the pending step, then its first batch; the poses go to LDS once ALL of that has landed;
  //   waves 1-3 request their first batch;
  //   waves 4-7 request their first batch after the barrier below.
  // First batches in two halves: a CU takes in ~10 B per clock, so eight first batches (48 KB) land together and both waves of a SIMD
  // then run phase A back to back with nothing under it; with four, the first wave of each SIMD works while the second wave's rows are
  // on their way.  The barrier is released by wave 0, i.e. about when the first four batches have landed (round 3, same box, K3 at
  // cfg2: this arrangement 26.8 us; all eight batches before the barrier 27.3; poses and partials on waves 4 / 5 27.4; partials requested
  // after the barrier 27.2-27.5; second half delayed by a further 512 / 1024 / 2048 cycles 27.2 / 27.4 / 28.2).
  //
  // The accept / reject decision of the pending step (pending == 1) is SPECULATED: a rejected step means this sweep has nothing to
  // do, an accepted one means it linearises at the trial poses -- so every workgroup starts phase A of its first step at the trial
  // poses as soon as those are in LDS, wave 0 adds up residual2 behind the barrier, and the decision is taken behind the first step's
  // barrier (a rejected step costs one phase A instead of none; the common, accepted one no longer waits for the sum).
  constexpr int K3_FIRST_WAVES = C::WAVES / 2;   // (round 5, rebuilt sweep, same box: 2 / 6 / 8 waves in front of the barrier are within noise of 4 -- gpurun_out/r5_s5)
  if constexpr (!FUSED) {
    if (wave != 0 && wave < K3_FIRST_WAVES && wave < cnt) { k3_load_clusters(pl, bs + wave, lane, e.c); k3_load_params<W>(pl, head, end, bs + wave, lane, stg); }
  } else {
    // fused launch: the poses are in LDS and this workgroup's cache planes are final -- every wave asks for its first batch at once (the
    // lines come out of the L2 / the Infinity Cache: the residual half has just written the planes, the clusters were read a sweep ago)
    if (wave < cnt) {
#ifdef VXBA_K23_DBG_NOPRE
      k3_load_clusters(pl, bs + wave, lane, e.c);
#else
#pragma unroll
      for (int k = 0; k < 10; k++) e.c[k] = pre_c[k];
#endif
      // (no request for the first batch's plane parameters: the residual half left the 18-double records of the workgroup's first eight
      // batches in the waves' staging corners -- the first phase A reads them from there, see `first_a` below)
    }
#ifdef VXBA_K23_DBG_BARRIER
    __syncthreads();
#endif
  }

  // LDS behind the two tile buffers: the poses (raw C-ABI layout: R column-major | p per frame) and what the LM decision needs
  double* poseA = lds + 2 * C::BUF;       // the poses `xa_src` selects
  double* lmv = poseA + 24 * W;           // [0] done, [1] calc_hess, [2] bench_mode, [3] residual1, [4] residual2 (written after the barrier)
  double* stage_lds = lmv + 8 + wave * K3Stage<W>::WAVE_DOUBLES;   // this wave's corner for redistributing the plane parameters
  LMResidual2Loads r2_loads;
  const bool decide_here = !FUSED && st && pending == 1;
  if (!FUSED && wave == 0) {
    // The poses are one contiguous run of 12 W doubles, copied as such: ceil(12 W / 64) load instructions of consecutive lanes = 15 cache
    // lines per workgroup (one lane per frame and twelve strided loads each were 120 requests per workgroup for the same 15 lines).
    const double* __restrict__ xa_src = poses.Rp;
    double v_done = 0.0, v_calc = 0.0, v_bench = 0.0, r1 = 0.0;
    if (st) {
      const LMCtl& in = st->ctl[c_in];
      // pending: 0 none (linearise at in.x), 1 decide here (see above: trial poses, or the kernel-argument poses when a new window
      // starts), 2 / 3 sharded speculative loop: no decision here -- linearise at the trial poses (2) or at the kernel-argument poses
      // (3, first sweep of a solve / window), skip only when the loop is done
      xa_src = ((pending == 1 && restart) || pending == 3) ? poses.Rp : (pending ? in.xt : in.x);
      v_done = in.done; v_calc = in.calc_hess; v_bench = in.bench_mode; r1 = in.residual1;
    }
    constexpr int NPL = (12 * W + 63) / 64;
    double xa[NPL];
#pragma unroll
    for (int k = 0; k < NPL; k++) xa[k] = (64 * k + lane < 12 * W) ? xa_src[64 * k + lane] : 0.0;
    if (decide_here) lm_residual2_issue(pend, r2_loads);   // in flight across the barrier
    if (cnt > 0) { k3_load_clusters(pl, bs, lane, e.c); k3_load_params<W>(pl, head, end, bs, lane, stg); }
#pragma unroll
    for (int k = 0; k < NPL; k++)
      if (64 * k + lane < 12 * W) poseA[64 * k + lane] = xa[k];
    if (st && lane == 0) { lmv[0] = v_done; lmv[1] = v_calc; lmv[2] = v_bench; lmv[3] = r1; }
    dbg_stamp(DBG, gw, 5);   // poses in LDS
  }
  dbg_stamp(DBG, gw, 2);     // first requests issued
  if constexpr (!FUSED) {
    k3_clear_pads<W>(lds, tid);
    dbg_stamp(DBG, gw, 4);     // tiles cleared
    __syncthreads();
  }
  dbg_stamp(DBG, gw, 1);
  bool undecided = false;
  if (!FUSED && st) {
    const bool in_done = lmv[0] != 0.0, in_calc = lmv[1] != 0.0;
    if (pending >= 2) {
      if (in_done) return;
    } else if (pending) {
      if (in_done) { if (g == nwg - 1) lm_carry(st, c_in, W); return; }
      undecided = true;
    } else {
      if (in_done || !in_calc) return;
    }
  }
  if (!FUSED && wave >= K3_FIRST_WAVES && wave < cnt) { k3_load_clusters(pl, bs + wave, lane, e.c); k3_load_params<W>(pl, head, end, bs + wave, lane, stg); }
  if (!FUSED && wave == 0 && undecided) {
    const double r2 = lm_residual2_finish(pend, r2_loads);
    if (lane == 0) lmv[4] = r2;
  }
  // Behind the first barrier after the one above: every thread takes the same decision from the same numbers (the last workgroup also
  // works out the damping update and persists the control block for the kernels that follow).  True: nothing (more) to do here.
  auto decide = [&]() __attribute__((always_inline)) -> bool {
    undecided = false;
    const bool bench = lmv[2] != 0.0;
    const double r1 = lmv[3], r2 = lmv[4];
    const bool accept = (r1 - r2) > 0;
    const bool done = !bench && fabs((r1 - r2) / r1) < 1e-6;
    if (g == nwg - 1) {
      const LMDecision d = lm_decide(st->ctl[c_in], r2, restart);
      lm_persist(st, c_in, d, restart, poses, W);
    }
    // Leave nothing in flight behind this once-per-launch path: the loop head below is a join of "came through here" and "did not", and
    // the compiler's wait-count bookkeeping at a join is the union of what may be pending -- loads of lm_decide / lm_persist into the
    // registers the MFMA operands are read into put an s_waitcnt vmcnt(0) at the head of EVERY phase M (the next batch's loads, issued
    // a moment earlier, then had to land before the first MFMA instead of having the whole phase to do so).
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0); the builtin, not inline asm: the bookkeeping pass has to see it
    return done || !(accept || restart);
  };
  const double* pose = poseA + 12 * fi;
  double acc[C::PPW];                     // the wave's pairs: block t of accumulator j holds the rows == t (mod 4) of every slab
#pragma unroll
  for (int t = 0; t < C::PPW; t++) acc[t] = 0.0;
  double dacc[DACC];
#pragma unroll
  for (int k = 0; k < DACC; k++) dacc[k] = 0.0;

  K3RowOfs ro = k3_row_offsets<W>(wave, vl, fi);   // lane constants: where the lane's three row pieces go inside a tile buffer
  double* const dump = lmv + 8 + C::WAVES * K3Stage<W>::WAVE_DOUBLES;   // 32 doubles behind the staging areas that nobody reads: where the idle lanes' (zero) rows go
  // (k3_lds_bytes reserves them.  A first version put the dump behind the poses: at W <= 2 that ran into the LM decision inputs -- found by the randomised sweep)
  if (!active) {
#pragma unroll
    for (int k = 0; k < 3; k++) ro.rp[k] = 8 * k;
  }
  const int opnd = (lane >> 2) * C::RS + (lane & 3);    // the lane's place in every operand read: row lane / 4 of a slab, column lane % 4 of a group
  int dbg_step = -1;                                      // instrumented build: the step phase_a is running for
  // Phase A of the wave's batch b into tile buffer `bo`, then the requests for its next batch nb (nb < 0: none).
  // Measured and rejected (round 3, same box): the LDS round trip of the plane parameters issued two thirds of the way through phase M
  // instead of at the head of phase A -- 29.8 -> 29.6 us at cfg2, but 166.9 -> 173.3 us at cfg4 (13 steps per workgroup): the wait
  // for the parameter loads then sits inside the MFMA stream, and with more traffic in flight they have not always landed by then.
  // fused launch: the first phase A of a wave finds its records in the staging corner (written by the residual half, which this workgroup ran
  // over the same voxels a barrier ago); the cache planes in memory are for the later steps, the reduction kernel and the next launch
  bool first_a = FUSED;
  auto unstage = [&](int b) __attribute__((always_inline)) {
    k3_unstage_params<W>(stg, stage_lds, head, end, b, active, vl, lane, e, !(FUSED && first_a));
    first_a = false;
  };
  auto phase_a = [&](int b, int bo) __attribute__((always_inline)) {
    // instrumented build, step 2 only: 7 parameters + pose back in registers, 14 rows computed, 15 rows stored
    const bool stamp_here = DBG && dbg_step == 2;
    unstage(b);
    if (DBG && stamp_here) { asm volatile("" :: "v"(e.u[0]), "v"(e.coe)); __builtin_amdgcn_sched_barrier(0); dbg_stamp(true, gw, 7); __builtin_amdgcn_sched_barrier(0); }
    // Rows go to the tile AS THEY ARE FINISHED (z row, G row 1, G row 2; round 4): stamps put ten 16-byte stores per wave, issued together
    // at the end of phase A by all four waves of a half at once, at 1.0-1.5k cycles until the last is performed -- which the wave then
    // waited out in front of the barrier.  With the first six under way while the second G row and the block-diagonal terms are still
    // being computed only the last three are young at the barrier.
    // (no branch around the stores of the 64 - NACT idle lanes: they write a dump area instead -- a branch makes the wait-count
    // bookkeeping at its join wait for the stores just issued)
    double* const rowbase = active ? lds + bo : dump;
    auto emit = [&](int r, const double row[6]) __attribute__((always_inline)) {
      k3_store_row<MIXED>(rowbase, ro, r, row);
      __builtin_amdgcn_sched_barrier(0);   // keep the stores where they are: the scheduler gathers them behind the last row otherwise
    };
    k3_phase_a<true>(e, fi, pose, dacc, emit);
    if (DBG && stamp_here) { __builtin_amdgcn_sched_barrier(0); dbg_stamp(true, gw, 14); __builtin_amdgcn_sched_barrier(0); }
  };

  // Full steps: step s = phase M of step s-1 (buffer (s-1)&1), phase A of step s (buffer s&1), one barrier.  The ragged last
  // step (cnt mod 8 batches) is peeled off below, so inside the loop every wave has a batch and both phases are straight-line code.
  // The requests for the batch of step s ride behind the slabs of phase M of step s-1, K3_LATE_PER per slab, parameters first (phase A
  // starts with their LDS round trip): a vector-memory instruction costs the wave ~60 cycles of issue when eight waves queue on the CU's
  // one address unit (round-4 stamps: 460-540 cycles for the eight of a batch), and in front of the barrier that was on the step's
  // critical path.  The loads are UNCONDITIONAL and gated by their descriptor (`more_m` switches the range check off: out-of-range lanes
  // return zeros without a memory request): with a branch around them the entry registers became a phi of (old, loaded) values, and the
  // copies that resolve it sat behind s_waitcnt vmcnt at the END of phase A.
  const int nfull = cnt / C::WAVES, nrag = cnt - nfull * C::WAVES;
  for (int s = 0; s <= nfull; s++) {
    if (s >= 1) {
      const bool more_m = (s < nfull) || (wave < nrag);
      const int nb_m = bs + s * C::WAVES + wave;
      constexpr int NQ = K3Stage<W>::Q + 1, NL = 5 + NQ;
      constexpr int PER = (K3_LATE_PER * C::KC >= NL) ? K3_LATE_PER : (NL + C::KC - 1) / C::KC;
      auto hook = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < PER; r++) {
          const int l = kk * PER + r;
          if (l < NQ) k3_load_param_q<W, PAUX>(pl, head, end, nb_m, lane, stg, more_m, l);
          else if (l < NL) k3_load_cluster_row(pl, nb_m, lane, e.c, more_m, l - NQ);
        }
      };
      k3_mfma_phase<W, true>(lds + ((s - 1) & 1) * C::BUF + opnd, wave, C::KC, acc, hook);
      if (s <= 4) dbg_stamp(DBG, gw, 13 + 3 * s);   // phase M of step s-1 done: slots 16, 19, 22, 25
    }
    if (s == nfull) break;
    if (DBG) dbg_step = s;
    phase_a(bs + s * C::WAVES + wave, (s & 1) * C::BUF);
    if (s >= 1 && s <= 4) dbg_stamp(DBG, gw, 14 + 3 * s);   // phase A of step s done: slots 17, 20, 23, 26
    // fused launch: the residual half's cache stores (issued a phase A ago) acknowledged by every wave before the barrier behind which the
    // next step's plane parameters are requested -- from the L2, by whichever wave of the workgroup owns the batch
    if constexpr (FUSED) { if (s == 0) __builtin_amdgcn_s_waitcnt(0x0f70); }
    __syncthreads();
    if (s < 6) dbg_stamp(DBG, gw, 8 + s);
    if (undecided && decide()) return;
  }
  if (nrag > 0) {
    // ragged last step: nrag < 8 batches.  Only ceil(nrag R / 16) slabs exist; the first idle wave makes the rows that round the step
    // up to a whole slab read as zeros.
    const int bo = (nfull & 1) * C::BUF;
    const int nch = (nrag * C::R + 15) >> 4;
    if (wave < nrag) phase_a(bs + nfull * C::WAVES + wave, bo);
    else if (wave == nrag) {
      double* z = lds + bo + nrag * C::R * C::RS;
      for (int k = lane; k < (nch * 16 - nrag * C::R) * C::RS; k += 64) z[k] = 0.0;
    }
    if constexpr (FUSED) { if (nfull == 0) __builtin_amdgcn_s_waitcnt(0x0f70); }
    __syncthreads();
    if (nfull < 6) dbg_stamp(DBG, gw, 8 + nfull);
    if (undecided && decide()) return;
    K3NoHook nohook;
    k3_mfma_phase<W, false>(lds + bo + opnd, wave, nch, acc, nohook);
  }
  dbg_stamp(DBG, gw, 3);

  // Epilogue.  One partial per workgroup:  [8 waves x PPWP pairs x 16 | W x DACC].
  double* pout = partial + (size_t)g * C::PLEN;
  double* park_d = lds;                                                   // [512 lanes][K3_DS] linear accumulators
  __syncthreads();  // every wave is done with the tiles
  dbg_stamp(DBG, gw, 27);
  if (undecided && decide()) return;   // a workgroup without a batch: first barrier since the prologue's
  // (1) per-frame linear accumulators: every lane parks them, then (behind the barrier) one thread per (frame, slot) sums the 8 NV lanes
#pragma unroll
  for (int k = 0; k < DACC; k++) park_d[(wave * 64 + lane) * K3_DS + k] = dacc[k];
  // (2) the wave's own pairs.  Fold the four blocks (lanes 16 i + 4 t + j, t = 0..3 -- inside a 16-lane row: two DPP row rotations, no LDS
  // traffic; every lane of a quadruple ends up with the same sum bit for bit: (a + c) + (d + b) and (b + d) + (a + c) add the same two
  // partial sums), then lane (i, t, j) hands pair 4 m + t to a wave-private staging area -- four full-wave 8-byte LDS stores -- from which
  // the 16 PPW sums leave as 16-byte written-through stores (an 8-byte written-through store costs 2.7x the time per byte).
  {
    const int t4 = (lane >> 2) & 3;
    const int np = C::npair(wave);
    auto ror = [](double v, auto ctrl) __attribute__((always_inline)) -> double {
      const v2i x = __builtin_bit_cast(v2i, v);
      v2i y;
      y[0] = __builtin_amdgcn_update_dpp(0, x[0], decltype(ctrl)::value, 0xf, 0xf, false);
      y[1] = __builtin_amdgcn_update_dpp(0, x[1], decltype(ctrl)::value, 0xf, 0xf, false);
      return __builtin_bit_cast(double, y);
    };
#pragma unroll
    for (int j = 0; j < C::PPW; j++) {
      double v = acc[j];
      v += ror(v, std::integral_constant<int, 0x128>{});   // row_ror:8 -- blocks t and t + 2
      v += ror(v, std::integral_constant<int, 0x124>{});   // row_ror:4 -- ... and the other two
      acc[j] = v;
    }
    double* stage = lds + (size_t)K3_BLOCK * K3_DS + wave * (C::PPWP * 16);   // behind the parked linear accumulators
#pragma unroll
    for (int m = 0; m < C::PPWP / 4; m++) {
      double v = 0.0;
#pragma unroll
      for (int t = 0; t < 4; t++)
        if (4 * m + t < C::PPW) v = (t4 == t) ? acc[4 * m + t] : v;
      stage[(4 * m + t4) * 16 + 4 * (lane >> 4) + (lane & 3)] = v;
    }
    __builtin_amdgcn_wave_barrier();   // same wave, LDS operations execute in order
    const __amdgpu_buffer_rsrc_t rout = k3_rsrc(pout + wave * (C::PPWP * 16));
#pragma unroll
    for (int r = 0; r < (C::PPWP * 8 + 63) / 64; r++) {
      const int el2 = 64 * r + lane;                     // pair of doubles el2 of the wave's PPWP * 16
      if (el2 < np * 8) {
        const v2d v = *reinterpret_cast<const v2d*>(stage + 2 * el2);
#if VXBA_WT_STORES
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), rout, el2 * 16, 0, 16);   // aux 16 = sc1 on gfx950
#else
        *reinterpret_cast<v2d*>(pout + wave * (C::PPWP * 16) + 2 * el2) = v;
#endif
      }
    }
  }
  dbg_stamp(DBG, gw, 28);
  __syncthreads();
  dbg_stamp(DBG, gw, 29);
  k3_sum_linear<W>(park_d, pout + C::NTILE, tid);
  dbg_stamp(DBG, gw, 6);
  if (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_stamp(true, gw, 31); }   // ... and acknowledged
}

template <int W, bool DBG = false, bool MIXED = false>
__global__ __launch_bounds__(K3_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k3_hessian_kernel(const double* __restrict__ clb, const double* __restrict__ cache_planes, const double* __restrict__ coe_plane,
                                                              LMState* __restrict__ st, int VS, int head, int end, int c_in, int pend_flags, int nwg,
                                                              PoseArg poses, LMPending pend, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double lds[];  // two tile buffers; reused by the epilogue
  k3_sweep_body<W, DBG, MIXED, false>(lds, clb, cache_planes, coe_plane, st, VS, head, end, c_in, pend_flags, nwg, (int)blockIdx.x, poses, pend, partial);
}
