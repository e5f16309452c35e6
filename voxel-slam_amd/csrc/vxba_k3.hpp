// K3 -- Hessian / gradient sweep (LidarFactor::acc_evaluate2, voxel_map.hpp:132-241).  Included by vxba_kernels.hip inside
// namespace vxk, after the LM helpers (lm_decide / lm_residual2 / lm_carry / lm_persist) and dbg_stamp.
//
// Per voxel  H_a = -B_a^T B_a + blockdiag_i(D_{a,i})  with B_a the 3 x 6W matrix of SURVEY A.4, so the window Hessian is a
// tall-skinny SYRK over the 3V stacked rows: v_mfma_f64_16x16x4_f64 accumulates the upper-triangular 16 x 16 tile pairs of
// S = sum B^T B; gradient, block-diagonal terms and the residual are 28 linear accumulators per frame.
//
// Round-2 mapping: ONE 8-wave workgroup per CU, two waves per SIMD.  On this chip the f64 MFMA and the f64 VALU share one
// datapath (strictly additive inside a SIMD, scripts/ubench/mfma_valu_overlap.hip), so the kernel is bound by its fp64 issue
// count; what a second wave per SIMD buys is that everything that is NOT fp64 -- LDS operand reads and row stores, address
// arithmetic, waits on global loads, the barrier -- runs under the other wave's fp64 work.  Two waves per SIMD means 256
// registers per lane, which the round-1 kernel (80 accumulator registers + two entry sets) did not fit.  Here the eight
// waves of the workgroup share the MFMA work of a STEP of eight batches instead of each wave doing all of its own:
//   phase A  every wave turns its own batch (NV voxels x W frames, one lane per entry) into 3 NV rows of B and stores them
//            in a workgroup-shared LDS tile of 8 x 3 NV rows (144 rows at W = 10: 36 K-steps, no K padding -- the round-1
//            kernel padded each batch's 18 rows to 20);
//   phase M  wave w owns tile set (w mod TSPLIT) -- half of the tile pairs at W >= 6 -- and the K range (w div TSPLIT) of the
//            step: 45 MFMAs per wave and step at W = 10 (it was 50 per batch), 40 accumulator registers instead of 80.
// The tile is double-buffered and the loop is skewed -- phase M of step s-1, then phase A of step s, then ONE barrier -- so a
// wave leaves the barrier straight into MFMAs whose operands only need an LDS read.  A single entry register set suffices:
// the loads of the next batch are issued right after phase A has consumed the current one and have the whole phase M to land.
// Workgroups take contiguous, evenly sized runs of batches (32 or 33 at cfg2); the run's last, partly filled step only
// covers the rows that exist (its K range is re-split over the waves), so the ragged end costs one phase A, not a step.
// The cross-wave reduction of the epilogue and the workgroup partial (k3_finalize's input) are laid out as in round 1.
// No float atomics: bitwise reproducible for a given launch geometry.
#pragma once
#include <type_traits>

template <int W>
struct K3Cfg {
  static constexpr int NT = (6 * W + 15) / 16;        // 16-wide column tiles
  static constexpr int NTP = NT * (NT + 1) / 2;       // upper-triangular tile pairs
  static constexpr int NCOL = 16 * NT;
  static constexpr int NVCAP = (NT <= 2) ? 12 : (NT == 3 ? 8 : 6);   // keeps two tile buffers of 8 batches within 144 KB of LDS
  static constexpr int NV = (64 / W) < NVCAP ? (64 / W) : NVCAP;     // voxels per wave-batch (== k3_nv(W))
  static constexpr int NACT = NV * W;                 // active lanes
  static constexpr int R = 3 * NV;                    // rows of B per batch
  static constexpr int WAVES = K3_BLOCK / 64;         // 8
  static constexpr int ROWS = WAVES * R;              // rows per step: 288 / 192 / 144, always a multiple of 4
  static constexpr int KS = ROWS / 4;                 // MFMA K-steps per full step
  static constexpr int TSPLIT = (NTP >= 6) ? 2 : 1;   // tile sets
  static constexpr int KSPLIT = WAVES / TSPLIT;       // K ranges
  static constexpr int TPW = NTP / TSPLIT;            // tile pairs (accumulators) per wave: 1, 3, 3, 5
  static constexpr int KPW = KS / KSPLIT;             // K-steps per wave and full step: 9, 9, 12, 9
  // Tile layout: rows are stored in PAIRS, column tiles interleaved -- element (row, col) sits at
  //   (row >> 1) * 2 NCOL + (col >> 4) * 32 + (row & 1) * 16 + (col & 15)
  // so the two rows a half-wave reads for one MFMA operand (ds_read_b64 is served 32 lanes at a time over 64 banks) are 32
  // CONTIGUOUS doubles: conflict-free without padding (which is what lets two buffers fit), and a lane's operand address is
  // one register plus immediates for the K-step and the column tile.
  // Spare (padding) columns 6W .. 6W+2 of the z rows carry sqrt2 sqrt(coe) u: the MFMA then delivers the block-diagonal
  // terms Drt, Dtt in S[.][6W + k] for free (vxm::k3_entry<false>) -- 15 accumulators (30 registers) and 22 fp64 operations
  // per entry less.  Window sizes without three spare columns (W = 5, 8) keep the register accumulators.
  static constexpr bool SPARE = (NCOL - 6 * W) >= 3;
  static constexpr int BUF = ROWS * NCOL;             // doubles (f64) or floats (mixed) per tile buffer
  __host__ __device__ static constexpr int at(int row, int col) { return (row >> 1) * 2 * NCOL + (col >> 4) * 32 + (row & 1) * 16 + (col & 15); }
  static_assert(ROWS % 4 == 0 && NTP % TSPLIT == 0 && KS % KSPLIT == 0, "step geometry");

  // Which tile pairs a wave multiplies.  Both tile sets run the SAME instruction stream: tile j of a wave multiplies operand
  // slot pa(j) (rows) with operand slot pb(j) (columns); only the column tile an operand slot reads differs between the sets
  // (a wave-uniform LDS offset).  With a run-time branch between two differently shaped sets the accumulators changed registers
  // at every merge point -- ~100 v_mov_b64 per wave and step.
  //   NT = 4: the six off-diagonal pairs of four column tiles split into two PATHS, 0-1-2-3 and 2-0-3-1 (the path graph on four
  //           nodes is self-complementary), each set taking the diagonal tiles of its path's two inner nodes: slots a-b-c-d,
  //           pairs (a,b) (b,c) (c,d) (b,b) (c,c).  Set 1 therefore holds (2,0) and (3,1) as LOWER tiles; k3_finalize and the
  //           spare-column lookup go through rowtile() / coltile().
  //   NT = 3: no symmetric split exists (three diagonal tiles); every tile gets its own two operand slots.
  //   NT <= 2: one set, slots = column tiles.
  static constexpr int NSLOT = (NT == 4) ? 4 : (NT == 3 ? 6 : NT);
  __host__ __device__ static constexpr int pa(int j) { return NT == 4 ? (j < 3 ? j : j - 2) : (NT == 3 ? 2 * j : k3_tile_I_(NT, j)); }
  __host__ __device__ static constexpr int pb(int j) { return NT == 4 ? (j < 3 ? j + 1 : j - 2) : (NT == 3 ? 2 * j + 1 : k3_tile_J_(NT, j)); }
  __host__ __device__ static constexpr int slot_tile(int set, int k) {
    if (NT == 4) return set == 0 ? k : (k == 0 ? 2 : (k == 1 ? 0 : (k == 2 ? 3 : 1)));
    if (NT == 3) return set == 0 ? (k < 3 ? 0 : 1) : (k == 2 ? 0 : (k == 4 ? 1 : 2));   // set 0: (0,0) (0,1) (1,1); set 1: (2,2) (0,2) (1,2)
    return k;
  }
  // tile t = set * TPW + j accumulates S[16 rowtile + i][16 coltile + jj]
  __host__ __device__ static constexpr int rowtile(int t) { return slot_tile(t / TPW, pa(t % TPW)); }
  __host__ __device__ static constexpr int coltile(int t) { return slot_tile(t / TPW, pb(t % TPW)); }
  // offset inside a workgroup partial of S[r][c], r <= c (f64 MFMA accumulator layout: register (row >> 2), lane ((row & 3) << 4) | col)
  __host__ __device__ static constexpr int elem_offset(int r, int c) {
    const int I = r >> 4, J = c >> 4;
    for (int t = 0; t < NTP; t++) {
      int row = -1, col = -1;
      if (rowtile(t) == I && coltile(t) == J) { row = r - 16 * I; col = c - 16 * J; }
      else if (rowtile(t) == J && coltile(t) == I) { row = c - 16 * J; col = r - 16 * I; }
      if (row >= 0) return t * 256 + (row >> 2) * 64 + ((row & 3) << 4) + col;
    }
    return -1;
  }
  __host__ __device__ static constexpr int k3_tile_I_(int nt, int t) { int I = 0; while (t >= nt - I) { t -= nt - I; I++; } return I; }
  __host__ __device__ static constexpr int k3_tile_J_(int nt, int t) { int I = 0; while (t >= nt - I) { t -= nt - I; I++; } return I + t; }
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
__device__ __forceinline__ double k3_ld64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
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
template <int W>
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
    st.v[q] = k3_ld64(rcache, (unsigned)plane * pl.vs8 + (unsigned)a * 8u, 0);
  }
  {
    int a = b * S::NV + (lane < S::NV ? lane : 0);
    a = (a >= head && a < end) ? a : head;
    st.coe = k3_ld64(rcoe, (unsigned)a * 8u, 0);
  }
}
// request q of the transposed plane parameters (q < Q) or the coe plane (q == Q): the interleaved form of k3_load_params
template <int W>
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
    st.v[q] = k3_ld64(rcache, (unsigned)plane * pl.vs8 + (unsigned)a * 8u, 0);
  } else {
    const __amdgpu_buffer_rsrc_t rcoe = k3_rsrc_gated(pl.coe_ptr, on);
    int a = b * S::NV + (lane < S::NV ? lane : 0);
    a = (a >= head && a < end) ? a : head;
    st.coe = k3_ld64(rcoe, (unsigned)a * 8u, 0);
  }
}
// park the staged values in the wave's LDS corner and read back the lane's own voxel record
template <int W>
__device__ __forceinline__ void k3_unstage_params(const K3Stage<W>& st, double* stage_lds, int head, int end, int b, bool active, int vl, int lane,
                                                  K3Entry& e) {
  using S = K3Stage<W>;
  using C = K3Cfg<W>;
#pragma unroll
  for (int q = 0; q < S::Q; q++) {
    const int t = lane + 64 * q;
    if (t < S::NITEM) {
      const int v = t / 17, k = t - 17 * v;
      stage_lds[v * S::REC + k] = st.v[q];
    }
  }
  if (lane < S::NV) stage_lds[lane * S::REC + 17] = st.coe;
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

// Store addresses of a lane's three row pieces (element offsets inside a tile buffer): lane constants, kept as 3 row parts +
// 3 column parts (at(row, col) separates) and added at store time.
struct K3RowOfs {
  int rp[3];   // row r of the lane's voxel
  int cp[3];   // column pair j of the lane's frame
};
template <int W>
__device__ __forceinline__ K3RowOfs k3_row_offsets(int wave, int vl, int fi) {
  using C = K3Cfg<W>;
  K3RowOfs ro;
  // (R is even, so a block of R rows starts on a row pair: at(wave R + x, 0) = at(x, 0) + wave R NCOL -- the pair-sync variant below re-bases
  // the same offsets by adding a multiple of R NCOL to the buffer pointer)
#pragma unroll
  for (int r = 0; r < 3; r++) ro.rp[r] = C::at(wave * C::R + 3 * vl + r, 0);
#pragma unroll
  for (int j = 0; j < 3; j++) ro.cp[j] = C::at(0, 6 * fi + 2 * j);
  return ro;
}
__device__ __forceinline__ void k3_store_row(double* buf, const K3RowOfs& ro, int r, const double row[6]) {
#pragma unroll
  for (int j = 0; j < 3; j++) *reinterpret_cast<v2d*>(buf + ro.rp[r] + ro.cp[j]) = (v2d){row[2 * j], row[2 * j + 1]};
}
__device__ __forceinline__ void k3_store_row_f32(float* buf, const K3RowOfs& ro, int r, const double row[6]) {
#pragma unroll
  for (int j = 0; j < 3; j++) *reinterpret_cast<v2f*>(buf + ro.rp[r] + ro.cp[j]) = (v2f){(float)row[2 * j], (float)row[2 * j + 1]};
}
__device__ __forceinline__ void k3_store_rows(double* buf, const K3RowOfs& ro, const double rows[3][6]) {
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int j = 0; j < 3; j++) *reinterpret_cast<v2d*>(buf + ro.rp[r] + ro.cp[j]) = (v2d){rows[r][2 * j], rows[r][2 * j + 1]};
}
// mixed precision (BASELINE configs[2]): the rows are rounded to f32 on the way into the tile (same geometry, in floats)
__device__ __forceinline__ void k3_store_rows_f32(float* buf, const K3RowOfs& ro, const double rows[3][6]) {
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int j = 0; j < 3; j++) *reinterpret_cast<v2f*>(buf + ro.rp[r] + ro.cp[j]) = (v2f){(float)rows[r][2 * j], (float)rows[r][2 * j + 1]};
}

// Phase M: K-steps [k0, k0 + nk) of the tile in `buf` into the wave's accumulators.  Lane l supplies row 4k + l/16, column
// 16c + l%16 of the column tile c its operand slot reads -- one register serves as A and as B operand.  FULL: nk == KPW at
// compile time.  (The slot indices are template constants: as plain constexpr calls inside the loop they were evaluated at run time.)
template <int W, int J>
__device__ __forceinline__ void k3_mfma_tiles(const double* x, v4d* acc) {
  using C = K3Cfg<W>;
  if constexpr (J < C::TPW) {
    acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[C::pa(J)], x[C::pb(J)], acc[J], 0, 0, 0);
    k3_mfma_tiles<W, J + 1>(x, acc);
  }
}
// element offset of operand slot k's column tile for tile set `set` (wave-uniform select between two constants)
template <int W>
__device__ __forceinline__ int k3_slot_offset(int set, int k) {
  using C = K3Cfg<W>;
  return 32 * (set == 0 ? C::slot_tile(0, k) : C::slot_tile(C::TSPLIT - 1, k));
}
struct K3NoHook { __device__ __forceinline__ void operator()(int) const {} };
template <int W, bool FULL, class Hook = K3NoHook>
__device__ __forceinline__ void k3_mfma_phase(const double* buf, int set, int k0, int nk, int lrow, int lcol, v4d* acc, Hook hook = Hook()) {
  using C = K3Cfg<W>;
  const double* bp[C::NSLOT];   // K-step kk: + kk * 4 NCOL
#pragma unroll
  for (int k = 0; k < C::NSLOT; k++) bp[k] = buf + C::at(4 * k0 + lrow, lcol) + k3_slot_offset<W>(set, k);
#if defined(K3_PREFETCH2) && K3_PREFETCH2
  if (FULL) {
    // experiment (the review's (b)): operands TWO K-steps ahead of the MFMAs that consume them
    double x[C::NSLOT], xn[C::NSLOT], xnn[C::NSLOT];
#pragma unroll
    for (int k = 0; k < C::NSLOT; k++) { x[k] = bp[k][0]; xn[k] = C::KPW > 1 ? bp[k][4 * C::NCOL] : 0.0; }
#pragma unroll
    for (int kk = 0; kk < C::KPW; kk++) {
      if (kk + 2 < C::KPW) {
#pragma unroll
        for (int k = 0; k < C::NSLOT; k++) xnn[k] = bp[k][(kk + 2) * 4 * C::NCOL];
        __builtin_amdgcn_sched_barrier(0);
      }
      k3_mfma_tiles<W, 0>(x, acc);
      hook(kk);
#pragma unroll
      for (int k = 0; k < C::NSLOT; k++) { x[k] = xn[k]; xn[k] = xnn[k]; }
    }
  } else
#endif
  if (FULL) {
    // operands of K-step kk+1 are requested before the MFMAs of K-step kk are issued
    double x[C::NSLOT], xn[C::NSLOT];
#pragma unroll
    for (int k = 0; k < C::NSLOT; k++) x[k] = bp[k][0];
#pragma unroll
    for (int kk = 0; kk < C::KPW; kk++) {
      if (kk + 1 < C::KPW) {
#pragma unroll
        for (int k = 0; k < C::NSLOT; k++) xn[k] = bp[k][(kk + 1) * 4 * C::NCOL];
        __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the MFMAs (the scheduler sinks them to their first use otherwise)
      }
      k3_mfma_tiles<W, 0>(x, acc);
      hook(kk);   // (K3_LATE_REQUESTS) one of the next batch's requests behind the MFMAs of this K-step
#pragma unroll
      for (int k = 0; k < C::NSLOT; k++) x[k] = xn[k];
    }
  } else {
    for (int kk = 0; kk < nk; kk++) {
      double x[C::NSLOT];
#pragma unroll
      for (int k = 0; k < C::NSLOT; k++) x[k] = bp[k][kk * 4 * C::NCOL];
      k3_mfma_tiles<W, 0>(x, acc);
    }
  }
}
// Mixed precision: f32 products on v_mfma_f32_16x16x4_f32 (32 cycles per instruction instead of 64), summed in f32 over the
// steps of ONE wave (<= 5 steps x 9 K-steps x 4 rows at cfg2/cfg3 sizes), then carried in f64 through the workgroup epilogue,
// the cross-workgroup reduction and the all-reduce -- "fp32 Jacobian, fp64 Hessian accumulation".  The f32 instruction leaves
// D(4 (l/16) + r, l % 16) in register r of lane l, the f64 one D((l/16) + 4 r, l % 16); feeding the A operand with the rows
// permuted by  m -> (m >> 2) + 4 (m & 3)  makes the two maps coincide, so the epilogue and k3_finalize are shared.
template <int W, int J>
__device__ __forceinline__ void k3_mfma_tiles_f32(const float* xa, const float* xb, v4f* af) {
  using C = K3Cfg<W>;
  if constexpr (J < C::TPW) {
    af[J] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[C::pa(J)], xb[C::pb(J)], af[J], 0, 0, 0);
    k3_mfma_tiles_f32<W, J + 1>(xa, xb, af);
  }
}
template <int W>
__host__ __device__ constexpr unsigned k3_slots_as_a() { unsigned m = 0; for (int j = 0; j < K3Cfg<W>::TPW; j++) m |= 1u << K3Cfg<W>::pa(j); return m; }
template <int W>
__host__ __device__ constexpr unsigned k3_slots_as_b() { unsigned m = 0; for (int j = 0; j < K3Cfg<W>::TPW; j++) m |= 1u << K3Cfg<W>::pb(j); return m; }
template <int W, bool FULL>
__device__ __forceinline__ void k3_mfma_phase_f32(const float* buf, int set, int k0, int nk, int lrow, int lcol, v4f* af) {
  using C = K3Cfg<W>;
  constexpr unsigned NEEDA = k3_slots_as_a<W>(), NEEDB = k3_slots_as_b<W>();
  const int pcol = (lcol >> 2) + 4 * (lcol & 3);
  const float* bpa[C::NSLOT];
  const float* bpb[C::NSLOT];
#pragma unroll
  for (int k = 0; k < C::NSLOT; k++) {
    bpb[k] = buf + C::at(4 * k0 + lrow, lcol) + k3_slot_offset<W>(set, k);
    bpa[k] = buf + C::at(4 * k0 + lrow, pcol) + k3_slot_offset<W>(set, k);
  }
  const int kend = FULL ? C::KPW : nk;
  if (FULL) {
#pragma unroll
    for (int kk = 0; kk < C::KPW; kk++) {
      float xa[C::NSLOT], xb[C::NSLOT];
#pragma unroll
      for (int k = 0; k < C::NSLOT; k++) {
        xb[k] = ((NEEDB >> k) & 1) ? bpb[k][kk * 4 * C::NCOL] : 0.0f;
        xa[k] = ((NEEDA >> k) & 1) ? bpa[k][kk * 4 * C::NCOL] : 0.0f;
      }
      k3_mfma_tiles_f32<W, 0>(xa, xb, af);
    }
  } else {
    for (int kk = 0; kk < kend; kk++) {
      float xa[C::NSLOT], xb[C::NSLOT];
#pragma unroll
      for (int k = 0; k < C::NSLOT; k++) {
        xb[k] = ((NEEDB >> k) & 1) ? bpb[k][kk * 4 * C::NCOL] : 0.0f;
        xa[k] = ((NEEDA >> k) & 1) ? bpa[k][kk * 4 * C::NCOL] : 0.0f;
      }
      k3_mfma_tiles_f32<W, 0>(xa, xb, af);
    }
  }
}

// Epilogue geometry.  With the spare columns in use only 13 of the 28 linear accumulators exist (g 0..5, Drr 6..11, residual 27);
// parked with stride 13 (odd: conflict-free column reads) they fit beside the parked MFMA accumulators, so the epilogue needs one
// barrier instead of three.  Slots that are not in use are never written to the partial and never read by k3_finalize.
template <int W>
struct K3Epi {
  using C = K3Cfg<W>;
  static constexpr int NUSED = C::SPARE ? 13 : DACC;
  static constexpr int DS = C::SPARE ? 13 : DACC + 1;
  __host__ __device__ static constexpr int slot(int k) { return C::SPARE ? (k < 12 ? k : 27) : k; }
  static constexpr bool ONE_PHASE = ((size_t)K3_BLOCK * DS + (size_t)C::WAVES * C::TPW * 256) * sizeof(double) <= 144 * 1024;
};
template <int W>
__device__ __forceinline__ void k3_sum_linear(const double* park_d, double* out, int tid) {
  using C = K3Cfg<W>;
  using E = K3Epi<W>;
  for (int el = tid; el < W * E::NUSED; el += K3_BLOCK) {
    const int i = el / E::NUSED, k = el % E::NUSED;
    double sum = 0.0;
    for (int w = 0; w < C::WAVES; w++)
#pragma unroll
      for (int v = 0; v < C::NV; v++) sum += park_d[(w * 64 + v * W + i) * E::DS + k];
    st_out(&out[i * DACC + E::slot(k)], sum);
  }
}

// Kernel arguments: the first 14 dwords are PRELOADED into SGPRs when a wave is launched (-mllvm -amdgpu-kernarg-preload-count=14 in the
// Makefile; gfx950 has 16 user SGPRs, two hold the kernarg pointer) -- everything a wave needs to request its first batch and the poses,
// so that those requests do not wait for a scalar load of the argument block first (a cold miss: the block was written by the host a few
// microseconds earlier).  Structs are not preloaded and stop the sequence, hence the flat list; what is not urgent follows as before.
//   pend_flags = pending | restart << 8, nwg = nwg (otherwise a load from the hidden arguments)
// K3_PAIR_SYNC (round 4, the review's experiment (a)): the two waves that share a SIMD (w and w + 4: a workgroup's waves go to the four SIMDs
// cyclically) own ONE K range -- the 2 R rows those same two waves produce in phase A -- and the two tile sets, so a step's dependency is
// local to the pair: a flag in LDS replaces the workgroup barrier from step 1 on (step 0 keeps it: the LM decision rides on it).
// MEASURED, same box (gpurun_out/r4_s11.log, profiles/r04_k3_phase_a): K3 26.2 us against 25.8 (cfg2), 148.6 against 146.6 (cfg4), 43.1
// against 42.6 (cfg3) -- no gain: the ~1k cycles a wave of the first half waits at the step barrier are spent waiting for ITS OWN partner
// (stamps: waves 0-3 finish phase A at 8.7k cycles of a step, their partners 4-7 at 9.9k, the barrier opens at 10.2k), which a pair flag
// waits for just the same, and the polling wave takes issue slots from the partner it waits for.  Off; kept as the record of the experiment.
// K3_PREFETCH2 ((b): operands two K-steps ahead): 26.0 / 146.8 / 42.6 -- no change; the MFMA stream does not wait for its operands.
#ifndef K3_PAIR_SYNC
#define K3_PAIR_SYNC 0
#endif
#ifndef K3_LATE_REQUESTS
#define K3_LATE_REQUESTS 1   // round 4: the default (cfg4 156.6 -> 147.4 us, cfg3 44.5 -> 43.2, cfg2 unchanged; same-box A/B, gpurun_out/r4_s3.log)
#endif
constexpr bool K3_LATE_REQ = K3_LATE_REQUESTS != 0;
#ifndef K3_LATE_MIXED
#define K3_LATE_MIXED 0
#endif
// K3_RAGGED_FIRST (round 4, experiment): the partly filled step of a workgroup (cnt mod 8 batches, one on 142 of 256 workgroups at cfg2) is
// taken in the FILL instead of behind the last full step: waves 4 .. request those batches before the prologue barrier (they would
// otherwise wait for the barrier with nothing in flight), run their phase A into the idle tile buffer while the first full batches are
// still on their way, and the short phase M + a barrier of its own sit in front of step 1.
#ifndef K3_RAGGED_FIRST
#define K3_RAGGED_FIRST 0
#endif
#ifndef K3_LATE_PER
#define K3_LATE_PER 4      // requests behind each K-step of phase M: all eight behind the first two of the nine K-steps at W = 10 (1 and 2 per K-step measured slower: later requests land later)
#endif
template <int W, bool DBG = false, bool MIXED = false>
__global__ __launch_bounds__(K3_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k3_hessian_kernel(const double* __restrict__ clb, const double* __restrict__ cache_planes, const double* __restrict__ coe_plane,
                                                              LMState* __restrict__ st, int VS, int head, int end, int c_in, int pend_flags, int nwg,
                                                              PoseArg poses, LMPending pend, double* __restrict__ partial) {
  const int pending = pend_flags & 0xff, restart = pend_flags >> 8;

  using C = K3Cfg<W>;
  // window sizes without spare tile columns (W = 5, 8) carry 15 more accumulators: no registers left for requests in flight across phase M
  // (mixed precision: phase M is half as long -- f32 products -- and the requests in front of it landed late: cfg3 mixed 35.6 -> 37.0 us; it keeps them in front of the barrier)
  constexpr bool LATE = K3_LATE_REQ && C::SPARE && !(MIXED && !K3_LATE_MIXED);
  extern __shared__ __attribute__((aligned(16))) double lds[];  // two tile buffers; reused by the epilogue
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // tell the compiler it is wave-uniform: scalar branches, descriptors in SGPRs
  const bool active = lane < C::NACT;
  const int vl = active ? lane / W : 0;
  const int fi = active ? lane % W : 0;
  const int lrow = lane >> 4, lcol = lane & 15;
  constexpr bool PAIR = (K3_PAIR_SYNC != 0) && C::TSPLIT == 2 && C::WAVES == 8 && (C::R % 2 == 0);
  const int set = PAIR ? (wave >> 2) : wave % C::TSPLIT, kq = PAIR ? (wave & 3) : wave / C::TSPLIT;
  // full steps: where this wave's R rows go inside a tile buffer, as a shift of the plain layout (rows wave R ..): K range kq = rows
  // [2 R kq, 2 R (kq + 1)) = the rows of wave kq (first half) and of wave kq + 4 (second half)
  const int rowshift = PAIR ? ((2 * kq + set) - wave) * C::R * C::NCOL : 0;
  const int gw = blockIdx.x * C::WAVES + wave;
  dbg_stamp(DBG, gw, 0);

  // this workgroup's run of batches (absolute: batch b = voxels [b NV, (b+1) NV), so the batch-major copy does not depend on `head`)
  const int b0 = head / C::NV, b1 = (end - 1) / C::NV;
  const int nb_all = b1 - b0 + 1, G = nwg;
  const int q = nb_all / G, rem = nb_all % G;
  const int g = blockIdx.x;
  const int cnt = q + (g < rem ? 1 : 0);
  const int bs = b0 + g * q + (g < rem ? g : rem);
  // (K3_RAGGED_FIRST) the ragged batches of this workgroup go first, on waves 4 .. 4 + nrag - 1
  constexpr bool RAGF_ON = (K3_RAGGED_FIRST != 0) && C::WAVES == 8 && (C::R % 2 == 0) && !MIXED;
  const bool ragf = RAGF_ON && (cnt / C::WAVES) >= 1 && (cnt % C::WAVES) >= 1 && (cnt % C::WAVES) <= 4;
  const bool ragw = ragf && wave >= 4 && (wave - 4) < (cnt % C::WAVES);
  const int b_rag = bs + (cnt / C::WAVES) * C::WAVES + (wave - 4);

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
  // on their way.  The barrier is released by wave 0, i.e. about when the first four batches have landed -- measured on the same box
  // (K3 at cfg2, us): this arrangement 26.8; all eight batches before the barrier 27.3; poses and partials on waves 4 / 5 (barrier 1.5k
  // cycles earlier) 27.4; partials requested after the barrier 27.2-27.5; second half delayed by a further 512 / 1024 / 2048 cycles
  // 27.2 / 27.4 / 28.2.
  //
  // The accept / reject decision of the pending step (pending == 1) is SPECULATED: a rejected step means this sweep has nothing to
  // do, an accepted one means it linearises at the trial poses -- so every workgroup starts phase A of its first step at the trial
  // poses as soon as those are in LDS, wave 0 adds up residual2 behind the barrier, and the decision is taken behind the first step's
  // barrier (a rejected step costs one phase A instead of none; the common, accepted one no longer waits for the sum).
#ifndef K3_FIRST_WAVES_V
#define K3_FIRST_WAVES_V (C::WAVES / 2)
#endif
  constexpr int K3_FIRST_WAVES = K3_FIRST_WAVES_V;
  if (wave != 0 && wave < K3_FIRST_WAVES && wave < cnt) { k3_load_clusters(pl, bs + wave, lane, e.c); k3_load_params<W>(pl, head, end, bs + wave, lane, stg); }
  if (RAGF_ON && ragw) { k3_load_clusters(pl, b_rag, lane, e.c); k3_load_params<W>(pl, head, end, b_rag, lane, stg); }

  // LDS behind the two tile buffers: the poses (raw C-ABI layout: R column-major | p per frame) and what the LM decision needs
  double* poseA = lds + 2 * C::BUF;       // the poses `xa_src` selects
  double* lmv = poseA + 24 * W;           // [0] done, [1] calc_hess, [2] bench_mode, [3] residual1, [4] residual2 (written after the barrier)
  double* stage_lds = lmv + 8 + wave * K3Stage<W>::WAVE_DOUBLES;   // this wave's corner for redistributing the plane parameters
  LMResidual2Loads r2_loads;
  const bool decide_here = st && pending == 1;
  if (wave == 0) {
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
  // Only the padding columns 6W .. NCOL of the two tile buffers have to start as zeros: phase A writes columns 0 .. 6W of every row of
  // a step (and the three spare columns of the z rows), the ragged step clears the rows it rounds up to, nothing else is read.
  // (Clearing both buffers whole -- 144 KB through a 128 B / clock LDS -- kept every wave 2.5k cycles from the barrier below.)
  {
    constexpr int PADC = C::NCOL - 6 * W;
    if constexpr (PADC > 0) {
      for (int k = tid; k < 2 * C::ROWS * PADC; k += K3_BLOCK) {
        const int b = k / (C::ROWS * PADC), rr = (k / PADC) % C::ROWS, cc = 6 * W + k % PADC;
        const int o = b * C::BUF + C::at(rr, cc);
        if (MIXED) reinterpret_cast<float*>(lds)[o] = 0.0f;
        else lds[o] = 0.0;
      }
    }
  }
  if (PAIR && tid < C::WAVES) reinterpret_cast<volatile int*>(lds + 2 * C::BUF + 12 * W)[tid] = 0;   // the pair flags (behind the poses)
  dbg_stamp(DBG, gw, 4);     // tiles cleared
  __syncthreads();
  dbg_stamp(DBG, gw, 1);
  bool undecided = false;
  if (st) {
    const bool in_done = lmv[0] != 0.0, in_calc = lmv[1] != 0.0;
    if (pending >= 2) {
      if (in_done) return;
    } else if (pending) {
      if (in_done) { if (blockIdx.x == nwg - 1) lm_carry(st, c_in, W); return; }
      undecided = true;
    } else {
      if (in_done || !in_calc) return;
    }
  }
  if (wave >= K3_FIRST_WAVES && wave < cnt && !(RAGF_ON && ragw)) { k3_load_clusters(pl, bs + wave, lane, e.c); k3_load_params<W>(pl, head, end, bs + wave, lane, stg); }
  if (wave == 0 && undecided) {
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
    if (blockIdx.x == nwg - 1) {
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
#if defined(K3_POSE_REGS) && K3_POSE_REGS
  // experiment: the lane's pose (a lane constant for the whole launch) held in 24 registers instead of six ds_read_b128 per step
  double pose[12];
#pragma unroll
  for (int k = 0; k < 12; k++) pose[k] = poseA[12 * fi + k];
#else
  const double* pose = poseA + 12 * fi;
#endif
  v4d acc[C::TPW];
#pragma unroll
  for (int t = 0; t < C::TPW; t++) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  v4f af[MIXED ? C::TPW : 1];   // mixed precision: the wave's f32 accumulators, widened into acc after the last step
#pragma unroll
  for (int t = 0; t < (MIXED ? C::TPW : 1); t++) af[t] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
  double dacc[DACC];
#pragma unroll
  for (int k = 0; k < DACC; k++) dacc[k] = 0.0;

  volatile int* pflag = reinterpret_cast<volatile int*>(poseA + 12 * W);   // (PAIR) per wave: steps whose rows are in LDS; zeroed below, before the first barrier that follows
  K3RowOfs ro = k3_row_offsets<W>(wave, vl, fi);   // lane constants: where the lane's three row pieces go inside a tile buffer
  double* const dump = lmv + 8 + C::WAVES * K3Stage<W>::WAVE_DOUBLES;   // 32 doubles behind the staging areas that nobody reads: where the idle lanes' (zero) rows go
  // (k3_lds_bytes reserves them.  A first version put the dump behind the poses: at W <= 2 that ran into the LM decision inputs -- found by the randomised sweep)
  if (!active) {
#pragma unroll
    for (int k = 0; k < 3; k++) { ro.rp[k] = 8 * k; ro.cp[k] = 2 * k; }
  }
  int dbg_step = -1;                                      // instrumented build: the step phase_a is running for
  // Phase A of the wave's batch b into tile buffer `bo`, then the requests for its next batch nb (nb < 0: none).
  // Measured and rejected (round 3, same box): the LDS round trip of the plane parameters issued two thirds of the way through phase M
  // instead of at the head of phase A -- 29.8 -> 29.6 us at cfg2, but 166.9 -> 173.3 us at cfg4 (13 steps per workgroup): the wait
  // for the parameter loads then sits inside the MFMA stream, and with more traffic in flight they have not always landed by then.
  auto unstage = [&](int b) __attribute__((always_inline)) { k3_unstage_params<W>(stg, stage_lds, head, end, b, active, vl, lane, e); };
  // `next`: 0 = no requests behind this phase A (compile-time at the call), 1 = request batch nb if `more` (wave-uniform, run time).
  // The run-time case issues the loads UNCONDITIONALLY through descriptors whose range check `more` switches off: with a branch around
  // them the entry registers became a phi of (old, loaded) values, and the copies that resolve it sat behind s_waitcnt vmcnt at the END
  // of phase A -- every wave waited out the latency of the loads it had just issued before it reached the barrier (round-4 find, from
  // the ISA: vmcnt(7) / (6) / (5) + six v_mov_b64 in front of the barrier, vmcnt(0) at the head of phase M).
  auto phase_a = [&](int b, int bo_in, int nb, bool more, auto next_tag) __attribute__((always_inline)) {
    const int bo = bo_in + (decltype(next_tag)::value == 1 ? rowshift : 0);      // full steps (next == 1): the pair layout; the ragged step: the plain one
    constexpr int next = decltype(next_tag)::value;
    // instrumented build, step 2 only: 7 parameters + pose back in registers, 14 rows computed, 15 rows stored, 18 = everything but the requests
    const bool stamp_here = DBG && dbg_step == 2;
    unstage(b);
    if (DBG && stamp_here) { asm volatile("" :: "v"(e.u[0]), "v"(e.coe)); __builtin_amdgcn_sched_barrier(0); dbg_stamp(true, gw, 7); __builtin_amdgcn_sched_barrier(0); }
    // voxel-level values for the spare columns, taken before phase A masks / consumes the entry
    const double spare_s = 1.4142135623730951 * e.sc;
    const double spare[3] = {spare_s * e.u[0], spare_s * e.u[1], spare_s * e.u[2]};
    if (active && C::SPARE && fi == W - 1) {   // one lane per voxel: columns 6W .. 6W+2 of the z row
      const int o = ro.rp[2] + C::at(0, 6 * W);
      if (MIXED) {
        float* zf = reinterpret_cast<float*>(lds) + bo + o;
        *reinterpret_cast<v2f*>(zf) = (v2f){(float)spare[0], (float)spare[1]};
        zf[2] = (float)spare[2];
      } else {
        double* zd = lds + bo + o;
        *reinterpret_cast<v2d*>(zd) = (v2d){spare[0], spare[1]};
        zd[2] = spare[2];
      }
    }
    // Rows go to the tile AS THEY ARE FINISHED (z row, G row 1, G row 2; round 4): stamps put ten 16-byte stores per wave, issued together
    // at the end of phase A by all four waves of a half at once, at 1.0-1.5k cycles until the last is performed -- which the wave then
    // waited out in front of the barrier.  With the first six under way while the second G row and the block-diagonal terms are still
    // being computed only the last three are young at the barrier.
    // (no branch around the stores of the 64 - NACT idle lanes: they write a dump area behind the poses instead -- a branch makes the wait-count
    // bookkeeping at its join wait for the stores just issued)
    double* const rowbase = active ? lds + bo : dump;
    float* const rowbase_f = active ? reinterpret_cast<float*>(lds) + bo : reinterpret_cast<float*>(dump);
    auto emit = [&](int r, const double row[6]) __attribute__((always_inline)) {
      if (MIXED) k3_store_row_f32(rowbase_f, ro, r, row);
      else k3_store_row(rowbase, ro, r, row);
#if !defined(K3_STORES_AT_END) || !K3_STORES_AT_END
      __builtin_amdgcn_sched_barrier(0);   // keep the stores where they are: the scheduler gathers them behind the last row otherwise
#endif
    };
    k3_phase_a<!C::SPARE>(e, fi, pose, dacc, emit);
    if (DBG && stamp_here) { __builtin_amdgcn_sched_barrier(0); dbg_stamp(true, gw, 14); __builtin_amdgcn_sched_barrier(0); }
    // next batch of this wave: in flight during the barrier and the whole of phase M.  Measured and rejected (round 2, same
    // box): the cluster rows requested a step earlier into a second register set (no change: the steps do not wait for loads);
    // the two waves of a SIMD taking phase M / phase A in opposite order (13.2k instead of 11.8k cycles per step).  Round 3: s_setprio
    // for one of a SIMD's two waves during phase A (either one: K3 26.6 -> 27.2 us), for phase M (no change).
    if (DBG && stamp_here) { __builtin_amdgcn_sched_barrier(0); dbg_stamp(true, gw, 15); asm volatile("" :: "v"(dacc[6]), "v"(dacc[11]), "v"(dacc[0])); __builtin_amdgcn_sched_barrier(0); dbg_stamp(true, gw, 18); }
    if constexpr (next == 2 || (next != 0 && !LATE)) {   // next == 2: requested here whatever the mode (the ragged batch taken first: its wave's first full batch)
      __builtin_amdgcn_sched_barrier(0);   // behind the last use of the entry: the loads go into the registers they free
#if defined(K3_PARAMS_FIRST) && K3_PARAMS_FIRST
      k3_load_params<W>(pl, head, end, nb, lane, stg, more);   // experiment: the values phase A needs first are requested first
      k3_load_clusters(pl, nb, lane, e.c, more);
#else
      k3_load_clusters(pl, nb, lane, e.c, more);
      k3_load_params<W>(pl, head, end, nb, lane, stg, more);
#endif
    }
  };

  // Full steps: step s = phase M of step s-1 (buffer (s-1)&1), phase A of step s (buffer s&1), one barrier.  The ragged last
  // step (cnt mod 8 batches) is peeled off below, so inside the loop every wave has a batch and both phases are straight-line code.
  const int nfull = cnt / C::WAVES, nrag = cnt - nfull * C::WAVES;
  const int k0_full = kq * C::KPW;
  // AF (experiment, -DK3_OPPOSITE=1): the second wave of every SIMD (w >= 4) takes phase A BEFORE phase M inside an iteration -- the
  // two touch different tile buffers, so the order is free -- to put one wave's VALU work under the other's MFMAs.
  auto ragged_m = [&](int bo) __attribute__((always_inline)) {
    // only ceil(nrag R / 4) K-steps exist; they are re-split over the K ranges
    const int ks = (nrag * C::R + 3) >> 2;
    const int k0 = (kq * ks) / C::KSPLIT, k1 = ((kq + 1) * ks) / C::KSPLIT;
    if (MIXED) k3_mfma_phase_f32<W, false>(reinterpret_cast<const float*>(lds) + bo, set, k0, k1 - k0, lrow, lcol, af);
    else k3_mfma_phase<W, false>(lds + bo, set, k0, k1 - k0, lrow, lcol, acc);
  };
  if constexpr (RAGF_ON) {
    if (ragf) {
      // rows of ragged batch r = wave - 4 at rows [R r, R (r + 1)) of buffer 1 (a shift by four row blocks of R rows: R is even, so the pair-interleaved
      // layout shifts linearly); behind it the wave's first full batch is requested.  Wave 0 clears the rows that round the step up to a whole K-step.
      if (ragw) phase_a(b_rag, C::BUF - 4 * C::R * C::NCOL, bs + wave, true, std::integral_constant<int, 2>{});
      else if (wave == 0) { double* z = lds + C::BUF + C::at(nrag * C::R, 0); for (int k = lane; k < 4 * C::NCOL; k += 64) z[k] = 0.0; }
    }
  }
  auto full_steps = [&](auto af_tag) __attribute__((always_inline)) -> bool {
    constexpr bool AF = decltype(af_tag)::value;
    for (int s = 0; s <= nfull; s++) {
      if constexpr (RAGF_ON) {
        if (ragf && s == 1) { ragged_m(C::BUF); __syncthreads(); }   // before phase A of step 1 overwrites buffer 1
      }
      auto phase_m = [&]() __attribute__((always_inline)) {
        const int bo = ((s - 1) & 1) * C::BUF;
        if (MIXED) {
          if constexpr (LATE) {   // mixed precision: the requests in front of the f32 products (not interleaved)
            const bool more_m = (s < nfull) || (wave < nrag && !ragf);
            const int nb_m = bs + s * C::WAVES + wave;
            k3_load_clusters(pl, nb_m, lane, e.c, more_m);
            k3_load_params<W>(pl, head, end, nb_m, lane, stg, more_m);
          }
          k3_mfma_phase_f32<W, true>(reinterpret_cast<const float*>(lds) + bo, set, k0_full, C::KPW, lrow, lcol, af);
        } else if constexpr (LATE) {
          // The requests for the batch of step s ride behind the K-steps of phase M of step s-1 (this iteration), one or two per K-step:
          // a vector-memory instruction costs the wave ~60 cycles of issue when eight waves queue on the CU's one address unit
          // (stamps: 460-540 cycles for the eight of a batch), and in front of the barrier that was on the step's critical path.
          const bool more_m = (s < nfull) || (wave < nrag && !ragf);
          const int nb_m = bs + s * C::WAVES + wave;
          constexpr int NQ = K3Stage<W>::Q + 1, NL = 5 + NQ;   // the parameters first: phase A starts with their LDS round trip
          constexpr int PER = (K3_LATE_PER * C::KPW >= NL) ? K3_LATE_PER : (NL + C::KPW - 1) / C::KPW;
          auto hook = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < PER; r++) {
              const int l = kk * PER + r;
              if (l < NQ) k3_load_param_q<W>(pl, head, end, nb_m, lane, stg, more_m, l);
              else if (l < NL) k3_load_cluster_row(pl, nb_m, lane, e.c, more_m, l - NQ);
            }
          };
          k3_mfma_phase<W, true>(lds + bo, set, k0_full, C::KPW, lrow, lcol, acc, hook);
        } else {
          k3_mfma_phase<W, true>(lds + bo, set, k0_full, C::KPW, lrow, lcol, acc);
        }
      };
      if (!AF && s >= 1) {
        phase_m();
        if (s <= 4) dbg_stamp(DBG, gw, 13 + 3 * s);   // phase M of step s-1 done: slots 16, 19, 22, 25
      }
      if (s < nfull) {
        const bool more = (s + 1 < nfull) || (wave < nrag && !ragf);
        if (DBG) dbg_step = s;
        phase_a(bs + s * C::WAVES + wave, (s & 1) * C::BUF, bs + (s + 1) * C::WAVES + wave, more, std::integral_constant<int, 1>{});
        if (s >= 1 && s <= 4) dbg_stamp(DBG, gw, 14 + 3 * s);   // phase A of step s done: slots 17, 20, 23, 26
      }
      if (AF && s >= 1) phase_m();
      if (s == nfull) break;
      if (PAIR && s >= 1 && !undecided) {
        // rows of step s are in LDS (the flag's store is queued behind them: one wave's LDS operations execute in order); then wait for
        // the partner's -- nobody else's rows are read in phase M of step s
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the compiler may not sink the row stores below the flag's)
        if (lane == 0) pflag[wave] = s + 1;
        while (pflag[wave ^ 4] < s + 1) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      } else {
        __syncthreads();
        if (PAIR && lane == 0) pflag[wave] = s + 1;     // keeps the counters in step when a later step switches to the flags
      }
      if (s < 6) dbg_stamp(DBG, gw, 8 + s);
      if (undecided && decide()) return true;
    }
    return false;
  };
#if defined(K3_OPPOSITE) && K3_OPPOSITE
  if ((wave >> 2) & 1 ? full_steps(std::true_type{}) : full_steps(std::false_type{})) return;
#else
  if (full_steps(std::false_type{})) return;
#endif
  if (nrag > 0 && !ragf) {
    // ragged last step: nrag < 8 batches.  Only ceil(nrag R / 4) K-steps exist; they are re-split over the K ranges, and the first
    // idle wave makes the rows that round the step up to a whole K-step read as zeros.
    const int bo = (nfull & 1) * C::BUF;
    if (wave < nrag) phase_a(bs + nfull * C::WAVES + wave, bo, 0, false, std::integral_constant<int, 0>{});
    else if (wave == nrag) {
      if (MIXED) { float* z = reinterpret_cast<float*>(lds) + bo + C::at(nrag * C::R, 0); for (int k = lane; k < 4 * C::NCOL; k += 64) z[k] = 0.0f; }
      else { double* z = lds + bo + C::at(nrag * C::R, 0); for (int k = lane; k < 4 * C::NCOL; k += 64) z[k] = 0.0; }
    }
    __syncthreads();
    if (nfull < 6) dbg_stamp(DBG, gw, 8 + nfull);
    if (undecided && decide()) return;
    const int ks = (nrag * C::R + 3) >> 2;
    const int k0 = (kq * ks) / C::KSPLIT, k1 = ((kq + 1) * ks) / C::KSPLIT;
    if (MIXED) k3_mfma_phase_f32<W, false>(reinterpret_cast<const float*>(lds) + bo, set, k0, k1 - k0, lrow, lcol, af);
    else k3_mfma_phase<W, false>(lds + bo, set, k0, k1 - k0, lrow, lcol, acc);
  }
  if (MIXED) {
#pragma unroll
    for (int t = 0; t < C::TPW; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[t][r] = (double)af[t][r];
  }
  dbg_stamp(DBG, gw, 3);

  // Deterministic in-block reduction through LDS (fixed order), one partial per workgroup:  [NTP tiles x 256 | W x DACC]
  using E = K3Epi<W>;
  constexpr int PLEN = C::NTP * 256 + W * DACC;
  double* pout = partial + (size_t)blockIdx.x * PLEN;
  double* park_d = lds;                                                   // [512 lanes][DS] linear accumulators
  double* park_t = E::ONE_PHASE ? lds + (size_t)K3_BLOCK * E::DS : lds;   // [8 waves][TPW][256] MFMA accumulators
  __syncthreads();  // every wave is done with the tiles
  dbg_stamp(DBG, gw, 27);
  if (undecided && decide()) return;   // a workgroup without a batch: first barrier since the prologue's
  // (1) per-frame linear accumulators: every lane parks the ones in use, then one thread per (frame, slot) sums the 8*NV lanes
#pragma unroll
  for (int k = 0; k < E::NUSED; k++) park_d[(wave * 64 + lane) * E::DS + k] = dacc[E::slot(k)];
  if (!E::ONE_PHASE) {
    __syncthreads();
    k3_sum_linear<W>(park_d, pout + C::NTP * 256, tid);
    __syncthreads();
  }
  // (2) MFMA accumulator tiles: wave (kq, set) parks its TPW tiles, then tile t of set s is the sum over the K ranges
#pragma unroll
  for (int j = 0; j < C::TPW; j++)
#pragma unroll
    for (int r = 0; r < 4; r++) park_t[(wave * C::TPW + j) * 256 + r * 64 + lane] = acc[j][r];
  dbg_stamp(DBG, gw, 28);
  __syncthreads();
  dbg_stamp(DBG, gw, 29);
  if (E::ONE_PHASE) k3_sum_linear<W>(park_d, pout + C::NTP * 256, tid);
  dbg_stamp(DBG, gw, 30);
  // two consecutive elements per thread: one 16-byte store (write-through like the other bulk outputs -- an 8-byte write-through store
  // costs 2.7x the time per byte of a 16-byte one, and this tail of 22 KB per workgroup is store-issue-bound)
  const __amdgpu_buffer_rsrc_t rout = k3_rsrc(pout);
  for (int el = 2 * tid; el < C::NTP * 256; el += 2 * K3_BLOCK) {
    const int t = el >> 8, x = el & 255;
    const int ts = t / C::TPW, j = t % C::TPW;
    v2d sum = (v2d){0.0, 0.0};
#pragma unroll
    for (int k = 0; k < C::KSPLIT; k++) sum += *reinterpret_cast<const v2d*>(park_t + ((PAIR ? (ts * C::KSPLIT + k) : (k * C::TSPLIT + ts)) * C::TPW + j) * 256 + x);
#if VXBA_WT_STORES
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, sum), rout, el * 8, 0, 16);   // aux 16 = sc1 on gfx950
#else
    *reinterpret_cast<v2d*>(pout + el) = sum;
#endif
  }
  dbg_stamp(DBG, gw, 6);
  if (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_stamp(true, gw, 31); }   // ... and acknowledged
}
