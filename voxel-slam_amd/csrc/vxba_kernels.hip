// Hand-written gfx950 (CDNA4, wave64) kernels of the LiDAR BA hot path.
//
//   K1  k1_build_kernel      per-(voxel,frame) cluster accumulation from bucketed points (LDS-staged)
//   K2  k2_residual_kernel   world merge + 3x3 covariance + Jacobi eigensolve + cache write + sum coe*lambda0
//   K3  k3_hessian_kernel    per-entry rank-3 rows (VALU) -> LDS -> f64 MFMA SYRK into the window Hessian,
//                            gradient / block-diagonal terms in registers, deterministic in-block reduction
//       k3_finalize_kernel   cross-workgroup reduction + packed [Hess | JacT | residual] assembly
//   K4  k4_plane_fit_kernel  batched eig(cluster.cov())
//
// Reference loops these replace: tools.hpp:326-331 (K1), voxel_map.hpp:243-279 (K2), voxel_map.hpp:132-241 (K3),
// voxel_map.hpp:1161-1163 / loop_refine.hpp:363-366 (K4), voxel_map.hpp:323-332 (reduction).
// Mathematics: vxba_math.hpp / SURVEY.md Appendix A.  No float atomics anywhere: every reduction is a fixed
// tree, so results are run-to-run bitwise reproducible for a given launch geometry.
#include "vxba_kernels.h"

#include <hip/hip_ext.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

#include "vxba_math.hpp"
#include "vxba_solve.hpp"

namespace vxk {

typedef double v4d __attribute__((ext_vector_type(4)));

// Development instrumentation: per-wave s_memtime stamps of the debug kernel instantiations (VXBA_DBG=1).
constexpr int DBG_SLOTS = 32;
constexpr int DBG_WAVES = 4096;
__device__ unsigned long long g_dbg[DBG_WAVES * DBG_SLOTS];
__device__ __forceinline__ void dbg_stamp(bool on, int wave_id, int slot) {
  if (on && wave_id < DBG_WAVES && (threadIdx.x & 63) == 0) g_dbg[wave_id * DBG_SLOTS + slot] = __builtin_readcyclecounter();
}

#include "vxba_solve4.hpp"

// (-DVXBA_WT_STORES=0 restores plain stores; same-box A/B at cfg2: 60.0 -> 59.0 us per LM step.)  The bulk outputs of the sweeps (K2's cache planes, K3's workgroup partials) as agent-scope
// write-through stores, so that the kernel boundary behind them has no dirty L2 lines to write back.
#ifndef VXBA_WT_STORES
#define VXBA_WT_STORES 1
#endif
__device__ __forceinline__ void st_out(double* p, double v) {
#if VXBA_WT_STORES
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}


// ------------------------------------------------------------------------------------------------
// Accept / reject + damping schedule of one LM step (voxel_map.hpp:411-439) as a pure function of the control
// block and residual2.  Called by the stand-alone decision kernel and by the prologue of every Hessian-sweep
// workgroup (all compute the same bits from the same inputs).
// ------------------------------------------------------------------------------------------------
struct LMDecision {
  int accept;
  double u, v, r2, q;
  int calc_hess, done, converge, rejected, iter, n_accept, n_reject;
};
// (CtlT: LMCtl itself, or the scalars of it a caller fetched ahead of time -- LMCtlScalars)
struct LMCtlScalars { double u, v, residual1, q1; int converge, n_accept, n_reject, iter, done, bench_mode; };
template <class CtlT>
__device__ __forceinline__ LMDecision lm_decide(const CtlT& in, double r2, int restart) {
  LMDecision d;
  const double r1 = in.residual1, q1 = in.q1;
  d.r2 = r2;
  d.q = r1 - r2;
  d.accept = d.q > 0;
  d.converge = in.converge;
  d.n_accept = in.n_accept;
  d.n_reject = in.n_reject;
  if (d.accept) {
    const double one_three = 1.0 / 3;
    const double rho = d.q / q1;
    const double t = 2 * rho - 1;
    const double g = 1 - t * t * t;   // pow(t, 3) upstream; a plain cube keeps the decision off the slow path (differs by <= 1 ulp in u)
    d.v = 2;
    d.u = in.u * (g < one_three ? one_three : g);
    d.calc_hess = 1;
    d.rejected = 0;
    d.n_accept += 1;
  } else {
    d.u = in.u * in.v;
    d.v = 2 * in.v;
    d.calc_hess = 0;
    d.converge = 0;
    d.rejected = 1;
    d.n_reject += 1;
  }
  d.iter = in.iter + 1;
  d.done = in.done;
  if (!in.bench_mode && fabs((r1 - r2) / r1) < 1e-6) d.done = 1;
  if (restart) { d.u = 0.01; d.v = 2.0; d.calc_hess = 1; d.rejected = 0; }
  return d;
}
// Deterministic residual2: the all-reduced scalar if supplied, else the sum of the residual sweep's wave partials.
// Every wave of every caller adds the same values in the same order (16 independent loads per lane and chunk, a
// fixed in-lane tree, an xor butterfly across the wave): no LDS, no barrier, identical bits everywhere.
__device__ __forceinline__ double lm_residual2(const LMPending& pend) {
  if (pend.d_scalar) return pend.d_scalar[0];
  const int lane = threadIdx.x & 63;
  const double* __restrict__ p = pend.partial;
  const int n = pend.nparts;
  double total = 0.0;
  for (int base = 0; base < n; base += 1024) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = base + 64 * k + lane;
      v[k] = i < n ? p[i] : 0.0;
    }
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
      for (int k = 0; k < w; k++) v[k] += v[k + w];
    total += v[0];
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) total += __shfl_xor(total, m, 64);
  return total;
}
// The same sum in two halves, for a caller that wants the loads in flight long before it needs the value (the Hessian sweep's prologue):
// lm_residual2_issue requests the all-reduced scalar or the first 1024 partials, lm_residual2_finish adds them up -- and any further
// chunks -- in exactly the order lm_residual2 uses, so both return the same bits.
struct LMResidual2Loads { double v[16]; };
__device__ __forceinline__ void lm_residual2_issue(const LMPending& pend, LMResidual2Loads& L) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 16; k++) L.v[k] = 0.0;
  if (pend.d_scalar) { L.v[0] = pend.d_scalar[0]; return; }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int i = 64 * k + lane;
    if (i < pend.nparts) L.v[k] = pend.partial[i];
  }
}
__device__ __forceinline__ double lm_residual2_finish(const LMPending& pend, LMResidual2Loads& L) {
  if (pend.d_scalar) return L.v[0];
  const int lane = threadIdx.x & 63;
  const double* __restrict__ p = pend.partial;
  const int n = pend.nparts;
#pragma unroll
  for (int w = 8; w > 0; w >>= 1)
#pragma unroll
    for (int k = 0; k < w; k++) L.v[k] += L.v[k + w];
  double total = 0.0;
  total += L.v[0];
  for (int base = 1024; base < n; base += 1024) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = base + 64 * k + lane;
      v[k] = i < n ? p[i] : 0.0;
    }
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
      for (int k = 0; k < w; k++) v[k] += v[k + w];
    total += v[0];
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) total += __shfl_xor(total, m, 64);
  return total;
}
// Loop already left: the control block moves to the other slot unchanged (one workgroup).
__device__ __forceinline__ void lm_carry(LMState* st, int c_in, int W) {
  LMCtl& out = st->ctl[c_in ^ 1];
  const LMCtl& in = st->ctl[c_in];
  const int tid = threadIdx.x;
  if (tid < 12 * W) { out.x[tid] = in.x[tid]; out.xt[tid] = in.xt[tid]; }
  if (tid == 0) {
    out.u = in.u; out.v = in.v; out.residual1 = in.residual1; out.residual2 = in.residual2; out.q1 = in.q1;
    out.resis[0] = in.resis[0]; out.resis[1] = in.resis[1];
    out.calc_hess = in.calc_hess; out.done = in.done; out.iter = in.iter; out.converge = in.converge; out.rejected = in.rejected;
    out.bench_mode = in.bench_mode; out.n_accept = in.n_accept; out.n_reject = in.n_reject;
  }
}
// ... by ONE wave (the solve workgroup of a fused residual + Hessian launch, when the loop was left before it)
__device__ __forceinline__ void lm_carry_wave(LMState* st, int c_in, int W) {
  LMCtl& out = st->ctl[c_in ^ 1];
  const LMCtl& in = st->ctl[c_in];
  const int lane = threadIdx.x & 63;
  for (int k = lane; k < 12 * W; k += 64) { out.x[k] = in.x[k]; out.xt[k] = in.xt[k]; }
  if (lane == 0) {
    out.u = in.u; out.v = in.v; out.residual1 = in.residual1; out.residual2 = in.residual2; out.q1 = in.q1;
    out.resis[0] = in.resis[0]; out.resis[1] = in.resis[1];
    out.calc_hess = in.calc_hess; out.done = in.done; out.iter = in.iter; out.converge = in.converge; out.rejected = in.rejected;
    out.bench_mode = in.bench_mode; out.n_accept = in.n_accept; out.n_reject = in.n_reject;
  }
}
// Persist the decided control block (one workgroup).  Poses: restart ? x0 : accept ? xt : x.
__device__ __forceinline__ void lm_persist(LMState* st, int c_in, const LMDecision& d, int restart, const PoseArg& x0, int W) {
  const LMCtl& in = st->ctl[c_in];
  LMCtl& out = st->ctl[c_in ^ 1];
  const int tid = threadIdx.x;
  if (tid < 12 * W) {
    const double xn = restart ? x0.Rp[tid] : (d.accept ? in.xt[tid] : in.x[tid]);
    out.x[tid] = xn;
    out.xt[tid] = restart ? x0.Rp[tid] : in.xt[tid];
  }
  if (tid == 0) {
    const int it = in.iter;
    double* tr = st->trace + 8 * (it < LM_MAX_ITER ? it : LM_MAX_ITER - 1);
    tr[0] = in.residual1; tr[1] = d.r2; tr[2] = in.u; tr[3] = in.v; tr[4] = d.q; tr[5] = in.q1; tr[6] = d.accept ? 1.0 : 0.0; tr[7] = in.calc_hess;
    out.u = d.u; out.v = d.v;
    out.residual1 = in.residual1; out.residual2 = d.r2; out.q1 = in.q1;
    out.resis[0] = in.resis[0]; out.resis[1] = d.r2;
    out.calc_hess = d.calc_hess; out.done = d.done; out.iter = d.iter; out.converge = d.converge; out.rejected = d.rejected;
    out.bench_mode = in.bench_mode; out.n_accept = d.n_accept; out.n_reject = d.n_reject;
  }
}

// lm_persist by ONE wave, restart = 0 (the solve workgroup of a fused residual + Hessian launch takes the decision itself)
__device__ __forceinline__ void lm_persist_wave(LMState* st, int c_in, const LMDecision& d, int W) {
  const LMCtl& in = st->ctl[c_in];
  LMCtl& out = st->ctl[c_in ^ 1];
  const int lane = threadIdx.x & 63;
  for (int k = lane; k < 12 * W; k += 64) {
    out.x[k] = d.accept ? in.xt[k] : in.x[k];
    out.xt[k] = in.xt[k];
  }
  if (lane == 0) {
    const int it = in.iter;
    double* tr = st->trace + 8 * (it < LM_MAX_ITER ? it : LM_MAX_ITER - 1);
    tr[0] = in.residual1; tr[1] = d.r2; tr[2] = in.u; tr[3] = in.v; tr[4] = d.q; tr[5] = in.q1; tr[6] = d.accept ? 1.0 : 0.0; tr[7] = in.calc_hess;
    out.u = d.u; out.v = d.v;
    out.residual1 = in.residual1; out.residual2 = d.r2; out.q1 = in.q1;
    out.resis[0] = in.resis[0]; out.resis[1] = d.r2;
    out.calc_hess = d.calc_hess; out.done = d.done; out.iter = d.iter; out.converge = d.converge; out.rejected = d.rejected;
    out.bench_mode = in.bench_mode; out.n_accept = d.n_accept; out.n_reject = d.n_reject;
  }
}

// ------------------------------------------------------------------------------------------------
// K3 -- Hessian / gradient sweep: vxba_k3.hpp (work mapping, tile layout and the history of the design are described there).
// H = -S + blockdiag(D) is assembled by k3_finalize_kernel below.
// ------------------------------------------------------------------------------------------------
#include "vxba_k3.hpp"

// Cross-workgroup reduction + assembly.  One lane per PARTIAL element (consecutive lanes -> consecutive
// addresses inside every workgroup partial: coalesced), 64 elements x 16 partial-slices per workgroup, fixed
// summation order.  An MFMA tile element (I,J,row,col) becomes Hess(r,c) = -S(r,c) [+ D term when r and c belong
// to the same frame] and is mirrored to Hess(c,r) (voxel_map.hpp:237-239); the linear elements become JacT and
// the residual.  Output buffer: Hess (6W)^2 column-major | JacT 6W | residual.
__device__ __forceinline__ int sym6_index(int a, int b) { return a == 0 ? b : (a == 1 ? 2 + b : 5); }  // a <= b

// Where element e of a workgroup partial goes: a tile element -> Hessian entry (r, c) [+ a second stream off1: the block-diagonal D
// element that lands on the same entry], a linear element -> output index lin (JacT / residual), or nothing.
struct FinMap { int r, c, lin, off1; };
// (I, J) of pair `pl` of wave `wv`: the tables of vxba_k3.hpp as device constants (fin_map indexes them at run time)
template <int NG> __device__ const K3PairTab k3_pair_tab_d = k3_make_pairs(NG);
template <int W>
__device__ __forceinline__ FinMap fin_map(int e) {
  using C = K3Cfg<W>;
  constexpr int n = 6 * W;
  constexpr int NTILE = C::NTILE;
  constexpr int PLEN = C::PLEN;
  FinMap m;
  m.r = -1; m.c = -1; m.lin = -1; m.off1 = -1;
  if (e < NTILE) {
    // S block (I, J) of a wave's pair pl, element (i, j): [wave][pair][4 i + j] (vxba_k3.hpp, epilogue)
    const int wv = e / (16 * C::PPWP), pl = (e >> 4) % C::PPWP, i = (e >> 2) & 3, j = e & 3;
    if (pl < C::npair(wv)) {
      const K3PairTab& T = k3_pair_tab_d<C::NG>;
      const int I = T.I[wv][pl], J = T.J[wv][pl];
      int r = 4 * I + i, c = 4 * J + j;
      if (r >= n || c >= n || r > c) { r = -1; c = -1; }   // padding columns (odd W); the lower half of a diagonal block is a duplicate
      else if (r / 6 == c / 6) {
        const int fr = r / 6, a = r % 6, bb = c % 6;       // a <= bb: the block-diagonal term that lands on the same entry
        int d;
        if (bb < 3) d = 6 + sym6_index(a, bb);
        else if (a < 3) d = 12 + 3 * a + (bb - 3);
        else d = 21 + sym6_index(a - 3, bb - 3);
        m.off1 = NTILE + fr * DACC + d;
      }
      m.r = r; m.c = c;
    }
  } else if (e < PLEN) {
    const int q = e - NTILE, i = q / DACC, d = q % DACC;
    if (d < 6) m.lin = n * n + 6 * i + d;
    else if (d == 27 && i == 0) m.lin = n * n + n;
  }
  return m;
}
// The reduced element (t0: its own stream, t1: the D stream) into the packed buffer [Hess (6W)^2 column-major | JacT 6W | residual] and,
// with write_state, into the LM state the solve reads.  COH: the state goes out as written-through agent-scope stores (the solve runs in
// the same launch, on another workgroup -- k2_residual_kernel).
template <int W, bool COH>
__device__ __forceinline__ void fin_emit(const FinMap& m, double t0, double t1, LMState* __restrict__ gate, int cb, int write_state, double* __restrict__ packed, int iter) {
  constexpr int n = 6 * W;
  auto put = [](double* p, double v) __attribute__((always_inline)) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
  };
  if (m.lin >= 0) {
    packed[m.lin] = t0;
    if (gate && write_state) {   // LM state: gauge-fixed gradient (voxel_map.hpp:400), residual1 (:388)
      if (m.lin < n * n + n) put(&gate->Jwork[m.lin - n * n], (m.lin - n * n < 6) ? 0.0 : t0);
      else { gate->ctl[cb].residual1 = t0; if (iter == 0) gate->ctl[cb].resis[0] = t0; }   // `iter` is the caller's early load: read here, behind the stores above, it cost this thread a load + the stores' acknowledgement at the very end of the kernel
    }
  } else {
    const int r = m.r, c = m.c;
    const double h = t1 - t0;
    packed[(size_t)c * n + r] = h;
    packed[(size_t)r * n + c] = h;
    if (gate && write_state) {   // LM state: *hess = Hess before the gauge fix (:391) and the gauge-fixed working copy (:397-400)
      gate->hess_out[(size_t)c * n + r] = h;
      gate->hess_out[(size_t)r * n + c] = h;
      const double hw = (r < 6 || c < 6) ? ((r == c) ? 1.0 : 0.0) : h;
      put(&gate->Hwork[(size_t)c * n + r], hw);
      put(&gate->Hwork[(size_t)r * n + c], hw);
    }
  }
}

#ifndef FIN_EL_
#define FIN_EL_ 16
#endif
constexpr int FIN_EL = FIN_EL_, FIN_SL = 64;   // FIN_EL * FIN_SL threads (<= 1024)
// reset_slots (nullable): n_reset doubles set to NaN by the last workgroup -- the residual slots of the NEXT fused residual + Hessian launch
// (vxba_k23.hpp: its solve workgroup waits for every sweep workgroup's sum to replace the NaN; this kernel runs between any two such launches).
template <int W, bool DBG = false>
__global__ __launch_bounds__(FIN_EL * FIN_SL) void k3_finalize_kernel(const double* __restrict__ partial, int nblocks, LMState* __restrict__ gate, int cb,
                                                           int write_state, double* __restrict__ packed, int force, const double* __restrict__ k2_partial,
                                                           int k2_nparts, double* __restrict__ reset_slots, int n_reset) {
  using C = K3Cfg<W>;
#ifndef VXBA_K23_DBG_NORESET
  if (reset_slots && blockIdx.x == gridDim.x - 1)
    for (int k = threadIdx.x; k < n_reset; k += FIN_EL * FIN_SL) reset_slots[k] = __builtin_nan("");
#endif
  const int dbg_w = 3000 + (int)blockIdx.x;            // instrumented build: stamps of wave 0 of every workgroup (rows 3000.. of the stamp table)
  dbg_stamp(DBG && threadIdx.x < 64, dbg_w, 0);
  // LM flags: requested now (vector loads: lane-dependent zero offset), tested after the partials are in flight
  const int zoff = threadIdx.x >> 30;
  int f_done = gate ? (&gate->ctl[cb].done)[zoff] : 0;
  int f_calc = gate ? (&gate->ctl[cb].calc_hess)[zoff] : 1;
  const int f_iter = gate ? (&gate->ctl[cb].iter)[zoff] : 1;
  constexpr int n = 6 * W;
  constexpr int PLEN = C::PLEN;
  // FIN_EL elements x FIN_SL slices of the workgroup partials per block: many small blocks, because one CU cannot pull
  // more than ~10 B/clk from L2/HBM -- 64 elements per block (45 blocks) left the reduction bound by 45 CUs' load paths
  __shared__ double red0[FIN_SL][FIN_EL];
  __shared__ double red1[FIN_SL][FIN_EL];
  __shared__ double mid0[8][FIN_EL];
  __shared__ double mid1[8][FIN_EL];
  const int el = threadIdx.x % FIN_EL, slice = threadIdx.x / FIN_EL;
  const int e = blockIdx.x * FIN_EL + el;
  const FinMap m = fin_map<W>(e);
  const int off1 = m.off1;
  const bool need0 = (m.r >= 0) || (m.lin >= 0);
  double s0 = 0.0, s1 = 0.0;
  bool flags_checked = false;
  if (need0) {
    // the loads of 4 partials are issued together (independent), added in fixed order
    int b = slice;
    for (; b + FIN_SL * 3 < nblocks; b += FIN_SL * 4) {
      double v0[4], v1[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double* pb = partial + (size_t)(b + FIN_SL * q) * PLEN;
        v0[q] = pb[e];
        v1[q] = off1 >= 0 ? pb[off1] : 0.0;
      }
      if (!flags_checked) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(f_done), "+v"(f_calc));
        flags_checked = true;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) { s0 += v0[q]; s1 += v1[q]; }
    }
    for (; b < nblocks; b += FIN_SL) {
      const double* pb = partial + (size_t)b * PLEN;
      s0 += pb[e];
      if (off1 >= 0) s1 += pb[off1];
    }
  }
  if (!flags_checked) asm volatile("" : "+v"(f_done), "+v"(f_calc));
  if (DBG) { asm volatile("" :: "v"(s0), "v"(s1)); dbg_stamp(threadIdx.x < 64, dbg_w, 1); }   // loads landed, summed
  if (f_done || (!f_calc && !force)) return;   // uniform over the grid
  // sharded speculative loop: the residual of the trial state (the residual sweep's wave partials) rides in the slot behind the
  // packed buffer, so that ONE all-reduce carries the system and the number the accept/reject test needs
  if (k2_partial && blockIdx.x == gridDim.x - 1) {
    double sum = 0.0;
    for (int k = threadIdx.x; k < k2_nparts; k += FIN_EL * FIN_SL) sum += k2_partial[k];
    double* red = &red0[0][0];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int m = FIN_EL * FIN_SL; m > 1; m = (m + 1) >> 1) {   // any block size: the upper half folds onto the lower
      const int half = (m + 1) >> 1;
      if ((int)threadIdx.x < m - half) red[threadIdx.x] += red[threadIdx.x + half];
      __syncthreads();
    }
    if (threadIdx.x == 0) packed[n * n + n + 1] = red[0];
    __syncthreads();
  }
  red0[slice][el] = s0;
  red1[slice][el] = s1;
  __syncthreads();
  if (slice < 8) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int k = 0; k < FIN_SL / 8; k++) { t0 += red0[slice * (FIN_SL / 8) + k][el]; t1 += red1[slice * (FIN_SL / 8) + k][el]; }
    mid0[slice][el] = t0;
    mid1[slice][el] = t1;
  }
  __syncthreads();
  if (slice == 0 && need0) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; k++) { t0 += mid0[k][el]; t1 += mid1[k][el]; }
    fin_emit<W, false>(m, t0, t1, gate, cb, write_state, packed, f_iter);
  }
  dbg_stamp(DBG && threadIdx.x < 64, dbg_w, 2);        // reduced through LDS, outputs issued
  if (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_stamp(threadIdx.x < 64, dbg_w, 3); }
}

// ------------------------------------------------------------------------------------------------
// K2 -- residual sweep.  One lane per voxel, frames unrolled: every load is a 512 B contiguous row of a
// frame-major plane, poses are wave-uniform (scalar loads), no cross-lane traffic until the final residual
// reduction; the eigensolver is warm-started from the cached eigenvectors.
// Tried and rejected (measured on MI355X, cfg2): splitting a voxel's frames over 4 lanes (quad shuffles) or over
// the 4 waves of a 256-thread workgroup (LDS merge, redundant eigensolve) to get ~3 waves per SIMD -- both ran
// 27-29 us against 17.5 us for this version: the extra pose loads / redundant fp64 work / barrier skew cost more
// than the hidden dependency stalls buy, because all workgroups start together and sit in the same phase.
// ------------------------------------------------------------------------------------------------
// LM mode with seq != 0: workgroup 0 of the grid is not a voxel workgroup but the damped SOLVE of this iteration
// (lm_solve_body4, vxba_solve4.hpp: four waves).  The voxel workgroups request their cluster rows first -- those do not depend on
// the poses -- and only then wait for workgroup 0 to publish the trial poses (relaxed store of `seq` to st->solve_seq, relaxed
// polls here), so the load phase of the sweep and one kernel launch disappear behind the solve.  Deadlock-free as long as
// workgroup 0 is resident before any workgroup that waits for it (dispatch in index order; the wait is bounded, see below).
// Workgroups are 4 waves (the solve wants four SIMDs); a voxel wave is as independent as it was when it was a workgroup of its
// own: wave v of the launch takes voxels [head + 64 v, ...), no barrier, its own corner of LDS for the poses.
constexpr unsigned K2_SPIN_LIMIT = 1u << 22;
constexpr int K2_WAVES = S4_WAVES;
constexpr int K2_THREADS = 64 * K2_WAVES;
#ifndef K2_HEAD_START
#define K2_HEAD_START 100   // x 64 cycles
#endif
#ifndef VXBA_PUBLISH_FENCE
#define VXBA_PUBLISH_FENCE 0
#endif
template <int W>
constexpr int k2_lds_doubles() { return S4<W>::DOUBLES > K2_WAVES * 12 * W ? S4<W>::DOUBLES : K2_WAVES * 12 * W; }
// F32: the cluster rows come from the f32 re-centred copy (fv.cl32, vxm::cluster_to_centred_f32; VXBA_OPT_F32_CLUSTERS, meant for the
// mixed-precision configuration): half the bytes of the sweep's dominant stream and half the registers of the load phase.
template <int W, bool DBG = false, bool F32 = false>
// Argument order: the first 14 dwords are preloaded into SGPRs at wave launch (see k3_hessian_kernel) -- what the solve workgroup needs to
// start (it is the critical path of a fused launch) and what a voxel wave needs for its done-check and its place in the sweep.
__global__ __launch_bounds__(K2_THREADS) void k2_residual_kernel(LMState* __restrict__ st, int c, unsigned seq, const double* li_rec, double* li_out,
                                                                 const double* host_feed, int head, int end, int VPB_arg, int head_start,
                                                                 double* __restrict__ partial, FactorView fv, PoseArg poses) {
  const int VPB = VPB_arg & 0xffff;
  __shared__ __attribute__((aligned(16))) double k2_lds[k2_lds_doubles<W>()];
  // LM mode: trial poses of ctl[c]; nothing to do once the loop is done
  // (the solve workgroup of a fused launch tests the flag itself, behind its loads: lm_solve_body4)
  if (st && !(seq != 0 && blockIdx.x == 0 && !host_feed) && st->ctl[c].done) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double* pose_lds = k2_lds + wave * 12 * W;
  int vb = blockIdx.x * K2_WAVES + wave;
  if (DBG) { if (blockIdx.x == 0) dbg_stamp(wave == 0, 4000, 30); else dbg_stamp(true, vb - K2_WAVES, 6); }   // kernel entry
  if (st && seq != 0) {
    if (blockIdx.x == 0) {
      if (host_feed) {
        // LiDAR-inertial shells: the trial poses come from the HOST (15W-dimensional solve there).  This launch was queued before
        // they existed -- the voxel workgroups already hold their cluster rows -- and wave 0 of workgroup 0 waits for the host to raise
        // the sequence number in mapped host memory, copies the poses into the control block and publishes them like a solve would.
        if (wave != 0) return;
        const volatile double* hf = host_feed;
        unsigned spins = 0;
        bool fed = true;
        while ((unsigned)hf[0] != seq) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > K2_SPIN_LIMIT) { fed = false; break; }    // the host went away: the voxel workgroups give up too, partials stay as the host left them
        }
        if (!fed) { if (lane == 0) st->error = 2; return; }
        for (int k = lane; k < 12 * W; k += 64) __hip_atomic_store(&st->ctl[c].xt[k], hf[1 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        lm_solve_body4<W, DBG>(st, c, k2_lds, li_rec, li_out, seq);    // li_rec: the LiDAR-inertial shells' reduced pose system (vxba_solve4.hpp)
        if (wave != 0) return;
      }
#if VXBA_PUBLISH_FENCE
      __threadfence();
      if (lane == 0) __hip_atomic_store(&st->solve_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
      // the trial poses went out as written-through agent-scope stores: wait for their acknowledgement, then publish -- no L2
      // write-back + invalidate on the critical path (everything else the solve wrote is only read by later kernels)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) __hip_atomic_store(&st->solve_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      return;
    }
    vb -= K2_WAVES;
  }
  const bool fused = st && seq != 0;
  const bool spare = vb * VPB >= end - head;   // the last workgroup's spare waves: no voxels; in a fused launch they stay for the workgroup's barrier
  if (spare && !fused) return;
  if (fused) {
    // let the solver's first (dependent) loads through before 780 waves put 50 MB of requests in front of them
    for (int k = 0; k < head_start; k += 10) __builtin_amdgcn_s_sleep(10);
  }
  const int a = head + vb * VPB + lane;
  const bool valid = !spare && lane < VPB && a < end;
  const size_t VS = (size_t)fv.VS;
  double res = 0.0;
  dbg_stamp(DBG, vb, 0);
  // issue every load of this voxel up front (10 + 10 W independent 512 B rows per wave): with < 1 wave per
  // SIMD at 50k voxels the sweep is latency-bound unless all of them are in flight together
  double fx[10], cl[F32 ? 1 : W][10], Up[9];
  float cr[F32 ? W : 1][10];
  // coe is requested HERE, with everything else: read where it is used -- behind the cache stores at the end -- the load could not be
  // moved above those stores (may alias, for all the compiler knows), and the wave waited for its latency AND for the acknowledgement of
  // the 22 written-through stores in front of it before its last four stores and the reduction (round 4, from the ISA: a
  // global_load_dwordx2 + s_waitcnt vmcnt(0) between the cache stores)
  double coe_v = 0.0;
  if (valid) {
    coe_v = fv.coe[a];
#pragma unroll
    for (int k = 0; k < 10; k++) fx[k] = fv.fix[k * VS + a];
    // previous eigenvectors (plane 3*col+row -> row-major): warm start of the eigensolver
#pragma unroll
    for (int col = 0; col < 3; col++)
#pragma unroll
      for (int row = 0; row < 3; row++) Up[3 * row + col] = fv.eigvec[(size_t)(3 * col + row) * VS + a];
    if (F32) {
#pragma unroll
      for (int i = 0; i < W; i++)
#pragma unroll
        for (int k = 0; k < 10; k++) cr[F32 ? i : 0][k] = fv.cl32[((size_t)i * 10 + k) * VS + a];
    } else {
#pragma unroll
      for (int i = 0; i < W; i++)
#pragma unroll
        for (int k = 0; k < 10; k++) cl[F32 ? 0 : i][k] = fv.cl[((size_t)i * 10 + k) * VS + a];
    }
  }
  // poses -> LDS (wave-uniform operands of the transform).  LM mode reads the trial poses with coherent loads, after
  // the solve has published them when it runs inside this launch.
  {
    if (fused) {
      // ONE wave per workgroup polls (196 pollers on one L2 line instead of 782, so they can poll four times as often) and fetches the
      // trial poses for all four; the others sleep in the workgroup barrier.  Relaxed polls and no acquire fence (either would
      // invalidate caches chip-wide, hundreds of times over): the poses are fetched with system-coherent (volatile) loads, issued in
      // program order after the poll that saw `seq`.
      __shared__ int k2_gave_up;
      pose_lds = k2_lds;
      if (wave == 0) {
        unsigned spins = 0;
        const unsigned spin_limit = (VPB_arg >> 16) ? 1u : 4u * K2_SPIN_LIMIT;   // bit 16 of VPB_arg: test hook (VXBA_OPT_DEBUG_SOLVE_TIMEOUT), give up at once
        bool seen = true;
        while (__hip_atomic_load(&st->solve_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > spin_limit) { seen = false; break; }   // never observed in production; a hang would cost the GPU
        }
        if (DBG) dbg_stamp(true, vb, 5);
        if (seen) {
          const volatile double* xt = st->ctl[c].xt;
          if (lane < 12 * W) pose_lds[lane] = xt[lane];
          if (lane + 64 < 12 * W) pose_lds[lane + 64] = xt[lane + 64];
        } else if (lane == 0) st->error = 1;
        if (lane == 0) k2_gave_up = seen ? 0 : 1;
      }
      __syncthreads();
      if (k2_gave_up || spare) return;
      // (Round 5, measured and rejected: pulling the FIRST STEP of the Hessian sweep that follows into the XCD-local L2 from here -- one dword
      // per 128-byte line of the 40 KB every sweep workgroup asks for first, by the workgroups of this launch that run on the same XCD.  Same
      // box: the sweep 23.6 -> 23.7 us, this launch 21.5 -> 22.7: the sweep's fill does not wait for those lines, gpurun_out r5_s3.)
    } else if (st) {
      const volatile double* xt = st->ctl[c].xt;
      if (lane < 12 * W) pose_lds[lane] = xt[lane];
      if (lane + 64 < 12 * W) pose_lds[lane + 64] = xt[lane + 64];
    } else {
      if (lane < 12 * W) pose_lds[lane] = poses.Rp[lane];
      if (lane + 64 < 12 * W) pose_lds[lane + 64] = poses.Rp[lane + 64];
    }
    __builtin_amdgcn_wave_barrier();
  }
  double SP[6], Sv[3], SN = 1.0, C[6], lam[3] = {0.0, 0.0, 0.0}, U[9];
  if (valid) {
    if (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_stamp(true, vb, 1); }
#pragma unroll
    for (int k = 0; k < 6; k++) SP[k] = fx[k];
#pragma unroll
    for (int k = 0; k < 3; k++) Sv[k] = fx[6 + k];
    SN = fx[9];
#pragma unroll
    for (int i = 0; i < W; i++) {
      double R[9], p[3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) R[3 * r + cc] = pose_lds[12 * i + 3 * cc + r];
#pragma unroll
      for (int k = 0; k < 3; k++) p[k] = pose_lds[12 * i + 9 + k];
      if (F32) vxm::transform_accumulate_centred(cr[F32 ? i : 0], R, p, SP, Sv, SN);
      else {
        // N == 0 <=> frame i did not observe this voxel (voxel_map.hpp:258): contributes nothing
        double* ci = cl[F32 ? 0 : i];
        const bool obs = ci[9] != 0.0;
#pragma unroll
        for (int k = 0; k < 10; k++) ci[k] = obs ? ci[k] : 0.0;
        vxm::transform_accumulate(ci, ci + 6, ci[9], R, p, SP, Sv, SN);
      }
    }
    vxm::cluster_cov(SP, Sv, SN, C);
    if (DBG) { asm volatile("" :: "v"(C[0]), "v"(C[3]), "v"(C[5])); dbg_stamp(true, vb, 2); }
    vxm::eig_sym3_warm(C, Up, lam, U);
    if (DBG) { asm volatile("" :: "v"(lam[0]), "v"(U[0]), "v"(U[8])); dbg_stamp(true, vb, 3); }
    res = coe_v * lam[0];
  }
  // fixed-tree wave reduction, and the workgroup's partial out BEFORE the cache stores: behind them (a join after 26 written-through stores)
  // the partial waited for their acknowledgement
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) res += __shfl_down(res, off);
  if (lane == 0) partial[vb] = res;   // (written through like the cache planes: measured, no change)
  if (valid) {
#pragma unroll
    for (int k = 0; k < 3; k++) st_out(&fv.eigval[k * VS + a], lam[k]);
#pragma unroll
    for (int col = 0; col < 3; col++)
#pragma unroll
      for (int row = 0; row < 3; row++) st_out(&fv.eigvec[(3 * col + row) * VS + a], U[3 * row + col]);
#pragma unroll
    for (int k = 0; k < 6; k++) st_out(&fv.merged[k * VS + a], SP[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) st_out(&fv.merged[(6 + k) * VS + a], Sv[k]);
    st_out(&fv.merged[9 * VS + a], SN);
    double s1, s2;
    vxm::gap_scales(lam, s1, s2);
    st_out(&fv.aux[a], s1);
    st_out(&fv.aux[VS + a], s2);
    st_out(&fv.aux[2 * VS + a], 1.0 / SN);
    st_out(&fv.aux[3 * VS + a], sqrt(coe_v));
  }
  if (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_stamp(true, vb, 4); }
}

// ------------------------------------------------------------------------------------------------
// K2 + K3 in one launch behind the in-launch solve: vxba_k23.hpp
// ------------------------------------------------------------------------------------------------
#include "vxba_k23.hpp"

__global__ __launch_bounds__(1024) void sum_partials_kernel(const double* __restrict__ partial, int n, double* __restrict__ out) {
  __shared__ double red[1024];
  double s = 0.0;
  for (int k = threadIdx.x; k < n; k += 1024) s += partial[k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void seed_aux_kernel(FactorView fv, int head, int end) {
  const int a = head + blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= end) return;
  const size_t VS = (size_t)fv.VS;
  const double lam[3] = {fv.eigval[a], fv.eigval[VS + a], fv.eigval[2 * VS + a]};
  double s1, s2;
  vxm::gap_scales(lam, s1, s2);
  fv.aux[a] = s1;
  fv.aux[VS + a] = s2;
  fv.aux[2 * VS + a] = 1.0 / fv.merged[9 * VS + a];
  fv.aux[3 * VS + a] = sqrt(fv.coe[a]);
}

// Voxel-sharded runs: k3_finalize only produced this rank's share of the packed buffer; after the all-reduce the LM state
// (what the solve reads) is filled from the REDUCED buffer here.  Gated like the sweep itself: when the sweep was skipped
// (rejected step, loop done) the buffer holds stale data that has just been summed again -- and is ignored.
__global__ __launch_bounds__(256) void lm_unpack_kernel(LMState* __restrict__ st, int cb, const double* __restrict__ packed, int n) {
  if (st->ctl[cb].done || !st->ctl[cb].calc_hess) return;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n * n) {
    const int r = e % n, c = e / n;
    const double h = packed[e];
    st->hess_out[e] = h;
    st->Hwork[e] = (r < 6 || c < 6) ? ((r == c) ? 1.0 : 0.0) : h;
  } else if (e < n * n + n) {
    const int i = e - n * n;
    st->Jwork[i] = i < 6 ? 0.0 : packed[e];
  } else if (e == n * n + n) {
    const double t0 = packed[e];
    st->ctl[cb].residual1 = t0;
    if (st->ctl[cb].iter == 0) st->ctl[cb].resis[0] = t0;
  }
}

// Sharded speculative loop: the Hessian sweep that just ran linearised at the TRIAL poses without knowing whether they would be
// accepted; the reduced buffer carries the system there AND (slot n^2+n+1) the trial residual.  Here the pending accept/reject
// decision is taken (into the other control block) and the system is adopted if the step was accepted (or a new window starts,
// or this is the first sweep of a solve); after a rejection the kept system stays and the sweep's work is discarded.
__global__ __launch_bounds__(256) void lm_spec_unpack_kernel(LMState* __restrict__ st, int c_in, const double* __restrict__ packed, int W, int has_pending,
                                                            int restart, PoseArg x0) {
  const int n = 6 * W;
  const LMCtl& in = st->ctl[c_in];
  if (in.done) { if (has_pending && blockIdx.x == 0) lm_carry(st, c_in, W); return; }
  // every workgroup takes the same decision from the same inputs; workgroup 0 persists it (into the other control block,
  // so nobody reads what it writes)
  int c_out = c_in;
  bool adopt = true;
  if (has_pending) {
    const LMDecision d = lm_decide(in, packed[n * n + n + 1], restart);
    if (blockIdx.x == 0) lm_persist(st, c_in, d, restart, x0, W);
    adopt = (d.accept || restart) && !d.done;
    c_out = c_in ^ 1;
  }
  if (!adopt) return;
  if (blockIdx.x == 0) {
    __syncthreads();   // lm_persist copied the old residual1 first
    if (threadIdx.x == 0) {
      const double t0 = packed[n * n + n];
      st->ctl[c_out].residual1 = t0;
      if (st->ctl[c_out].iter == 0) st->ctl[c_out].resis[0] = t0;
    }
  }
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n * n + n; e += gridDim.x * 256) {
    if (e < n * n) {
      const int r = e % n, c = e / n;
      const double h = packed[e];
      st->hess_out[e] = h;
      st->Hwork[e] = (r < 6 || c < 6) ? ((r == c) ? 1.0 : 0.0) : h;
    } else {
      const int i = e - n * n;
      st->Jwork[i] = i < 6 ? 0.0 : packed[e];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K1 -- cluster build.  256 consecutive cells per workgroup; their points are one contiguous range, staged
// through LDS with fully coalesced loads; each lane then folds its own cell's points in input order with
// unfused multiply/add, i.e. exactly PointCluster::push (tools.hpp:326-331) -- bit-identical to the CPU.
// ------------------------------------------------------------------------------------------------
constexpr int K1_CHUNK = 1024;  // points per LDS stage (24 KB)
// Cells longer than this are not folded by one lane (a 100k-point node of a coarse top-level voxelisation took 95 ms that way) but by
// a whole workgroup (k1_long_cells_kernel).  Up to this length the sum is the reference's sequential one, bit for bit; beyond it the
// order is a different fixed one (deterministic, ~1e-16 relative from the sequential sum).
constexpr long long K1_LONG = 2048;

template <bool AOS>
__global__ __launch_bounds__(256) void k1_build_kernel(const double* __restrict__ xyz, const long long* __restrict__ cell_ptr,
                                                       int n_voxels, int W, FactorView fv, int v0, double* __restrict__ aos, long long ncells_aos) {
#pragma clang fp contract(off)
  __shared__ double pts[3 * K1_CHUNK];
  const int tid = threadIdx.x;
  const long long ncells = AOS ? ncells_aos : (long long)n_voxels * W;
  const long long c0 = (long long)blockIdx.x * 256;
  const long long c = c0 + tid;
  const long long cend = (c0 + 256 < ncells) ? c0 + 256 : ncells;
  const long long p_begin = cell_ptr[c0], p_end = cell_ptr[cend];
  long long lo = p_end, hi = p_end;
  if (c < ncells) { lo = cell_ptr[c]; hi = cell_ptr[c + 1]; }
  const bool is_long = AOS && hi - lo > K1_LONG;     // left to k1_long_cells_kernel
  if (is_long) lo = hi;
  double P0 = 0, P1 = 0, P2 = 0, P3 = 0, P4 = 0, P5 = 0, vx = 0, vy = 0, vz = 0, N = 0;
  // Staging pays when the workgroup's 256 cells are small (a stage then feeds every lane).  With big cells (hundreds of points each:
  // the world clusters of whole nodes) a stage overlaps one or two cells, the other lanes idle through every stage, and the workgroup
  // degenerates into ONE lane folding all its points (21 ms for a coarse top-level layer).  There every lane streams its own cell
  // straight from memory instead -- same sequential sum per cell, all lanes busy.
  if (AOS && p_end - p_begin > 256 * 24) {
    for (long long q = lo; q < hi; q++) {
      const double x = xyz[3 * q], y = xyz[3 * q + 1], z = xyz[3 * q + 2];
      N += 1.0;
      P0 += x * x; P1 += x * y; P2 += x * z; P3 += y * y; P4 += y * z; P5 += z * z;
      vx += x; vy += y; vz += z;
    }
    if (c < ncells && !is_long) {
      double* o = aos + 10 * c;
      o[0] = P0; o[1] = P1; o[2] = P2; o[3] = P3; o[4] = P4; o[5] = P5; o[6] = vx; o[7] = vy; o[8] = vz; o[9] = N;
    }
    return;
  }
  for (long long chunk = p_begin; chunk < p_end; chunk += K1_CHUNK) {
    const int cnt = (int)((p_end - chunk < K1_CHUNK) ? (p_end - chunk) : K1_CHUNK);
    if (AOS) {   // stages that lie entirely inside long cells are not loaded
      const int need = lo < chunk + cnt && hi > chunk;
      if (!__syncthreads_or(need)) continue;
    }
    const double* src = xyz + 3 * chunk;
    for (int k = tid; k < 3 * cnt; k += 256) pts[k] = src[k];
    __syncthreads();
    const long long s = lo > chunk ? lo : chunk;
    const long long e = hi < chunk + cnt ? hi : chunk + cnt;
    for (long long q = s; q < e; q++) {
      const int j = 3 * (int)(q - chunk);
      const double x = pts[j], y = pts[j + 1], z = pts[j + 2];
      N += 1.0;
      P0 += x * x; P1 += x * y; P2 += x * z; P3 += y * y; P4 += y * z; P5 += z * z;
      vx += x; vy += y; vz += z;
    }
    __syncthreads();
  }
  if (AOS) {
    if (c < ncells && !is_long) {
      double* o = aos + 10 * c;
      o[0] = P0; o[1] = P1; o[2] = P2; o[3] = P3; o[4] = P4; o[5] = P5; o[6] = vx; o[7] = vy; o[8] = vz; o[9] = N;
    }
  } else if (c < ncells) {
    const int i = (int)(c / n_voxels);
    const int a = v0 + (int)(c % n_voxels);
    const size_t VS = (size_t)fv.VS;
    double* o = fv.cl + (size_t)i * 10 * VS + a;
    o[0] = P0; o[VS] = P1; o[2 * VS] = P2; o[3 * VS] = P3; o[4 * VS] = P4; o[5 * VS] = P5;
    o[6 * VS] = vx; o[7 * VS] = vy; o[8 * VS] = vz; o[9 * VS] = N;
  }
}

// K1 for packed cells (the voxeliser's cell / node clusters, vxba_build_clusters), round 5: SIXTEEN LANES PER CELL.  k1_build_kernel<true> above gives
// every lane a cell of its own -- fine for the frame-major planes of push_points (hundreds of thousands of small cells), but the voxeliser's launches
// are 5k-40k cells of 5-400 points: 20-160 workgroups, each lane streaming its own cell from memory point by point (24-byte strided loads) -- 50 us
// per launch, 0.04 of the HBM roofline, the dominant kernel of a hierarchical-BA pass (profiles/r05_cfg5).  Here a 16-lane row takes one cell: the
// lanes load sixteen consecutive points (one coalesced 384-byte run), every lane then folds those sixteen points IN ORDER out of its neighbours'
// registers (v_mov_b64_dpp row_newbcast: no LDS, no barrier) -- the sums are the sequential sums of PointCluster::push, bit for bit, sixteen times
// over -- and lane 0 of the row stores them.  Sixteen cells per workgroup: ten to a hundred times as many workgroups in flight.
template <int K>
__device__ __forceinline__ double row_bcast(double v) {   // lane K of the caller's 16-lane row, to every lane of that row
  typedef long long i64;
  return __builtin_bit_cast(double, (i64)__builtin_amdgcn_update_dpp((i64)0, __builtin_bit_cast(i64, v), 0x150 + K, 0xf, 0xf, false));
}
template <int K>
__device__ __forceinline__ void k1_fold16(double x, double y, double z, long long left, double& P0, double& P1, double& P2, double& P3, double& P4, double& P5, double& vx,
                                          double& vy, double& vz, double& N) {
#pragma clang fp contract(off)
  if constexpr (K < 16) {
    // points behind the cell's end arrive as (+0, +0, +0): adding their products and themselves changes no sum; only the count is gated
    const double a = row_bcast<K>(x), b = row_bcast<K>(y), c = row_bcast<K>(z);
    N += (K < left) ? 1.0 : 0.0;
    P0 += a * a; P1 += a * b; P2 += a * c; P3 += b * b; P4 += b * c; P5 += c * c;
    vx += a; vy += b; vz += c;
    k1_fold16<K + 1>(x, y, z, left, P0, P1, P2, P3, P4, P5, vx, vy, vz, N);
  }
}
__global__ __launch_bounds__(256) void k1_build_rows_kernel(const double* __restrict__ xyz, const long long* __restrict__ cell_ptr, long long ncells,
                                                            double* __restrict__ aos) {
#pragma clang fp contract(off)
  const int gl = threadIdx.x & 15;
  const long long c = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  long long lo = 0, hi = 0;
  if (c < ncells) { lo = cell_ptr[c]; hi = cell_ptr[c + 1]; }
  const bool is_long = hi - lo > K1_LONG;               // left to k1_long_cells_kernel
  if (is_long) hi = lo;
  double P0 = 0, P1 = 0, P2 = 0, P3 = 0, P4 = 0, P5 = 0, vx = 0, vy = 0, vz = 0, N = 0;
  // rows of one wave run as long as the longest of their four cells (a row that is done folds zeros)
  long long span = hi - lo;
  span = max(span, __shfl_xor(span, 16));
  span = max(span, __shfl_xor(span, 32));
  for (long long q0 = 0; q0 < span; q0 += 16) {
    const long long q = lo + q0 + gl;
    double x = 0.0, y = 0.0, z = 0.0;
    if (q < hi) { x = xyz[3 * q]; y = xyz[3 * q + 1]; z = xyz[3 * q + 2]; }
    k1_fold16<0>(x, y, z, hi - lo - q0, P0, P1, P2, P3, P4, P5, vx, vy, vz, N);
  }
  if (gl == 0 && c < ncells && !is_long) {                // an empty cell gets its ten zeros, as before
    double* o = aos + 10 * c;
    o[0] = P0; o[1] = P1; o[2] = P2; o[3] = P3; o[4] = P4; o[5] = P5; o[6] = vx; o[7] = vy; o[8] = vz; o[9] = N;
  }
}

// Long cells (> K1_LONG points): one workgroup per cell, points staged through LDS tile by tile, thread t folds points 4t .. 4t+3 of
// every tile, fixed-order tree over the 256 partial clusters.
__global__ __launch_bounds__(256) void k1_long_cells_kernel(const double* __restrict__ xyz, const long long* __restrict__ cell_ptr, long long ncells,
                                                            double* __restrict__ aos) {
#pragma clang fp contract(off)
  __shared__ double pts[3 * K1_CHUNK];
  __shared__ double red[10][256];
  __shared__ unsigned long long mask_s;
  const int tid = threadIdx.x;
  // tiles of 64 cells (one ballot), dealt round-robin to the workgroups: consecutive long cells spread over many workgroups
  const long long ntiles = (ncells + 63) / 64;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (tid < 64) {
      const long long c = tile * 64 + tid;
      bool lng = false;
      if (c < ncells) lng = cell_ptr[c + 1] - cell_ptr[c] > K1_LONG;
      const unsigned long long m = __ballot(lng);
      if (tid == 0) mask_s = m;
    }
    __syncthreads();
    {
      unsigned long long mm = mask_s;
      while (mm) {
        const int bit = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        const long long cc = tile * 64 + bit;
        const long long lo = cell_ptr[cc], hi = cell_ptr[cc + 1];
        double a[10];
#pragma unroll
        for (int k = 0; k < 10; k++) a[k] = 0.0;
        for (long long chunk = lo; chunk < hi; chunk += K1_CHUNK) {
          const int cnt = (int)((hi - chunk < K1_CHUNK) ? (hi - chunk) : K1_CHUNK);
          const double* src = xyz + 3 * chunk;
          __syncthreads();
          for (int k = tid; k < 3 * cnt; k += 256) pts[k] = src[k];
          __syncthreads();
#pragma unroll
          for (int j = 0; j < K1_CHUNK / 256; j++) {
            const int q = (K1_CHUNK / 256) * tid + j;
            if (q < cnt) {
              const double x = pts[3 * q], y = pts[3 * q + 1], z = pts[3 * q + 2];
              a[9] += 1.0;
              a[0] += x * x; a[1] += x * y; a[2] += x * z; a[3] += y * y; a[4] += y * z; a[5] += z * z;
              a[6] += x; a[7] += y; a[8] += z;
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 10; k++) red[k][tid] = a[k];
        __syncthreads();
        for (int step = 128; step >= 1; step >>= 1) {
          if (tid < step) {
#pragma unroll
            for (int k = 0; k < 10; k++) red[k][tid] += red[k][tid + step];
          }
          __syncthreads();
        }
        if (tid < 10) aos[10 * cc + tid] = red[tid][0];
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// K4 -- batched plane fit: eig(P/N - c c^T)   (voxel_map.hpp:1161-1163)
// ------------------------------------------------------------------------------------------------
__global__ void k4_plane_fit_kernel(const double* __restrict__ clusters, long long n, double* __restrict__ eigval, double* __restrict__ eigvec,
                                    PlaneCriteria crit, unsigned char* __restrict__ flags) {
  const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const double* c = clusters + 10 * a;
  double Cm[6], lam[3], U[9];
  vxm::cluster_cov(c, c + 6, c[9], Cm);
  vxm::eig_sym3(Cm, lam, U);
  if (flags) {
    // voxel_map.hpp:1155 (N > min_point), :1015-1019 (plane_judge), :1314 (factor filter lambda0/lambda1 <= 0.12)
    unsigned char fl = 0;
    if ((int)c[9] > crit.min_point) fl |= 1;
    if (lam[0] < crit.min_eigen_value && (lam[0] / lam[2]) < crit.eigen_ratio_thre) fl |= 2;
    if (!(lam[0] / lam[1] > crit.factor_ratio_max)) fl |= 4;
    flags[a] = fl;
  }
  for (int k = 0; k < 3; k++) eigval[3 * a + k] = lam[k];
  for (int col = 0; col < 3; col++)
    for (int row = 0; row < 3; row++) eigvec[9 * a + 3 * col + row] = U[3 * row + col];
}

// ------------------------------------------------------------------------------------------------
// layout plumbing
// ------------------------------------------------------------------------------------------------
// frame-major planes -> K3's batch-major copy for the batches touching voxels [v0, v0+n)
__global__ void build_clb_kernel(FactorView fv, int nv, int V_hi, int b_lo, int nbatches) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (batch, pair j, lane)
  if (t >= (long long)nbatches * 320) return;
  const int lane = (int)(t % 64), j = (int)((t / 64) % 5), b = b_lo + (int)(t / 320);
  const int W = fv.W;
  double x0 = 0.0, x1 = 0.0;
  if (lane < nv * W) {
    const int a = b * nv + lane / W, i = lane % W;
    if (a < V_hi) {
      const double* s = fv.cl + ((size_t)i * 10 + 2 * j) * fv.VS + a;
      x0 = s[0];
      x1 = s[fv.VS];
    }
  }
  double* d = fv.clb + (((size_t)b * 5 + j) * 64 + lane) * 2;
  d[0] = x0;
  d[1] = x1;
}

// frame-major f64 planes -> the f32 re-centred copy the residual sweep reads under VXBA_OPT_F32_CLUSTERS, voxels [v0, v0+n)
__global__ void build_cl32_kernel(FactorView fv, int v0, int n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (frame, voxel)
  if (t >= (long long)n * fv.W) return;
  const int a = v0 + (int)(t % n), i = (int)(t / n);
  const size_t VS = (size_t)fv.VS;
  double c[10];
  float r[10];
#pragma unroll
  for (int k = 0; k < 10; k++) c[k] = fv.cl[((size_t)i * 10 + k) * VS + a];
  vxm::cluster_to_centred_f32(c, r);
#pragma unroll
  for (int k = 0; k < 10; k++) fv.cl32[((size_t)i * 10 + k) * VS + a] = r[k];
}

__global__ void scatter_clusters_kernel(const double* __restrict__ src, FactorView fv, int v0, int n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = fv.W;
  if (t >= (long long)n * W) return;
  const int al = (int)(t % n), i = (int)(t / n);
  const double* s = src + ((size_t)al * W + i) * 10;
  double* o = fv.cl + (size_t)i * 10 * fv.VS + v0 + al;
  for (int k = 0; k < 10; k++) o[(size_t)k * fv.VS] = s[k];
}
// CSR input: entry e of voxel v0 + a (row_ptr[a] <= e < row_ptr[a + 1]) is the cluster of frame frame_idx[e]; frames that are not listed
// are unobserved (all-zero cluster, N == 0).  One thread per (voxel, frame) slot zeroes it, then one thread per entry fills its slot.
__global__ void csr_zero_kernel(FactorView fv, int v0, int n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = (long long)n * fv.W * 10;
  if (t >= tot) return;
  const int a = (int)(t % n);
  const long long pk = t / n;                // plane index frame * 10 + k
  fv.cl[(size_t)pk * fv.VS + v0 + a] = 0.0;
}
__global__ void csr_scatter_kernel(const long long* __restrict__ row_ptr, const int* __restrict__ frame_idx, const double* __restrict__ clusters, FactorView fv, int v0,
                                   int n, int* __restrict__ bad) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  int prev = -1;
  for (long long e = row_ptr[a]; e < row_ptr[a + 1]; e++) {
    const int fr = frame_idx[e];
    if (fr <= prev || fr >= fv.W) { atomicExch(bad, 1); return; }      // frames must be strictly increasing inside a voxel and < win_size
    prev = fr;
    const double* c = clusters + (size_t)e * 10;
#pragma unroll
    for (int k = 0; k < 10; k++) fv.cl[((size_t)fr * 10 + k) * fv.VS + v0 + a] = c[k];
  }
}
__global__ void gather_clusters_kernel(FactorView fv, int head, int n, double* __restrict__ dst) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = fv.W;
  if (t >= (long long)n * W) return;
  const int al = (int)(t % n), i = (int)(t / n);
  double* d = dst + ((size_t)al * W + i) * 10;
  const double* s = fv.cl + (size_t)i * 10 * fv.VS + head + al;
  for (int k = 0; k < 10; k++) d[k] = s[(size_t)k * fv.VS];
}
__global__ void scatter_rows_kernel(const double* __restrict__ src, double* __restrict__ planes, int VS, int v0, int n, int K) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * K) return;
  const int al = (int)(t % n), k = (int)(t / n);
  planes[(size_t)k * VS + v0 + al] = src[(size_t)al * K + k];
}
// the five per-voxel records a pushed voxel carries besides its clusters (fix 10 | coe 1 | eigval 3 | eigvec 9 | merged 10) into their planes, one launch
__global__ void scatter_voxel_records_kernel(const double* __restrict__ fix, const double* __restrict__ coe, const double* __restrict__ eigval, const double* __restrict__ eigvec,
                                             const double* __restrict__ merged, FactorView fv, int v0, int n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * 33) return;
  const int al = (int)(t % n), k = (int)(t / n);
  const size_t VS = (size_t)fv.VS;
  if (k < 10) fv.fix[(size_t)k * VS + v0 + al] = fix[(size_t)al * 10 + k];
  else if (k == 10) fv.coe[v0 + al] = coe[al];
  else if (k < 14) fv.eigval[(size_t)(k - 11) * VS + v0 + al] = eigval[(size_t)al * 3 + (k - 11)];
  else if (k < 23) fv.eigvec[(size_t)(k - 14) * VS + v0 + al] = eigvec[(size_t)al * 9 + (k - 14)];
  else fv.merged[(size_t)(k - 23) * VS + v0 + al] = merged[(size_t)al * 10 + (k - 23)];
}
__global__ void gather_rows_kernel(const double* __restrict__ planes, int VS, int head, int n, int K, double* __restrict__ dst) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * K) return;
  const int al = (int)(t % n), k = (int)(t / n);
  dst[(size_t)al * K + k] = planes[(size_t)k * VS + head + al];
}
__global__ void fill_kernel(double* p, size_t n, double val) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) p[t] = val;
}
__global__ void copy_planes_kernel(const double* __restrict__ src, int src_vs, double* __restrict__ dst, int dst_vs, int nplanes, int n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)nplanes * n) return;
  const int a = (int)(t % n), k = (int)(t / n);
  dst[(size_t)k * dst_vs + a] = src[(size_t)k * src_vs + a];
}
__global__ void count_nnz_kernel(FactorView fv, int V, unsigned long long* out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long one = 0;
  if (t < (long long)V * fv.W) {
    const int a = (int)(t % V), i = (int)(t / V);
    one = fv.cl[((size_t)i * 10 + 9) * fv.VS + a] != 0.0;
  }
  const unsigned long long cnt = __popcll(__ballot(one != 0));
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, cnt);
}

// ------------------------------------------------------------------------------------------------
// LM shell on the device (what Lidar_BA_Optimizer::damping_iter does between the two sweeps, voxel_map.hpp:391-439).
// One workgroup; the (6W)^2 system lives in LDS.  LDL^T with symmetric diagonal pivoting like the reference's
// Eigen::LDLT (largest remaining |diagonal| first), then the damped step, the trial poses and q1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lm_init_kernel(LMState* stp, PoseArg x0, int W, int bench_mode) {
  LMCtl* st = &stp->ctl[0];
  const int t = threadIdx.x;
  if (t < 12 * W) { st->x[t] = x0.Rp[t]; st->xt[t] = x0.Rp[t]; }
  if (t == 0) {
    st->u = 0.01; st->v = 2.0;
    st->residual1 = 0; st->residual2 = 0; st->q1 = 0; st->resis[0] = 0; st->resis[1] = 0;
    st->calc_hess = 1; st->done = 0; st->iter = 0; st->converge = 1; st->rejected = 0; st->bench_mode = bench_mode;
    st->n_accept = 0; st->n_reject = 0;
    stp->error = 0;   // a timed-out in-launch solve of an earlier call must not fail this one (the host retries without fusion)
  }
}
// stand-alone launch of the damped solve (VXBA_FUSED_SOLVE=0 and the retry after a timed-out in-launch solve; the default runs it
// as workgroup 0 of the residual sweep): lm_solve_body4, vxba_solve4.hpp
template <int W, bool DBG = false>
__global__ __launch_bounds__(S4_THREADS) void lm_solve_kernel(LMState* st, int c) {
  __shared__ __attribute__((aligned(16))) double lds[S4<W>::DOUBLES];
  lm_solve_body4<W, DBG>(st, c, lds);
}

// Stand-alone decision kernel that closes the loop after the last residual sweep (inside the loop the decision is
// taken in the prologue of the next Hessian sweep).  One workgroup.
__global__ __launch_bounds__(256) void lm_update_kernel(LMState* st, int c_in, LMPending pend, PoseArg x0, int W) {
  if (st->ctl[c_in].done) { lm_carry(st, c_in, W); return; }
  const double r2 = lm_residual2(pend);
  const LMDecision d = lm_decide(st->ctl[c_in], r2, pend.restart);
  lm_persist(st, c_in, d, pend.restart, x0, W);
}

// Layout probe for v_mfma_f64_16x16x4_f64 (used by a GPU unit test): D(16x16) = A(16x4) B(4x16) with the operand /
// result lane maps K3 relies on -- A[i=l%16][k=l/16], B[k=l/16][j=l%16], D[row = l/16 + 4*reg][col = l%16].
__global__ __launch_bounds__(64) void mfma_probe_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ D) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];
  const double b = B[(l >> 4) * 16 + (l & 15)];
  v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; j++) D[((l >> 4) + 4 * j) * 16 + (l & 15)] = acc[j];
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void debug_read_stamps(unsigned long long* host, size_t n) {
  if (n > (size_t)DBG_WAVES * DBG_SLOTS) n = (size_t)DBG_WAVES * DBG_SLOTS;
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dbg), n * sizeof(unsigned long long));
}
void debug_clear_stamps() {
  static unsigned long long z[DBG_WAVES * DBG_SLOTS];
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof z);
}
void launch_mfma_probe(const double* dA, const double* dB, double* dD, hipStream_t s) {
  mfma_probe_kernel<<<dim3(1), dim3(64), 0, s>>>(dA, dB, dD);
}

#define VXK_DISPATCH_W(W, ...)                                 \
  switch (W) {                                                 \
    case 1: { constexpr int WW = 1; __VA_ARGS__; } break;      \
    case 2: { constexpr int WW = 2; __VA_ARGS__; } break;      \
    case 3: { constexpr int WW = 3; __VA_ARGS__; } break;      \
    case 4: { constexpr int WW = 4; __VA_ARGS__; } break;      \
    case 5: { constexpr int WW = 5; __VA_ARGS__; } break;      \
    case 6: { constexpr int WW = 6; __VA_ARGS__; } break;      \
    case 7: { constexpr int WW = 7; __VA_ARGS__; } break;      \
    case 8: { constexpr int WW = 8; __VA_ARGS__; } break;      \
    case 9: { constexpr int WW = 9; __VA_ARGS__; } break;      \
    case 10: { constexpr int WW = 10; __VA_ARGS__; } break;    \
    default: break;                                            \
  }

// Voxels per residual-sweep workgroup.  One lane per voxel, full waves (64) by default.  Tried: ceil(V / (k * cus)) voxels per
// workgroup so that every CU owns the same number of voxels (49 instead of 64 at cfg2, 4 workgroups on every CU): 20.0 us instead of
// 18.0 -- the partly filled waves cost more than the ragged last round.  VXBA_OPT_K2_VOXELS_PER_BLOCK keeps the experiment reproducible.
// Workgroups that take part in the in-launch Hessian reduction.  They must all be resident while the solve workgroup waits for them
// (they are the first ones dispatched, one 256-thread workgroup each: 192 + the solve fit any MI355X), and there must be enough of them
// for the reduction to take a pass or two (16 elements of the partial per workgroup and pass) -- else the stand-alone kernel is faster.
// Voxels per wave of the residual sweep for an option value (32..64, else 64) and the number of wave partials a sweep over nvoxels writes:
// the ONE place this geometry lives -- the launch, the in-launch reduction's sizing and the host shells' sentinel fills all call it.
int k2_voxels_per_wave(int voxels_per_block) {
  const int vpb0 = voxels_per_block & 0xffff;
  return (vpb0 >= 32 && vpb0 <= 64) ? vpb0 : 64;
}
int k2_nparts(int nvoxels, int voxels_per_block) {
  const int vpb = k2_voxels_per_wave(voxels_per_block);
  return nvoxels > 0 ? (nvoxels + vpb - 1) / vpb : 0;
}
int launch_k2_residual(const FactorView& fv, const PoseArg& poses, LMState* st, int c, unsigned fused_seq, int head, int end, double* d_partial,
                       int voxels_per_block, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, const double* host_feed, const double* li_rec, double* li_out) {
  const int vpb = k2_voxels_per_wave(voxels_per_block);
  const int vpb_arg = vpb | (voxels_per_block & 0x10000);   // bit 16: the voxel workgroups do not wait for the in-launch solve (test hook)
  const int nblocks = k2_nparts(end - head, voxels_per_block);          // voxel WAVES = partials
  if (nblocks <= 0) return 0;
  const unsigned seq = st ? fused_seq : 0u;
  const int grid = (nblocks + K2_WAVES - 1) / K2_WAVES + (seq != 0 ? 1 : 0);   // + the solve workgroup
  static int dbg = -1, head_start = -1;
  if (dbg < 0) { const char* ev = getenv("VXBA_DBG"); dbg = (ev && ev[0] == '1') ? 1 : 0; }
  if (head_start < 0) { const char* ev = getenv("VXBA_K2_HEAD_START"); head_start = ev ? atoi(ev) : K2_HEAD_START; if (head_start < 0) head_start = 0; }   // development knob, x 64 cycles
  const dim3 g(grid), b(K2_THREADS);
  if (fv.cl32) {   // f32 re-centred cluster rows (the caller built them: vxba_capi.hip, residual_view)
    if (ev_start) {
      VXK_DISPATCH_W(fv.W, hipExtLaunchKernelGGL((k2_residual_kernel<WW, false, true>), g, b, 0, s, ev_start, ev_stop, 0, st, c, seq, li_rec, li_out, host_feed, head, end, vpb_arg, head_start, d_partial, fv, poses));
    } else { VXK_DISPATCH_W(fv.W, (k2_residual_kernel<WW, false, true><<<g, b, 0, s>>>(st, c, seq, li_rec, li_out, host_feed, head, end, vpb_arg, head_start, d_partial, fv, poses))); }
  } else if (dbg) { VXK_DISPATCH_W(fv.W, k2_residual_kernel<WW, true><<<g, b, 0, s>>>(st, c, seq, li_rec, li_out, host_feed, head, end, vpb_arg, head_start, d_partial, fv, poses)); }
  else if (ev_start) {
    VXK_DISPATCH_W(fv.W, hipExtLaunchKernelGGL((k2_residual_kernel<WW, false>), g, b, 0, s, ev_start, ev_stop, 0, st, c, seq, li_rec, li_out, host_feed, head, end, vpb_arg, head_start, d_partial, fv, poses));
  } else { VXK_DISPATCH_W(fv.W, k2_residual_kernel<WW><<<g, b, 0, s>>>(st, c, seq, li_rec, li_out, host_feed, head, end, vpb_arg, head_start, d_partial, fv, poses)); }
  return nblocks;
}

void launch_sum_partials(const double* d_partial, int n, double* d_out, hipStream_t s) {
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(1024), 0, s, d_partial, n, d_out);
}

void launch_seed_aux(const FactorView& fv, int head, int end, hipStream_t s) {
  if (end <= head) return;
  hipLaunchKernelGGL(seed_aux_kernel, dim3((end - head + 255) / 256), dim3(256), 0, s, fv, head, end);
}

int k3_grid_blocks(int device_cus) { return device_cus; }  // one 8-wave workgroup per CU = two waves per SIMD

// dynamic LDS of the Hessian sweep: two tile buffers + the poses, or the epilogue's parking area, whichever is larger
template <int W>
constexpr size_t k3_lds_bytes() {
  using C = K3Cfg<W>;
  constexpr size_t main_d = (size_t)2 * C::BUF + 24 * W + 8 + (size_t)C::WAVES * K3Stage<W>::WAVE_DOUBLES + 32;   // tiles | poses | LM decision inputs | parameter staging | dump for the idle lanes' rows
  constexpr size_t epi = (size_t)K3_BLOCK * K3_DS + (size_t)C::WAVES * C::PPWP * 16;   // parked linear accumulators | the waves' folded pair sums on their way out
  constexpr size_t m = main_d > epi ? main_d : epi;
  return m * sizeof(double);
}

int launch_k3_hessian(const FactorView& fv_in, const PoseArg& poses, LMState* st, int c_in, const LMPending& pend, const double* cache_src,
                      int head, int end, double* d_partial, int nblocks, int mixed, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  FactorView fv = fv_in;
  if (cache_src) {   // eigval(3) eigvec(9) merged(10) aux(4): 26 consecutive planes
    const size_t VS = (size_t)fv.VS;
    fv.eigval = const_cast<double*>(cache_src);
    fv.eigvec = fv.eigval + 3 * VS;
    fv.merged = fv.eigvec + 9 * VS;
    fv.aux = fv.merged + 10 * VS;
  }
  {  // the sweep addresses the cache as 26 consecutive planes behind eigval (one buffer descriptor): true for the factor's live
     // cache and for snapshots; refuse anything else loudly rather than read the wrong plane
    const size_t VS = (size_t)fv.VS;
    if (fv.eigvec != fv.eigval + 3 * VS || fv.merged != fv.eigvec + 9 * VS || fv.aux != fv.merged + 10 * VS) return -1;
  }
  static int dbg = -1;   // development knob: VXBA_DBG=1 runs the s_memtime-instrumented instantiation
  if (dbg < 0) { const char* ev = getenv("VXBA_DBG"); dbg = (ev && ev[0] == '1') ? 1 : 0; }
#define K3_ARGS fv.clb, fv.eigval, fv.coe, st, (int)fv.VS, head, end, c_in, (pend.pending & 0xff) | (pend.restart << 8), nblocks, poses, pend, d_partial
  VXK_DISPATCH_W(fv.W, {
    constexpr size_t lds_bytes = k3_lds_bytes<WW>();
    static_assert(lds_bytes <= 160 * 1024, "K3 tile buffers exceed the CU's LDS");
    // > 64 KB of dynamic LDS must be opted into per kernel AND per device: one bit per (device, variant), set atomically
    // (a process-wide "done once" flag missed a second device and raced between factors on different threads)
    static std::atomic<unsigned long long> opted{0ull};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int variant = mixed ? 2 : (dbg ? 1 : 0);
    const unsigned long long bit = 1ull << ((3 * dev + variant) & 63);
    if (!(opted.load(std::memory_order_relaxed) & bit) || dev > 20) {
      const void* fn = mixed ? (const void*)k3_hessian_kernel<WW, false, true> : (dbg ? (const void*)k3_hessian_kernel<WW, true> : (const void*)k3_hessian_kernel<WW, false>);
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      opted.fetch_or(bit, std::memory_order_relaxed);
    }
    if (mixed) {
      if (ev_start)
        hipExtLaunchKernelGGL((k3_hessian_kernel<WW, false, true>), dim3(nblocks), dim3(K3_BLOCK), (uint32_t)lds_bytes, s, ev_start, ev_stop, 0, K3_ARGS);
      else k3_hessian_kernel<WW, false, true><<<dim3(nblocks), dim3(K3_BLOCK), lds_bytes, s>>>(K3_ARGS);
    } else if (dbg) {
      k3_hessian_kernel<WW, true><<<dim3(nblocks), dim3(K3_BLOCK), lds_bytes, s>>>(K3_ARGS);
    } else {
      if (ev_start)
        hipExtLaunchKernelGGL((k3_hessian_kernel<WW, false>), dim3(nblocks), dim3(K3_BLOCK), (uint32_t)lds_bytes, s, ev_start, ev_stop, 0, K3_ARGS);
      else k3_hessian_kernel<WW, false><<<dim3(nblocks), dim3(K3_BLOCK), lds_bytes, s>>>(K3_ARGS);
    }
  });
#undef K3_ARGS
  return nblocks;
}

void launch_k3_finalize(const double* d_partial, int nblocks, int W, LMState* st, int c, int write_state, double* d_packed, hipStream_t s, int force,
                        const double* k2_partial, int k2_nparts, double* reset_slots, int n_reset) {
  const int plen = (int)k3_partial_len(W);
  static int dbg = -1;
  if (dbg < 0) { const char* ev = getenv("VXBA_DBG"); dbg = (ev && ev[0] == '1') ? 1 : 0; }
  if (dbg) {
    VXK_DISPATCH_W(W, (k3_finalize_kernel<WW, true><<<dim3((plen + FIN_EL - 1) / FIN_EL), dim3(FIN_EL * FIN_SL), 0, s>>>(d_partial, nblocks, st, c, write_state, d_packed,
                                                                                                                    force, k2_partial, k2_nparts, reset_slots, n_reset)));
    return;
  }
  VXK_DISPATCH_W(W, k3_finalize_kernel<WW><<<dim3((plen + FIN_EL - 1) / FIN_EL), dim3(FIN_EL * FIN_SL), 0, s>>>(d_partial, nblocks, st, c, write_state, d_packed,
                                                                                                           force, k2_partial, k2_nparts, reset_slots, n_reset));
}
// dynamic LDS of the fused launch: the Hessian half's tiles, or the solve's block columns (workgroup 0), whichever is larger
template <int W>
constexpr size_t k23_lds_bytes() {
  constexpr size_t a = k3_lds_bytes<W>(), b = (size_t)S4<W>::DOUBLES * sizeof(double);
  return a > b ? a : b;
}
int k23_sweep_blocks(int nbatches, int device_cus) {   // one CU runs the solve
#ifdef VXBA_K23_DBG_CUS
  device_cus = VXBA_K23_DBG_CUS + 1;
#endif
  const int b = k3_blocks_for(nbatches, device_cus > 1 ? device_cus - 1 : 1);
  return b < K23_MAX_SWEEP_BLOCKS ? b : K23_MAX_SWEEP_BLOCKS;
}
bool k23_supported(const FactorView& fv) {
  // one buffer descriptor spans every plane: 32-bit offsets
  return !fv.cl32 && fv.W >= 1 && fv.W <= MAXW && (unsigned long long)(10 * fv.W + 37) * (unsigned long long)fv.VS * 8ull < (1ull << 32);
}
int launch_k23_fused(const FactorView& fv, LMState* st, int c, unsigned seq, int head, int end, double* d_partial2, double* d_partial3, int nwg, int mixed, int flags,
                     hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, const double* host_feed, const double* li_rec, double* li_out) {
  if (!st || seq == 0 || nwg < 1 || nwg > K23_MAX_SWEEP_BLOCKS || end <= head || !k23_supported(fv)) return -1;
  {  // the planes of a factor are one allocation in the order K23Planes assumes (vxc::view): refuse anything else loudly
    const size_t VS = (size_t)fv.VS, W = (size_t)fv.W;
    if (fv.fix != fv.cl + 10 * W * VS || fv.coe != fv.fix + 10 * VS || fv.eigval != fv.coe + VS || fv.eigvec != fv.eigval + 3 * VS || fv.merged != fv.eigvec + 9 * VS ||
        fv.aux != fv.merged + 10 * VS)
      return -1;
  }
  static int dbg = -1, head_start = -1;
  if (dbg < 0) { const char* ev = getenv("VXBA_DBG"); dbg = (ev && (ev[0] == '1' || ev[0] == '2')) ? 1 : 0; }   // 2: this launch only (the other kernels' stamps share the table's rows)
  if (head_start < 0) { const char* ev = getenv("VXBA_K2_HEAD_START"); head_start = ev ? atoi(ev) : K2_HEAD_START; if (head_start < 0) head_start = 0; }
  const int flags_hs = (flags & 0xff) | (head_start << 8);
#define K23_ARGS st, c, seq, li_rec, li_out, host_feed, head, end, nwg, flags_hs, fv.cl, fv.clb, (int)fv.VS, d_partial2, d_partial3
#define K23_LAUNCH(...)                                                                                                      \
  do {                                                                                                                     \
    if (ev_start) hipExtLaunchKernelGGL((k23_fused_kernel<__VA_ARGS__>), gr, bl, (uint32_t)lds_bytes, s, ev_start, ev_stop, 0, K23_ARGS); \
    else k23_fused_kernel<__VA_ARGS__><<<gr, bl, lds_bytes, s>>>(K23_ARGS);                                                \
  } while (0)
  static int pair_env = -1;   // development knob: VXBA_K23_PAIR=0 keeps the one-lane-per-voxel residual half at every size
  if (pair_env < 0) { const char* ev = getenv("VXBA_K23_PAIR"); pair_env = (ev && ev[0] == '0') ? 0 : 1; }
  VXK_DISPATCH_W(fv.W, {
    constexpr size_t lds_bytes = k23_lds_bytes<WW>();
    static_assert(lds_bytes + 64 <= 160 * 1024, "fused launch: LDS");
    // PAIR (a voxel's frames over a lane pair, vxba_k23.hpp): when no sweep workgroup owns more than 256 voxels -- the kernel's own split
    constexpr int NV = K3Cfg<WW>::NV;
    const int nb_all = (end - 1) / NV - head / NV + 1;
    const bool pair = pair_env && ((nb_all + nwg - 1) / nwg) * NV <= K3_BLOCK / 2;
    static std::atomic<unsigned long long> opted{0ull};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int variant = (mixed ? 2 : (dbg ? 1 : 0)) + (pair ? 3 : 0);
    const unsigned long long bit = 1ull << ((6 * dev + variant) & 63);
    const void* fn = pair ? (mixed ? (const void*)k23_fused_kernel<WW, false, true, true> : (dbg ? (const void*)k23_fused_kernel<WW, true, false, true> : (const void*)k23_fused_kernel<WW, false, false, true>))
                          : (mixed ? (const void*)k23_fused_kernel<WW, false, true> : (dbg ? (const void*)k23_fused_kernel<WW, true> : (const void*)k23_fused_kernel<WW, false>));
    if (!(opted.load(std::memory_order_relaxed) & bit) || dev > 9) {
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      opted.fetch_or(bit, std::memory_order_relaxed);
    }
    const dim3 gr(nwg + 1), bl(K3_BLOCK);
    if (pair) {
      if (mixed) K23_LAUNCH(WW, false, true, true);
      else if (dbg) K23_LAUNCH(WW, true, false, true);
      else K23_LAUNCH(WW, false, false, true);
    } else {
      if (mixed) K23_LAUNCH(WW, false, true, false);
      else if (dbg) K23_LAUNCH(WW, true, false, false);
      else K23_LAUNCH(WW, false, false, false);
    }
  });
#undef K23_LAUNCH
#undef K23_ARGS
  return nwg;
}

void launch_lm_spec_unpack(LMState* st, int c_in, const double* d_packed, int W, int has_pending, int restart, const PoseArg& x0, hipStream_t s) {
  const int n = 6 * W;
  lm_spec_unpack_kernel<<<dim3((n * n + n + 255) / 256), dim3(256), 0, s>>>(st, c_in, d_packed, W, has_pending, restart, x0);
}
void launch_lm_unpack(LMState* st, int c, const double* d_packed, int W, hipStream_t s) {
  const int n = 6 * W, total = n * n + n + 1;
  lm_unpack_kernel<<<dim3((total + 255) / 256), dim3(256), 0, s>>>(st, c, d_packed, n);
}

void launch_k1_build(const double* d_xyz, const int64_t* d_cell_ptr, int n_voxels, int W, const FactorView& fv, int v0, hipStream_t s) {
  const long long ncells = (long long)n_voxels * W;
  if (ncells <= 0) return;
  k1_build_kernel<false><<<dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, s>>>(d_xyz, (const long long*)d_cell_ptr, n_voxels, W, fv, v0, nullptr, 0);
}

void launch_k4_plane_fit(const double* d_clusters, int64_t n, double* d_eigval, double* d_eigvec, const PlaneCriteria* crit, unsigned char* d_flags,
                         hipStream_t s) {
  if (n <= 0) return;
  PlaneCriteria pc = crit ? *crit : PlaneCriteria{0, 0.0, 0.0, 0.0};
  k4_plane_fit_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(d_clusters, (long long)n, d_eigval, d_eigvec, pc, crit ? d_flags : nullptr);
}
void launch_k1_build_aos(const double* d_xyz, const int64_t* d_cell_ptr, int64_t n_cells, double* d_clusters, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (n_cells <= 0) return;
  // sixteen lanes per cell (k1_build_rows_kernel); same-box A/B against one lane per cell on a cfg5 pass: 28.3 against 52.3 us per launch (606 launches)
  const dim3 grid_rows((unsigned)((n_cells + 15) / 16));
  if (ev_start)   // events tied to the dispatch itself (the interval rocprofv3 reports)
    hipExtLaunchKernelGGL(k1_build_rows_kernel, grid_rows, dim3(256), 0, s, ev_start, ev_stop, 0, d_xyz, (const long long*)d_cell_ptr, (long long)n_cells, d_clusters);
  else
    k1_build_rows_kernel<<<grid_rows, dim3(256), 0, s>>>(d_xyz, (const long long*)d_cell_ptr, (long long)n_cells, d_clusters);
  const long long tiles = (n_cells + 63) / 64;
  k1_long_cells_kernel<<<dim3((unsigned)(tiles < 2048 ? tiles : 2048)), dim3(256), 0, s>>>(d_xyz, (const long long*)d_cell_ptr, (long long)n_cells, d_clusters);
}

static inline unsigned nblk(long long n, int b) { return (unsigned)((n + b - 1) / b); }

void launch_build_cl32(const FactorView& fv, int v0, int n, hipStream_t s) {
  if (n <= 0 || !fv.cl32) return;
  hipLaunchKernelGGL(build_cl32_kernel, dim3(nblk((long long)n * fv.W, 256)), dim3(256), 0, s, fv, v0, n);
}
void launch_build_clb(const FactorView& fv, int v0, int n, hipStream_t s) {
  if (n <= 0) return;
  const int nv = k3_nv(fv.W);
  const int b_lo = v0 / nv, b_hi = (v0 + n - 1) / nv;
  const int nbatches = b_hi - b_lo + 1;
  hipLaunchKernelGGL(build_clb_kernel, dim3(nblk((long long)nbatches * 320, 256)), dim3(256), 0, s, fv, nv, v0 + n, b_lo, nbatches);
}
void launch_scatter_clusters(const double* d_src, const FactorView& fv, int v0, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scatter_clusters_kernel, dim3(nblk((long long)n * fv.W, 256)), dim3(256), 0, s, d_src, fv, v0, n);
}
void launch_scatter_clusters_csr(const long long* d_row_ptr, const int* d_frame_idx, const double* d_clusters, const FactorView& fv, int v0, int n, int* d_bad,
                                 hipStream_t s) {
  if (n <= 0) return;
  const long long tot = (long long)n * fv.W * 10;
  csr_zero_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s>>>(fv, v0, n);
  csr_scatter_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(d_row_ptr, d_frame_idx, d_clusters, fv, v0, n, d_bad);
}
void launch_gather_clusters(const FactorView& fv, int head, int n, double* d_dst, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_clusters_kernel, dim3(nblk((long long)n * fv.W, 256)), dim3(256), 0, s, fv, head, n, d_dst);
}
void launch_scatter_voxel_records(const double* d_fix, const double* d_coe, const double* d_eigval, const double* d_eigvec, const double* d_merged, const FactorView& fv, int v0, int n,
                                  hipStream_t s) {
  if (n <= 0) return;
  const long long total = (long long)n * 33;
  hipLaunchKernelGGL(scatter_voxel_records_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_fix, d_coe, d_eigval, d_eigvec, d_merged, fv, v0, n);
}
void launch_scatter_rows(const double* d_src, double* planes, int VS, int v0, int n, int K, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(nblk((long long)n * K, 256)), dim3(256), 0, s, d_src, planes, VS, v0, n, K);
}
void launch_gather_rows(const double* planes, int VS, int head, int n, int K, double* d_dst, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(nblk((long long)n * K, 256)), dim3(256), 0, s, planes, VS, head, n, K, d_dst);
}
void launch_fill(double* p, size_t n, double val, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(fill_kernel, dim3(nblk((long long)n, 256)), dim3(256), 0, s, p, n, val);
}
void launch_copy_planes(const double* src, int src_vs, double* dst, int dst_vs, int nplanes, int n, hipStream_t s) {
  if (n <= 0 || nplanes <= 0) return;
  hipLaunchKernelGGL(copy_planes_kernel, dim3(nblk((long long)nplanes * n, 256)), dim3(256), 0, s, src, src_vs, dst, dst_vs, nplanes, n);
}
void launch_count_nnz(const FactorView& fv, int V, unsigned long long* d_out, hipStream_t s) {
  if (V <= 0) return;
  hipLaunchKernelGGL(count_nnz_kernel, dim3(nblk((long long)V * fv.W, 256)), dim3(256), 0, s, fv, V, d_out);
}

// what hipMemsetAsync(st, 0, sizeof(LMState)) does to the fields anything reads before writing them (the control scalars of both blocks, the
// error word) -- without a 70 KB fill in front of the first sweep of a call (the LiDAR-inertial shells reset the state on every call)
// Coverage (keep in step with LMState): ctl[].{u, v, residual1, residual2, q1, resis, calc_hess, done, iter, converge, rejected, bench_mode, n_accept,
// n_reject}, error, solve_seq.  NOT reset, because every reader is preceded by a writer inside the same call: ctl[].x / xt (lm_init or the
// shell's pose upload), trace (written per iteration, read up to `iter`), Jwork / Hwork / dxi / hess_out (k3_finalize writes them before the solve
// reads).  The block is zeroed once at allocation (vxba_create), so a future reader-before-writer sees zeros or a previous call's values, never
// uninitialised memory.
__global__ void lm_reset_kernel(LMState* st) {
  if (threadIdx.x < 2) {
    LMCtl& c = st->ctl[threadIdx.x];
    c.u = 0; c.v = 0; c.residual1 = 0; c.residual2 = 0; c.q1 = 0; c.resis[0] = 0; c.resis[1] = 0;
    c.calc_hess = 0; c.done = 0; c.iter = 0; c.converge = 0; c.rejected = 0; c.bench_mode = 0; c.n_accept = 0; c.n_reject = 0;
  }
  if (threadIdx.x == 2) { st->error = 0; st->solve_seq = 0; }
}
void launch_lm_reset(LMState* st, hipStream_t s) { lm_reset_kernel<<<dim3(1), dim3(64), 0, s>>>(st); }
void launch_lm_init(LMState* st, const PoseArg& x0, int W, int bench_mode, hipStream_t s) {
  lm_init_kernel<<<dim3(1), dim3(256), 0, s>>>(st, x0, W, bench_mode);
}
void launch_lm_solve(LMState* st, int c, int W, hipStream_t s) {
  static int dbg = -1;
  if (dbg < 0) { const char* ev = getenv("VXBA_DBG"); dbg = (ev && ev[0] == '1') ? 1 : 0; }
  if (dbg) { VXK_DISPATCH_W(W, lm_solve_kernel<WW, true><<<dim3(1), dim3(S4_THREADS), 0, s>>>(st, c)); }
  else { VXK_DISPATCH_W(W, lm_solve_kernel<WW><<<dim3(1), dim3(S4_THREADS), 0, s>>>(st, c)); }
}
void launch_lm_update(LMState* st, int c_in, const LMPending& pend, const PoseArg& restart_x0, int W, hipStream_t s) {
  lm_update_kernel<<<dim3(1), dim3(256), 0, s>>>(st, c_in, pend, restart_x0, W);
}

}  // namespace vxk
