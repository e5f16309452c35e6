// Sweeps for WIDE windows (win_size 11 .. 128): the top level of the hierarchical BA runs Lidar_BA_Optimizer on ~100 submap poses
// (voxelslam.cpp:2485-2595, loop_refine.hpp:273-537) where a voxel is seen from a handful of the frames.  The MFMA kernels of
// vxba_kernels.hip are built around 6W <= 64 dense columns; here the incidence is sparse, the Hessian is (6W)^2 = up to 768^2, and
// the per-voxel work is sum over observed PAIRS -- so: same mathematics (vxm::k3_entry rows, H = blockdiag(D) - G^T G), different
// mapping.  Clusters live in a compressed-row store over the observed (voxel, frame) entries (WideStore, vxba_wide.h); the per-voxel
// planes (fix, coe, cache) are those of every factor.
#include "vxba_wide.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include "vxba_math.hpp"

namespace vxw {

using vxk::FactorView;

typedef double v2d_t __attribute__((ext_vector_type(2)));
// an entry's cluster: ten consecutive doubles (80 bytes, 16-byte aligned) -- five 16-byte loads, one or two cache lines
__device__ __forceinline__ void load_entry(const double* __restrict__ ecl, long long e, double c[10]) {
  const v2d_t* q = reinterpret_cast<const v2d_t*>(ecl + (size_t)e * 10);
#pragma unroll
  for (int k = 0; k < 5; k++) { const v2d_t t = q[k]; c[2 * k] = t.x; c[2 * k + 1] = t.y; }
}

__global__ __launch_bounds__(64) void k2_wide_kernel(WideView wv, const double* __restrict__ poses, int head, int end, double* __restrict__ partial) {
  __shared__ double pl[12 * WIDE_MAXW];
  const FactorView& fv = wv.fv;
  const int lane = threadIdx.x, W = fv.W;
  for (int k = lane; k < 12 * W; k += 64) pl[k] = poses[k];
  __syncthreads();
  const int a = head + blockIdx.x * 64 + lane;
  const size_t VS = (size_t)fv.VS;
  double res = 0.0;
  if (a < end) {
    double SP[6], Sv[3], SN, Up[9];
    const double coe = fv.coe[a];   // with the other loads: read behind the cache stores below, the load cannot move above them and the wave waits it out at its very end
#pragma unroll
    for (int k = 0; k < 6; k++) SP[k] = fv.fix[k * VS + a];
#pragma unroll
    for (int k = 0; k < 3; k++) Sv[k] = fv.fix[(6 + k) * VS + a];
    SN = fv.fix[9 * VS + a];
#pragma unroll
    for (int col = 0; col < 3; col++)
#pragma unroll
      for (int row = 0; row < 3; row++) Up[3 * row + col] = fv.eigvec[(size_t)(3 * col + row) * VS + a];
    const long long e1 = wv.eptr[a + 1];
    for (long long e = wv.eptr[a]; e < e1; e++) {   // the voxel's observed frames, ascending (voxel_map.hpp:256-262)
      const int i = wv.eframe[e];
      double c[10];
      load_entry(wv.ecl, e, c);
      if (c[9] == 0.0) continue;         // an entry without points contributes nothing (voxel_map.hpp:258)
      double R[9], p[3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) R[3 * r + cc] = pl[12 * i + 3 * cc + r];
#pragma unroll
      for (int k = 0; k < 3; k++) p[k] = pl[12 * i + 9 + k];
      vxm::transform_accumulate(c, c + 6, c[9], R, p, SP, Sv, SN);
    }
    double C[6], lam[3], U[9];
    vxm::cluster_cov(SP, Sv, SN, C);
    vxm::eig_sym3_warm(C, Up, lam, U);
#pragma unroll
    for (int k = 0; k < 3; k++) fv.eigval[k * VS + a] = lam[k];
#pragma unroll
    for (int col = 0; col < 3; col++)
#pragma unroll
      for (int row = 0; row < 3; row++) fv.eigvec[(3 * col + row) * VS + a] = U[3 * row + col];
#pragma unroll
    for (int k = 0; k < 6; k++) fv.merged[k * VS + a] = SP[k];
#pragma unroll
    for (int k = 0; k < 3; k++) fv.merged[(6 + k) * VS + a] = Sv[k];
    fv.merged[9 * VS + a] = SN;
    double s1, s2;
    vxm::gap_scales(lam, s1, s2);
    fv.aux[a] = s1;
    fv.aux[VS + a] = s2;
    fv.aux[2 * VS + a] = 1.0 / SN;
    fv.aux[3 * VS + a] = sqrt(coe);
    res = coe * lam[0];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) res += __shfl_down(res, off);
  if (lane == 0) partial[blockIdx.x] = res;
}

__device__ __forceinline__ int sym6(int a, int b) { return a == 0 ? b : (a == 1 ? 2 + b : 5); }   // a <= b < 3

// ------------------------------------------------------------------------------------------------------------------
// Pair-major Hessian assembly (deterministic, no atomics).
// The incidence structure of a wide factor -- which (voxel, frame) entries exist and which entry PAIRS share a voxel --
// depends on the clusters only, not on the poses, so it is built once per factor content (WideIndex, rebuilt after a push):
//   entries  e -> (voxel, frame), voxel-major, frames ascending
//   pairs    (e_i, e_j) of one voxel with frame_i <= frame_j, keyed by frame_i * W + frame_j, stably radix-sorted by key:
//            every 6x6 block of the Hessian becomes one contiguous run of pair records, in voxel order.
// A sweep is then (A) one lane per entry: the rank-3 rows and the gradient / block-diagonal terms (vxm::k3_entry) into a row
// buffer, (B) one wave per Hessian block: the lanes stride over the block's pair records, each accumulating the 36 products
// in registers, a fixed butterfly adds the lanes, and the block (and its mirror image) is written -- each block is owned by
// exactly one wave, so the result is bitwise reproducible.
// ------------------------------------------------------------------------------------------------------------------

__global__ void wi_paircount_kernel(const long long* __restrict__ eptr, int V, long long* __restrict__ pc) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < V) { const long long k = eptr[a + 1] - eptr[a]; pc[a] = k * (k + 1) / 2; }
  if (a == V) pc[a] = 0;
}
// pairs of voxel a at pair_ptr[a]..: key, original position (payload of the sort) and the two entries
__global__ void wi_fill_kernel(const long long* __restrict__ eptr, const int* __restrict__ eframe, int V, int W, const long long* __restrict__ pair_ptr,
                               unsigned int* __restrict__ pair_key, unsigned int* __restrict__ pair_idx, unsigned int* __restrict__ pair_ei,
                               unsigned int* __restrict__ pair_ej) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= V) return;
  const long long e0 = eptr[a];
  const int k = (int)(eptr[a + 1] - e0);
  long long p = pair_ptr[a];
  for (int i = 0; i < k; i++)
    for (int j = i; j < k; j++) {
      pair_key[p] = (unsigned int)(eframe[e0 + i] * W + eframe[e0 + j]);
      pair_idx[p] = (unsigned int)p;
      pair_ei[p] = (unsigned int)(e0 + i);
      pair_ej[p] = (unsigned int)(e0 + j);
      p++;
    }
}
__global__ void wi_gather_kernel(const unsigned int* __restrict__ order, long long np, const unsigned int* __restrict__ ei, const unsigned int* __restrict__ ej,
                                 unsigned int* __restrict__ sei, unsigned int* __restrict__ sej) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < np) { sei[q] = ei[order[q]]; sej[q] = ej[order[q]]; }
}
__global__ void wi_widen_kernel(const unsigned int* __restrict__ in, long long n, long long* __restrict__ out) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) out[q] = (long long)in[q];
}

// Per-voxel record of what the Hessian sweep needs from the cache, packed: [u0 3 | u1 3 | u2 3 | vbar 3 | s1 s2 1/N sqrt(coe) coe | pad]
// = 144 contiguous bytes.  The pair records of a block hit voxels in no particular order; gathering the 26 cache values from their
// planes cost a 64-byte sector apiece (3 KB of traffic per record with the clusters laid out the same way: the sweep ran at the
// speed of that traffic), the record costs three.  Rebuilt from the planes at the start of every sweep (one coalesced pass, ~10 us
// at 100k voxels): whoever wrote the cache -- a residual sweep, a push with a seeded cache, a restored snapshot -- it is current.
__global__ __launch_bounds__(256) void k3w_vrec_kernel(FactorView fv, int head, int end, double* __restrict__ vrec) {
  const int a = head + blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= end) return;
  const size_t VS = (size_t)fv.VS;
  double* o = vrec + (size_t)a * WIDE_VREC;
  const double invN = fv.aux[2 * VS + a];
#pragma unroll
  for (int k = 0; k < 9; k++) o[k] = fv.eigvec[(size_t)k * VS + a];
#pragma unroll
  for (int k = 0; k < 3; k++) o[9 + k] = fv.merged[(size_t)(6 + k) * VS + a] * invN;
  o[12] = fv.aux[a]; o[13] = fv.aux[VS + a]; o[14] = invN; o[15] = fv.aux[3 * VS + a]; o[16] = fv.coe[a]; o[17] = 0.0;
}

// One wave per TASK = up to WIDE_TASK_RECORDS consecutive pair records of one Hessian block (the lanes take one record each, a
// latency-serial loop of gathers: a block of 5 000 self-pairs on one wave was 79 such rounds and set the length of the whole sweep;
// now the longest wave does 8), then k3w_combine_kernel adds a block's task partials in task order.  A record's two entries belong to the same voxel and to the
// block's two frames, so the lanes compute their rank-3 rows right here (vxm::k3_entry: ~230 fp64 operations per entry against the
// 368 bytes a record gathers -- clusters of both entries + the voxel's cache) instead of reading them back from a row buffer that
// an extra kernel had to fill first: 45 doubles per entry, 180 MB at nnz = 500k, gone, together with that kernel.
__device__ __forceinline__ void k3w_pose(const double* pl, int f, double R[9], double p[3]) {
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int cc = 0; cc < 3; cc++) R[3 * r + cc] = pl[12 * f + 3 * cc + r];
#pragma unroll
  for (int k = 0; k < 3; k++) p[k] = pl[12 * f + 9 + k];
}
__global__ __launch_bounds__(256) void k3w_blocks_kernel(WideView wv, const double* __restrict__ poses, int head, int end, const unsigned int* __restrict__ key_list,
                                                         const WideTask* __restrict__ tasks, int ntasks, const unsigned int* __restrict__ sei,
                                                         const unsigned int* __restrict__ sej, const double* __restrict__ vrec, double* __restrict__ task_partial) {
  __shared__ double pl[12 * WIDE_MAXW];
  const FactorView& fv = wv.fv;
  const int W = fv.W;
  for (int k = threadIdx.x; k < 12 * W; k += blockDim.x) pl[k] = poses[k];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tq = blockIdx.x * 4 + wave;
  if (tq >= ntasks) return;
  const WideTask task = tasks[tq];
  const unsigned int key = key_list[task.key];
  const int fi = (int)(key / (unsigned)W), fj = (int)(key % (unsigned)W);
  const bool diag = fi == fj;
  double s[36];
#pragma unroll
  for (int e = 0; e < 36; e++) s[e] = 0.0;
  double g[6] = {0, 0, 0, 0, 0, 0};
  for (long long q = task.lo + lane; q < task.hi; q += 64) {
    const unsigned int ei = sei[q], ej = sej[q];
    const int a = wv.evoxel[ei];
    if (a < head || a >= end) continue;          // outside the requested voxel range: contributes nothing
    double ci[10], cj[10];
    load_entry(wv.ecl, ei, ci);
    load_entry(wv.ecl, ej, cj);
    if (ci[9] == 0.0 || cj[9] == 0.0) continue;   // an entry without points has all-zero rows and terms
    vxm::VoxelCache vc;
    {
      double t[WIDE_VREC];
      const v2d_t* q2 = reinterpret_cast<const v2d_t*>(vrec + (size_t)a * WIDE_VREC);
#pragma unroll
      for (int k = 0; k < WIDE_VREC / 2; k++) { const v2d_t u = q2[k]; t[2 * k] = u.x; t[2 * k + 1] = u.y; }
#pragma unroll
      for (int k = 0; k < 3; k++) { vc.u0[k] = t[k]; vc.u1[k] = t[3 + k]; vc.u2[k] = t[6 + k]; vc.vbar[k] = t[9 + k]; }
      vc.s1 = t[12]; vc.s2 = t[13]; vc.invN = t[14]; vc.sc = t[15]; vc.coe = t[16];
    }
    double R[9], p[3], ri[3][6], acc[27];
#pragma unroll
    for (int k = 0; k < 27; k++) acc[k] = 0.0;
    k3w_pose(pl, fi, R, p);
    vxm::k3_entry(ci, ci + 6, ci[9], R, p, vc, ri, acc);
    if (diag) {          // self pair of an entry: its gradient and block-diagonal terms ride along
#pragma unroll
      for (int x = 0; x < 6; x++)
#pragma unroll
        for (int y = 0; y < 6; y++) s[6 * x + y] -= ri[0][x] * ri[0][y] + ri[1][x] * ri[1][y] + ri[2][x] * ri[2][y];
#pragma unroll
      for (int d = 0; d < 6; d++) g[d] += acc[d];
#pragma unroll
      for (int x = 0; x < 3; x++)
#pragma unroll
        for (int y = 0; y < 3; y++) {
          const double rr = acc[6 + (x <= y ? sym6(x, y) : sym6(y, x))], tt = acc[21 + (x <= y ? sym6(x, y) : sym6(y, x))], rt = acc[12 + 3 * x + y];
          s[6 * x + y] += rr;
          s[6 * (3 + x) + 3 + y] += tt;
          s[6 * x + 3 + y] += rt;
          s[6 * (3 + y) + x] += rt;
        }
    } else {
      double rj[3][6], accj[27];
#pragma unroll
      for (int k = 0; k < 27; k++) accj[k] = 0.0;   // not read: the compiler drops the terms
      k3w_pose(pl, fj, R, p);
      vxm::k3_entry(cj, cj + 6, cj[9], R, p, vc, rj, accj);
#pragma unroll
      for (int x = 0; x < 6; x++)
#pragma unroll
        for (int y = 0; y < 6; y++) s[6 * x + y] -= ri[0][x] * rj[0][y] + ri[1][x] * rj[1][y] + ri[2][x] * rj[2][y];
    }
  }
  // fixed butterfly over the wave: every lane ends with the same bits
#pragma unroll
  for (int e = 0; e < 36; e++)
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) s[e] += __shfl_xor(s[e], m, 64);
  if (diag) {
#pragma unroll
    for (int d = 0; d < 6; d++)
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) g[d] += __shfl_xor(g[d], m, 64);
  }
  // the task's partial block (and gradient piece): [36 | 6]
  double v = 0.0;
#pragma unroll
  for (int e = 0; e < 36; e++) v = (e == lane) ? s[e] : v;
#pragma unroll
  for (int e = 0; e < 6; e++) v = (36 + e == lane) ? g[e] : v;
  if (lane < WIDE_TASK_OUT) task_partial[(size_t)tq * WIDE_TASK_OUT + lane] = v;
}
// Sum of a block's task partials in task order (fixed), written as block (fi, fj) and its mirror image; diagonal blocks carry the
// gradient piece.  One 64-lane workgroup per key.
__global__ __launch_bounds__(64) void k3w_combine_kernel(const unsigned int* __restrict__ key_list, const int* __restrict__ key_task_ptr, int nkeys,
                                                         const double* __restrict__ task_partial, int W, double* __restrict__ packed) {
  const int kq = blockIdx.x, lane = threadIdx.x;
  if (kq >= nkeys || lane >= WIDE_TASK_OUT) return;
  const int n = 6 * W;
  const unsigned int key = key_list[kq];
  const int fi = (int)(key / (unsigned)W), fj = (int)(key % (unsigned)W);
  const bool diag = fi == fj;
  double v = 0.0;
  for (int t = key_task_ptr[kq]; t < key_task_ptr[kq + 1]; t++) v += task_partial[(size_t)t * WIDE_TASK_OUT + lane];
  if (lane < 36) {
    const int x = lane / 6, y = lane % 6;
    packed[(size_t)(6 * fj + y) * n + 6 * fi + x] = v;            // block (fi, fj) ...
    if (!diag) packed[(size_t)(6 * fi + x) * n + 6 * fj + y] = v;  // ... and its mirror image (voxel_map.hpp:237-239)
  } else if (diag) {
    packed[(size_t)n * n + 6 * fi + (lane - 36)] = v;
  }
}

// residual = sum coe * lambda0 over [head, end): wave partials, summed by sum kernel in fixed order
__global__ __launch_bounds__(64) void k3w_residual_kernel(FactorView fv, int head, int end, double* __restrict__ partial) {
  const int a = head + blockIdx.x * 64 + threadIdx.x;
  double r = a < end ? fv.coe[a] * fv.eigval[a] : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ __launch_bounds__(256) void k3w_residual_sum_kernel(const double* __restrict__ partial, int nparts, double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int k = threadIdx.x; k < nparts; k += 256) s += partial[k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

int launch_k2_wide(const WideView& wv, const double* d_poses, int head, int end, double* d_partial, hipStream_t s) {
  const int nblocks = (end - head + 63) / 64;
  if (nblocks <= 0) return 0;
  k2_wide_kernel<<<dim3(nblocks), dim3(64), 0, s>>>(wv, d_poses, head, end, d_partial);
  return nblocks;
}

// The index's block pool.  Blocks are reused only by THIS index, i.e. behind each other on the factor's one stream: a block released while a
// sweep that reads it is still queued is next written by a kernel of the following build, queued behind that sweep (hipFree's implicit
// device-wide wait is not needed for that, and it stalled the pass's other streams).  Best fit among the free blocks that are not more than
// four times too large; a miss allocates with a quarter of slack, so that the slightly larger index of the next re-voxelisation round fits.
struct WidePool {
  struct B { void* p; size_t cap; };
  std::vector<B> free_blocks, live;
};
static hipError_t pool_alloc(WidePool& P, void** q, size_t bytes) {
  if (bytes < 8) bytes = 8;
  int best = -1;
  for (int k = 0; k < (int)P.free_blocks.size(); k++) {
    const size_t c = P.free_blocks[k].cap;
    if (c >= bytes && c <= 4 * bytes + ((size_t)1 << 20) && (best < 0 || c < P.free_blocks[best].cap)) best = k;
  }
  if (best >= 0) {
    *q = P.free_blocks[best].p;
    P.live.push_back(P.free_blocks[best]);
    P.free_blocks.erase(P.free_blocks.begin() + best);
    return hipSuccess;
  }
  const size_t cap = bytes + bytes / 4;
  hipError_t e = hipMalloc(q, cap);
  if (e != hipSuccess) {   // under memory pressure: give the cached blocks back and try once more
    for (auto& b : P.free_blocks) (void)hipFree(b.p);
    P.free_blocks.clear();
    e = hipMalloc(q, cap);
    if (e != hipSuccess) return e;
  }
  P.live.push_back({*q, cap});
  return hipSuccess;
}
static void pool_release(WidePool& P, void* p) {
  if (!p) return;
  for (size_t k = 0; k < P.live.size(); k++)
    if (P.live[k].p == p) { P.free_blocks.push_back(P.live[k]); P.live.erase(P.live.begin() + (long)k); return; }
  (void)hipFree(p);   // not one of the pool's (never happens: every array of the index comes from pool_alloc)
}
void free_index(WideIndex& wi) {
  void* ptrs[] = {wi.sei, wi.sej, wi.key_list, wi.key_ptr, wi.tasks, wi.key_task_ptr, wi.task_partial, wi.vrec};
  WidePool* pool = wi.pool;
  for (void* q : ptrs) if (q) { if (pool) pool_release(*pool, q); else (void)hipFree(q); }
  wi = WideIndex();
  wi.pool = pool;
}
void destroy_index(WideIndex& wi) {
  free_index(wi);
  if (wi.pool) {
    for (auto& b : wi.pool->free_blocks) (void)hipFree(b.p);
    for (auto& b : wi.pool->live) (void)hipFree(b.p);
    delete wi.pool;
    wi.pool = nullptr;
  }
}

size_t index_bytes(const WideIndex& wi, int W) {
  if (!wi.sei) return 0;
  return (size_t)wi.np * 8 + (size_t)W * W * 4 + (size_t)(wi.nkeys + 1) * 12 + (size_t)wi.ntasks * (sizeof(WideTask) + 8 * WIDE_TASK_OUT) + (size_t)wi.V * WIDE_VREC * 8;
}
size_t store_bytes(const WideStore& st) { return (size_t)st.ES * (80 + 8) + (st.eptr ? ((size_t)st.vcap + 1) * 8 : 0) + st.tmp_cap; }

#define WV(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { *err = hipGetErrorString(e_); return -1; } } while (0)

int build_index(const WideView& wv, int V, WideIndex& wi, hipStream_t s, const char** err) {
  free_index(wi);
  wi.V = V;
  if (V == 0) return 0;
  if (!wi.pool) wi.pool = new WidePool();
  WidePool& pool = *wi.pool;
  // temporaries of the build: back into the pool when the build returns (its last act is a stream synchronisation; on an error path the kernels
  // queued so far may still run -- the blocks stay this index's and are only written again behind them)
  struct Tmp { WidePool& P; std::vector<void*> p; ~Tmp() { for (void* q : p) pool_release(P, q); } } tmp{pool, {}};
  auto talloc = [&](void** q, size_t bytes) { hipError_t e = pool_alloc(pool, q, bytes); if (e == hipSuccess) tmp.p.push_back(*q); return e; };
  auto palloc = [&](void** q, size_t bytes) { return pool_alloc(pool, q, bytes); };
  const unsigned gV = (unsigned)((V + 256) / 256);
  const int W = wv.fv.W;
  long long *pc, *pair_ptr;
  WV(talloc((void**)&pc, sizeof(long long) * (V + 1)));
  WV(talloc((void**)&pair_ptr, sizeof(long long) * (V + 1)));
  wi_paircount_kernel<<<gV, 256, 0, s>>>(wv.eptr, V, pc);
  size_t tb = 0, t1 = 0;
  WV(rocprim::exclusive_scan(nullptr, tb, pc, pair_ptr, 0ll, (size_t)V + 1, rocprim::plus<long long>(), s));
  char* d_temp;
  WV(talloc((void**)&d_temp, tb));
  t1 = tb;
  WV(rocprim::exclusive_scan(d_temp, t1, pc, pair_ptr, 0ll, (size_t)V + 1, rocprim::plus<long long>(), s));
  long long tot = 0;
  WV(hipMemcpyAsync(&tot, pair_ptr + V, sizeof(long long), hipMemcpyDeviceToHost, s));
  WV(hipStreamSynchronize(s));
  wi.nnz = wv.nnz;
  wi.np = tot;
  if (wi.np >= 0xffffffffll || wi.nnz >= 0xffffffffll) { *err = "wide index: more than 2^32 entries or entry pairs"; return -1; }
  if (wi.nnz == 0) return 0;
  WV(palloc((void**)&wi.sei, sizeof(unsigned int) * wi.np));
  WV(palloc((void**)&wi.sej, sizeof(unsigned int) * wi.np));
  unsigned int *pkey, *pidx, *pkey_s, *pidx_s, *pei, *pej, *kcnt, *nruns;
  WV(talloc((void**)&pkey, sizeof(unsigned int) * wi.np)); WV(talloc((void**)&pidx, sizeof(unsigned int) * wi.np));
  WV(talloc((void**)&pkey_s, sizeof(unsigned int) * wi.np)); WV(talloc((void**)&pidx_s, sizeof(unsigned int) * wi.np));
  WV(talloc((void**)&pei, sizeof(unsigned int) * wi.np)); WV(talloc((void**)&pej, sizeof(unsigned int) * wi.np));
  const size_t maxkeys = (size_t)W * W;
  WV(talloc((void**)&kcnt, sizeof(unsigned int) * maxkeys)); WV(talloc((void**)&nruns, sizeof(unsigned int)));
  wi_fill_kernel<<<gV, 256, 0, s>>>(wv.eptr, wv.eframe, V, W, pair_ptr, pkey, pidx, pei, pej);
  int key_bits = 1;
  while ((1u << key_bits) < maxkeys) key_bits++;
  size_t tb2 = 0;
  WV(rocprim::radix_sort_pairs(nullptr, tb2, pkey, pkey_s, pidx, pidx_s, (size_t)wi.np, 0, key_bits, s));
  char* d_temp2;
  WV(talloc((void**)&d_temp2, tb2));
  WV(rocprim::radix_sort_pairs(d_temp2, tb2, pkey, pkey_s, pidx, pidx_s, (size_t)wi.np, 0, key_bits, s));   // stable: voxel order inside a key
  wi_gather_kernel<<<(unsigned)((wi.np + 255) / 256), 256, 0, s>>>(pidx_s, wi.np, pei, pej, wi.sei, wi.sej);
  WV(palloc((void**)&wi.key_list, sizeof(unsigned int) * maxkeys));
  size_t tb3 = 0;
  WV(rocprim::run_length_encode(nullptr, tb3, pkey_s, (size_t)wi.np, wi.key_list, kcnt, nruns, s));
  char* d_temp3;
  WV(talloc((void**)&d_temp3, tb3));
  WV(rocprim::run_length_encode(d_temp3, tb3, pkey_s, (size_t)wi.np, wi.key_list, kcnt, nruns, s));
  unsigned int h_runs = 0;
  WV(hipMemcpyAsync(&h_runs, nruns, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
  WV(hipStreamSynchronize(s));
  wi.nkeys = (int)h_runs;
  WV(palloc((void**)&wi.key_ptr, sizeof(long long) * (wi.nkeys + 1)));
  long long* kc64;
  WV(talloc((void**)&kc64, sizeof(long long) * (wi.nkeys + 1)));
  wi_widen_kernel<<<(unsigned)((wi.nkeys + 255) / 256), 256, 0, s>>>(kcnt, wi.nkeys, kc64);
  size_t tb4 = 0;
  WV(rocprim::exclusive_scan(nullptr, tb4, kc64, wi.key_ptr, 0ll, (size_t)wi.nkeys + 1, rocprim::plus<long long>(), s));
  char* d_temp4;
  WV(talloc((void**)&d_temp4, tb4));
  WV(rocprim::exclusive_scan(d_temp4, tb4, kc64, wi.key_ptr, 0ll, (size_t)wi.nkeys + 1, rocprim::plus<long long>(), s));
  // tasks: the runs cut into pieces of at most WIDE_TASK_RECORDS records (a fixed partition: the sums stay reproducible)
  std::vector<long long> h_ptr((size_t)wi.nkeys + 1);
  WV(hipMemcpyAsync(h_ptr.data(), wi.key_ptr, sizeof(long long) * h_ptr.size(), hipMemcpyDeviceToHost, s));
  WV(hipStreamSynchronize(s));
  std::vector<WideTask> h_tasks;
  std::vector<int> h_ktp((size_t)wi.nkeys + 1, 0);
  for (int k = 0; k < wi.nkeys; k++) {
    h_ktp[k] = (int)h_tasks.size();
    for (long long lo = h_ptr[k]; lo < h_ptr[k + 1]; lo += WIDE_TASK_RECORDS) h_tasks.push_back(WideTask{lo, std::min(lo + WIDE_TASK_RECORDS, h_ptr[k + 1]), k, 0});
  }
  h_ktp[wi.nkeys] = (int)h_tasks.size();
  wi.ntasks = (int)h_tasks.size();
  WV(palloc((void**)&wi.tasks, sizeof(WideTask) * std::max<size_t>(1, h_tasks.size())));
  WV(palloc((void**)&wi.key_task_ptr, sizeof(int) * h_ktp.size()));
  WV(palloc((void**)&wi.task_partial, sizeof(double) * WIDE_TASK_OUT * std::max<size_t>(1, h_tasks.size())));
  WV(palloc((void**)&wi.vrec, sizeof(double) * WIDE_VREC * (size_t)V));
  WV(hipMemcpyAsync(wi.tasks, h_tasks.data(), sizeof(WideTask) * h_tasks.size(), hipMemcpyHostToDevice, s));
  WV(hipMemcpyAsync(wi.key_task_ptr, h_ktp.data(), sizeof(int) * h_ktp.size(), hipMemcpyHostToDevice, s));
  WV(hipStreamSynchronize(s));
  WV(hipGetLastError());
  return 0;
}

void launch_k3_wide(const WideView& wv, const double* d_poses, const WideIndex& wi, int head, int end, double* d_packed, double* d_partial,
                    hipStream_t s) {
  const FactorView& fv = wv.fv;
  const int n = 6 * fv.W;
  (void)hipMemsetAsync(d_packed, 0, ((size_t)n * n + n + 1) * sizeof(double), s);
  if (end <= head || wi.nnz == 0) return;
  k3w_vrec_kernel<<<dim3((unsigned)((end - head + 255) / 256)), dim3(256), 0, s>>>(fv, head, end, wi.vrec);
  k3w_blocks_kernel<<<dim3((unsigned)((wi.ntasks + 3) / 4)), dim3(256), 0, s>>>(wv, d_poses, head, end, wi.key_list, wi.tasks, wi.ntasks, wi.sei, wi.sej, wi.vrec,
                                                                               wi.task_partial);
  k3w_combine_kernel<<<dim3((unsigned)wi.nkeys), dim3(64), 0, s>>>(wi.key_list, wi.key_task_ptr, wi.nkeys, wi.task_partial, fv.W, d_packed);
  const int nparts = (end - head + 63) / 64;
  k3w_residual_kernel<<<dim3(nparts), dim3(64), 0, s>>>(fv, head, end, d_partial);
  k3w_residual_sum_kernel<<<dim3(1), dim3(256), 0, s>>>(d_partial, nparts, d_packed + (size_t)n * n + n);
}


// ------------------------------------------------------------------------------------------------------------------
// The compressed-row cluster store (WideStore, vxba_wide.h)
// ------------------------------------------------------------------------------------------------------------------
// one lane per new voxel: its entries from the caller's compressed rows; frames must rise strictly and stay below W
__global__ void st_append_csr_kernel(const long long* __restrict__ ptr, const int* __restrict__ fr, const double* __restrict__ cl, int n, int W, int v0,
                                     long long e_base, double* __restrict__ ecl, long long ES, long long* __restrict__ eptr, int* __restrict__ eframe,
                                     int* __restrict__ evoxel, int* __restrict__ bad) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const long long p0 = ptr[a], p1 = ptr[a + 1];
  int prev = -1;
  for (long long q = p0; q < p1; q++) {
    const int f = fr[q];
    if (f <= prev || f >= W) { atomicOr(bad, 1); return; }
    prev = f;
    const long long e = e_base + q;
    eframe[e] = f;
    evoxel[e] = v0 + a;
#pragma unroll
    for (int k = 0; k < 10; k++) ecl[(size_t)e * 10 + k] = cl[(size_t)q * 10 + k];
  }
  eptr[v0 + a + 1] = e_base + p1;
}
// dense input: cluster of (voxel a, frame f) at dense[cell * 10], cell = a * W + f (voxel-major) or f * n + a (frame-major)
__device__ __forceinline__ size_t st_cell(int a, int f, int n, int W, int frame_major) { return frame_major ? (size_t)f * n + a : (size_t)a * W + f; }
__global__ void st_count_dense_kernel(const double* __restrict__ dense, int n, int W, int frame_major, long long* __restrict__ cnt) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a > n) return;
  long long k = 0;
  if (a < n)
    for (int f = 0; f < W; f++) k += dense[st_cell(a, f, n, W, frame_major) * 10 + 9] != 0.0 ? 1 : 0;
  cnt[a] = k;   // cnt[n] = 0: the scan leaves the total there
}
__global__ void st_append_dense_kernel(const double* __restrict__ dense, const long long* __restrict__ ptr, int n, int W, int frame_major, int v0, long long e_base,
                                       double* __restrict__ ecl, long long ES, long long* __restrict__ eptr, int* __restrict__ eframe, int* __restrict__ evoxel) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  long long e = e_base + ptr[a];
  for (int f = 0; f < W; f++) {
    const double* c = dense + st_cell(a, f, n, W, frame_major) * 10;
    if (c[9] == 0.0) continue;
    eframe[e] = f;
    evoxel[e] = v0 + a;
#pragma unroll
    for (int k = 0; k < 10; k++) ecl[(size_t)e * 10 + k] = c[k];
    e++;
  }
  eptr[v0 + a + 1] = e;
}
__global__ void st_expand_kernel(const double* __restrict__ ecl, long long ES, const long long* __restrict__ eptr, const int* __restrict__ eframe, int head, int n,
                                 int W, double* __restrict__ dense) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  for (long long e = eptr[head + a]; e < eptr[head + a + 1]; e++) {
    double* o = dense + ((size_t)a * W + eframe[e]) * 10;
#pragma unroll
    for (int k = 0; k < 10; k++) o[k] = ecl[(size_t)e * 10 + k];
  }
}
__global__ void st_count_observed_kernel(const double* __restrict__ npl, long long nnz, unsigned long long* __restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool one = e < nnz && npl[(size_t)e * 10 + 9] != 0.0;
  const unsigned long long m = __ballot(one);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}

void store_free(WideStore& st) {
  void* ptrs[] = {st.ecl, st.eptr, st.eframe, st.evoxel, st.tmp};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  st = WideStore();
}
static int store_tmp(WideStore& st, size_t bytes, hipStream_t s, const char** err) {
  if (bytes <= st.tmp_cap) return 0;
  WV(hipStreamSynchronize(s));
  if (st.tmp) WV(hipFree(st.tmp));
  st.tmp = nullptr; st.tmp_cap = 0;
  WV(hipMalloc((void**)&st.tmp, bytes + bytes / 4));
  st.tmp_cap = bytes + bytes / 4;
  return 0;
}
int store_reserve(WideStore& st, int vcap, long long ecap, int V, hipStream_t s, const char** err) {
  if (vcap > st.vcap || !st.eptr) {
    const int want = std::max(std::max(vcap, 2 * st.vcap), 64);
    long long* np = nullptr;
    WV(hipMalloc((void**)&np, sizeof(long long) * ((size_t)want + 1)));
    if (st.eptr) WV(hipMemcpyAsync(np, st.eptr, sizeof(long long) * ((size_t)V + 1), hipMemcpyDeviceToDevice, s));
    else WV(hipMemsetAsync(np, 0, sizeof(long long), s));
    if (st.eptr) { WV(hipStreamSynchronize(s)); WV(hipFree(st.eptr)); }
    st.eptr = np;
    st.vcap = want;
  }
  if (ecap > st.ES) {
    const long long want = (std::max(std::max(ecap, 2 * st.ES), 1024ll) + 63) / 64 * 64;
    double* ncl = nullptr; int *nf = nullptr, *nv = nullptr;
    WV(hipMalloc((void**)&ncl, sizeof(double) * 10 * (size_t)want));
    WV(hipMalloc((void**)&nf, sizeof(int) * (size_t)want));
    WV(hipMalloc((void**)&nv, sizeof(int) * (size_t)want));
    if (st.nnz > 0) {
      WV(hipMemcpyAsync(ncl, st.ecl, sizeof(double) * 10 * (size_t)st.nnz, hipMemcpyDeviceToDevice, s));
      WV(hipMemcpyAsync(nf, st.eframe, sizeof(int) * (size_t)st.nnz, hipMemcpyDeviceToDevice, s));
      WV(hipMemcpyAsync(nv, st.evoxel, sizeof(int) * (size_t)st.nnz, hipMemcpyDeviceToDevice, s));
    }
    if (st.ecl) { WV(hipStreamSynchronize(s)); WV(hipFree(st.ecl)); WV(hipFree(st.eframe)); WV(hipFree(st.evoxel)); }
    st.ecl = ncl; st.eframe = nf; st.evoxel = nv;
    st.ES = want;
  }
  return 0;
}
void store_append_csr(WideStore& st, int v0, int n, const long long* d_ptr, const int* d_fr, const double* d_cl, long long nnz_new, int W, int* d_bad,
                      hipStream_t s) {
  (void)nnz_new;
  st_append_csr_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(d_ptr, d_fr, d_cl, n, W, v0, st.nnz, st.ecl, st.ES, st.eptr, st.eframe, st.evoxel, d_bad);
}
long long store_append_dense(WideStore& st, int v0, int n, const double* d_dense, int W, int V, hipStream_t s, const char** err, int frame_major) {
  if (n <= 0) return 0;
  size_t tb = 0;
  long long* dummy = nullptr;
  if (rocprim::exclusive_scan(nullptr, tb, dummy, dummy, 0ll, (size_t)n + 1, rocprim::plus<long long>(), s) != hipSuccess) { *err = "scan sizing failed"; return -1; }
  const size_t b_cnt = (sizeof(long long) * ((size_t)n + 1) + 255) / 256 * 256;
  if (store_tmp(st, 2 * b_cnt + tb + 256, s, err)) return -1;
  long long* cnt = (long long*)st.tmp;
  long long* ptr = (long long*)(st.tmp + b_cnt);
  char* d_temp = st.tmp + 2 * b_cnt;
  st_count_dense_kernel<<<(unsigned)((n + 256) / 256), 256, 0, s>>>(d_dense, n, W, frame_major, cnt);
  if (rocprim::exclusive_scan(d_temp, tb, cnt, ptr, 0ll, (size_t)n + 1, rocprim::plus<long long>(), s) != hipSuccess) { *err = "scan failed"; return -1; }
  long long tot = 0;
  if (hipMemcpyAsync(&tot, ptr + n, sizeof(long long), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { *err = "append_dense: D2H failed"; return -1; }
  if (store_reserve(st, v0 + n, st.nnz + tot, V, s, err)) return -1;
  st_append_dense_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(d_dense, ptr, n, W, frame_major, v0, st.nnz, st.ecl, st.ES, st.eptr, st.eframe, st.evoxel);
  return tot;
}
void store_expand(const WideStore& st, int head, int n, int W, double* d_dense, hipStream_t s) {
  if (n <= 0) return;
  (void)hipMemsetAsync(d_dense, 0, sizeof(double) * (size_t)n * W * 10, s);
  st_expand_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(st.ecl, st.ES, st.eptr, st.eframe, head, n, W, d_dense);
}
long long store_count_observed(const WideStore& st, int V, hipStream_t s, const char** err) {
  (void)V;
  if (st.nnz == 0) return 0;
  WideStore& mst = const_cast<WideStore&>(st);
  if (store_tmp(mst, 256, s, err)) return -1;
  unsigned long long* d = (unsigned long long*)mst.tmp;
  unsigned long long h = 0;
  if (hipMemsetAsync(d, 0, sizeof h, s) != hipSuccess) { *err = "count: memset failed"; return -1; }
  st_count_observed_kernel<<<(unsigned)((st.nnz + 255) / 256), 256, 0, s>>>(st.ecl, st.nnz, d);
  if (hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { *err = "count: D2H failed"; return -1; }
  return (long long)h;
}

// ------------------------------------------------------------------------------------------------------------------
// Dense solve of the wide LM step on the device: (H + u D) dxi = -JacT with the gauge rows / columns replaced by identity,
// n = 6W up to 768.  Round 1 took potrf / potrs from hipSOLVER (dlopen on first use: 0.3 s when the libraries were resident, minutes
// off a cold disk, and 2 ms per factorisation at n = 594).  Round 2: a blocked right-looking Cholesky written for this size.
//   * The matrix is stored column-major with ONE EXTRA ROW that holds the right-hand side: treating b^T as row n of the lower
//     triangle makes the panel step compute L^-1 b along the way (forward substitution for free).
//   * Block step k (32 columns): wchol_panel_kernel -- every workgroup factors the 32 x 32 diagonal block for itself in LDS (32
//     pivots; cheaper than a launch and a device-wide hand-over) and forward-substitutes its 256 panel rows, one row per thread in
//     registers, L_kk broadcast from LDS; then wchol_trailing_kernel -- one workgroup per 32 x 32 tile of the lower trailing matrix,
//     A_ij -= L_i L_j^T with both 32 x 32 panels in LDS.  19 steps of two launches at n = 594.
//   * wchol_backsolve_kernel: L^T x = y, blocked, one workgroup; dots over the later rows run down contiguous columns.
// A non-positive pivot sets `info`; the caller then takes the host's pivoted LDL^T for this step (the reference's own solver).
// ------------------------------------------------------------------------------------------------------------------
#ifndef VXBA_WC_NB
#define VXBA_WC_NB 32   // measured at n = 594: 32 -> 0.75 ms per solve, 64 -> 1.2 ms (half the barriers, but the 64-pivot chain and the 3-workgroup panel phase grow faster)
#endif
constexpr int WC_NB = VXBA_WC_NB;      // block size of the wide Cholesky (a multiple of 16, at most 64: one wave factors a diagonal block)
constexpr size_t WC_LDS_BYTES = sizeof(double) * (2 * WC_NB * (WC_NB + 1) + WC_NB + WIDE_MAXW * 6);
// Cholesky factor of a kb x kb diagonal block (kb <= 32) by ONE wave, lane r holding row r of the lower triangle in registers.  Per
// pivot: pivot by v_readlane, 1/sqrt by v_rsq_f64 + two Newton steps, the scaled column to LDS once and back as broadcast reads -- no
// workgroup barrier inside the 32-step chain.  Result: Dout[c][r] = L(r, c) (zeros elsewhere).  A non-positive pivot is reported.
__device__ __forceinline__ double wchol_readlane(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wchol_factor_block(double (&d)[WC_NB], int lane, int kb, int blk_base, double* colbuf, double (*Dout)[WC_NB + 1], int* info, bool report) {
  const int r = lane;
#pragma unroll
  for (int p = 0; p < WC_NB; p++) {
    if (p < kb) {
      double piv = wchol_readlane(d[p], p);
      if (!(piv > 0.0)) { if (report && lane == 0) atomicMax(info, blk_base + p + 1); piv = 1.0; }
      double inv = __builtin_amdgcn_rsq(piv);
      inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
      inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
      double sq = piv * inv;
      sq = fma(fma(-sq, sq, piv), 0.5 * inv, sq);
      const double l = (r == p) ? sq : (r > p ? d[p] * inv : 0.0);
      d[p] = l;
      if (r < WC_NB) colbuf[r] = l;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = p + 1; c < WC_NB; c++) {
        const double lc = colbuf[c];
        d[c] -= (c <= r) ? l * lc : 0.0;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (r < WC_NB) {
#pragma unroll
    for (int c = 0; c < WC_NB; c++) Dout[c][r] = (r < kb && c <= r) ? d[c] : 0.0;
  }
}
// block step at column k0 (kb columns): forward substitution of the rows below the diagonal block against its factor Lkk (column-major
// 32 x 32, in a side buffer: A keeps the unfactored diagonal blocks).  One row per thread in registers, L_kk broadcast from LDS.
// The factorisation's workgroups hand data to each other inside one launch (L2 is per XCD, not coherent across them): everything they
// share goes through agent-scope accesses -- stores written through, loads served from the coherent level -- so that a device-wide
// barrier only has to wait for the store acknowledgements instead of writing back and invalidating L2 (measured: 39 us per barrier
// with __threadfence() on both sides, 41 barriers per solve).
__device__ __forceinline__ double wc_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wc_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wchol_load_lkk(const double* __restrict__ Lkk, double (*D)[WC_NB + 1]) {
  for (int e = threadIdx.x; e < WC_NB * WC_NB; e += blockDim.x) { const int r = e % WC_NB, c = e / WC_NB; D[c][r] = wc_ld(Lkk + e); }
}
// rows k0 + kb + 256 chunk .. of the panel; D[c][r] = L(r, c) of the diagonal block
__device__ __forceinline__ void wchol_panel_rows(double* __restrict__ A, int lda, int nrows, int k0, int kb, int chunk, const double (*D)[WC_NB + 1]) {
  const int r = k0 + kb + chunk * 256 + threadIdx.x;
  if (r < nrows) {
    double l[WC_NB];
#pragma unroll
    for (int c = 0; c < WC_NB; c++) l[c] = c < kb ? wc_ld(A + (size_t)(k0 + c) * lda + r) : 0.0;
#pragma unroll
    for (int c = 0; c < WC_NB; c++) {
      if (c < kb) {
        double sum = l[c];
#pragma unroll
        for (int j = 0; j < c; j++) sum -= l[j] * D[j][c];     // L(c, j)
        l[c] = sum / D[c][c];
      }
    }
#pragma unroll
    for (int c = 0; c < WC_NB; c++) if (c < kb) wc_st(A + (size_t)(k0 + c) * lda + r, l[c]);
  }
}
// A[i][j] -= sum_c L[i][k0 + c] L[j][k0 + c] on the lower tiles of the trailing matrix (rows / columns from k0 + kb).  The workgroup of
// tile (0, 0) -- the NEXT diagonal block, complete after this update -- factors it right away (one wave) into Lkk_next, so that the
// 32-pivot chain runs beside the other tiles' updates instead of heading the next step.
__device__ __forceinline__ void wchol_trailing_tile(double* __restrict__ A, int lda, int nrows, int ncols, int k0, int kb, int I, int J, double* __restrict__ Lkk_next,
                                                    int* __restrict__ info, double (*Li)[WC_NB + 1], double (*Lj)[WC_NB + 1], double* colbuf) {
  const int base = k0 + kb, r0 = base + I * WC_NB, c0 = base + J * WC_NB;
  const int tid = threadIdx.x;
  for (int e = tid; e < WC_NB * WC_NB; e += 256) {
    const int rr = e % WC_NB, k = e / WC_NB;
    Li[rr][k] = (k < kb && r0 + rr < nrows) ? wc_ld(A + (size_t)(k0 + k) * lda + r0 + rr) : 0.0;
    Lj[rr][k] = (k < kb && c0 + rr < nrows) ? wc_ld(A + (size_t)(k0 + k) * lda + c0 + rr) : 0.0;
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;    // 16 x 16 threads, Q x Q outputs each: rows tx + 16 a, columns ty + 16 b
  constexpr int Q = WC_NB / 16;
  double acc[Q][Q];
#pragma unroll
  for (int a = 0; a < Q; a++)
#pragma unroll
    for (int b = 0; b < Q; b++) acc[a][b] = 0.0;
#pragma unroll 8
  for (int k = 0; k < WC_NB; k++) {
    double iv[Q], jv[Q];
#pragma unroll
    for (int a = 0; a < Q; a++) { iv[a] = Li[tx + 16 * a][k]; jv[a] = Lj[ty + 16 * a][k]; }
#pragma unroll
    for (int a = 0; a < Q; a++)
#pragma unroll
      for (int b = 0; b < Q; b++) acc[a][b] += iv[a] * jv[b];
  }
  const bool diag = (I == 0 && J == 0);
  __syncthreads();                            // Li is reused below as the updated diagonal block (tile (0, 0) only)
  auto upd = [&](int rr, int cc, double v) {
    const int r = r0 + rr, c = c0 + cc;
    if (r < nrows && c < ncols && r >= c) {
      const double nv = wc_ld(A + (size_t)c * lda + r) - v;
      wc_st(A + (size_t)c * lda + r, nv);
      if (diag) Li[rr][cc] = nv;              // rows / columns of the next diagonal block
    }
  };
#pragma unroll
  for (int a = 0; a < Q; a++)
#pragma unroll
    for (int b = 0; b < Q; b++) upd(tx + 16 * a, ty + 16 * b, acc[a][b]);
  if (diag) {
    __syncthreads();
    if (tid < 64) {
      const int kbn = ncols - base < WC_NB ? ncols - base : WC_NB;     // size of the next diagonal block
      double d[WC_NB];
#pragma unroll
      for (int c = 0; c < WC_NB; c++) d[c] = (tid < kbn && c <= tid) ? Li[tid][c] : 0.0;
      __builtin_amdgcn_wave_barrier();
      wchol_factor_block(d, tid, kbn, base, colbuf, Lj, info, true);     // Lj[c][r] = L(r, c)
      __builtin_amdgcn_wave_barrier();
      for (int e = tid; e < WC_NB * WC_NB; e += 64) { const int rr = e % WC_NB, c = e / WC_NB; wc_st(Lkk_next + e, Lj[c][rr]); }
    }
  }
  __syncthreads();                            // the tile buffers are free again
}
// L^T x = y with y = row n of the factored matrix; one workgroup of 1024 threads; x -> out[0..n).  Left-looking by blocks, last block
// first: one wave solves the block's own 32 x 32 triangle, then every earlier unknown j subtracts what the block contributes to it,
// sum_r L(k0 + r, j) x_(k0 + r) -- 32 CONTIGUOUS doubles of column j per thread, all loads independent.  (A first version let 32
// threads walk down each column with one load in flight per thread: 173 us per solve.)
__device__ __forceinline__ void wchol_backsolve(const double* __restrict__ A, int lda, int n, const double* __restrict__ Lkk_all, double* __restrict__ x_out, double* x) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < n; i += nt) x[i] = wc_ld(A + (size_t)i * lda + n);
  __syncthreads();
  const int nblk = (n + WC_NB - 1) / WC_NB;
  for (int blk = nblk - 1; blk >= 0; blk--) {
    const int k0 = blk * WC_NB, kb = (n - k0 < WC_NB) ? n - k0 : WC_NB;
    if (tid < 64) {
      const double* Lkk = Lkk_all + (size_t)blk * WC_NB * WC_NB;      // column-major: L(r, c) at [c * 32 + r]
      double mine = tid < kb ? x[k0 + tid] : 0.0;                     // lane c owns x_(k0 + c)
      double lrow[WC_NB];                                             // L(cc, tid) for cc > tid: column tid of the block, below the diagonal
#pragma unroll
      for (int cc = 0; cc < WC_NB; cc++) lrow[cc] = (tid < kb && cc < kb && cc >= tid) ? wc_ld(Lkk + tid * WC_NB + cc) : (cc == tid ? 1.0 : 0.0);
#pragma unroll
      for (int cc = WC_NB - 1; cc >= 0; cc--) {
        if (cc < kb) {
          const double xc = wchol_readlane(mine, cc) / wchol_readlane(lrow[cc], cc);   // x_cc = y_cc / L(cc, cc)
          mine = (tid == cc) ? xc : (tid < cc ? mine - lrow[cc] * xc : mine);
        }
      }
      if (tid < kb) x[k0 + tid] = mine;
    }
    __syncthreads();
    // earlier unknowns: j < k0
    for (int j = tid; j < k0; j += nt) {
      const double* col = A + (size_t)j * lda + k0;
      double sum = 0.0;
#pragma unroll
      for (int r = 0; r < WC_NB; r++) sum += (r < kb) ? wc_ld(col + r) * x[k0 + r] : 0.0;
      x[j] -= sum;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += nt) x_out[i] = x[i];
  __syncthreads();
}
// out[0..n) = dxi, out[n] = q1 = 0.5 dxi . (u D dxi - JacT), out[n+1] = residual1 (packed's last slot)
__device__ __forceinline__ void wide_q1(const double* __restrict__ packed, const double* __restrict__ dxi, const double* __restrict__ dvec, int n, double u,
                                        double* __restrict__ out, double* red) {
  double s = 0.0;
  for (int r = threadIdx.x; r < n; r += 256) {
    const double d = dxi[r];
    out[r] = d;
    const double j = r < 6 ? 0.0 : packed[(size_t)n * n + r];
    s += d * (u * wc_ld(dvec + r) * d - j);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[n] = 0.5 * red[0]; out[n + 1] = packed[(size_t)n * n + n]; }
}

// Device-wide barrier of the persistent solve: a monotone arrival counter (zeroed by the host before the launch); shared data moves
// through agent-scope accesses (wc_ld / wc_st), so arriving only takes the acknowledgement of this wave's stores.  All workgroups of the launch are resident (the grid is at most half the CUs and the sweeps before
// it on the stream have finished); the wait is bounded all the same -- a barrier that gives up sets `info` and every later one falls
// straight through, the host sees info != 0 and takes the step on the CPU.
constexpr unsigned WC_SPIN_LIMIT = 1u << 20;
__device__ __forceinline__ void wchol_grid_barrier(unsigned* counter, unsigned& target, unsigned nwg, int* info, unsigned spin_limit) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's written-through stores are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    target += nwg;
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > spin_limit || __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) {
        __hip_atomic_store(info, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}

// The whole damped step of a wide window in ONE launch: damping + gauge rows, blocked right-looking Cholesky with the right-hand side
// as row n, back substitution, q1.  The same phases as before (they were ~42 launches of a few microseconds each on a GPU that idled
// and clocked down between them: 0.9 of the 1.45 ms of a W = 99 iteration), now separated by device-wide barriers.
__global__ __launch_bounds__(256) void wchol_persistent_kernel(const double* __restrict__ packed, int n, double u, double* __restrict__ A, double* __restrict__ Lkk_all,
                                                               double* __restrict__ dvec, double* __restrict__ xbuf, double* __restrict__ out, int* __restrict__ info,
                                                               unsigned* __restrict__ counter, unsigned spin_limit) {
  extern __shared__ __attribute__((aligned(16))) double wc_lds[];     // two tile buffers, the pivot column, the solution (WC_LDS_BYTES)
  double (*Li)[WC_NB + 1] = reinterpret_cast<double (*)[WC_NB + 1]>(wc_lds);
  double (*Lj)[WC_NB + 1] = reinterpret_cast<double (*)[WC_NB + 1]>(wc_lds + WC_NB * (WC_NB + 1));
  double* colbuf = wc_lds + 2 * WC_NB * (WC_NB + 1);
  double* xs = colbuf + WC_NB;
  const int lda = n + 1, nrows = n + 1;
  const unsigned nwg = gridDim.x, wg = blockIdx.x;
  const int tid = threadIdx.x;
  unsigned target = 0;
  // A = H + u D with the gauge rows, right-hand side in row n
  const long long tot = (long long)n * n + n;
  for (long long t = (long long)wg * 256 + tid; t < tot; t += (long long)nwg * 256) {
    if (t < (long long)n * n) {
      const int r = (int)(t % n), c = (int)(t / n);
      double h = (r < 6 || c < 6) ? ((r == c) ? 1.0 : 0.0) : packed[t];     // gauge fix on frame 0 (voxel_map.hpp:397-400)
      if (r == c) { wc_st(dvec + r, h); h += u * h; }                         // D = diag(H), A = H + u D (:402-403)
      wc_st(A + (size_t)c * lda + r, h);
    } else {
      const int i = (int)(t - (long long)n * n);
      wc_st(A + (size_t)i * lda + n, i < 6 ? 0.0 : -packed[t]);              // row n: the right-hand side
    }
  }
  wchol_grid_barrier(counter, target, nwg, info, spin_limit);
  // first diagonal block (the later ones are factored by the trailing update that completes them)
  if (wg == 0 && tid < 64) {
    const int kb0 = n < WC_NB ? n : WC_NB;
    double d[WC_NB];
#pragma unroll
    for (int c = 0; c < WC_NB; c++) d[c] = (tid < kb0 && c <= tid) ? wc_ld(A + (size_t)c * lda + tid) : 0.0;
    wchol_factor_block(d, tid, kb0, 0, colbuf, Li, info, true);
    __builtin_amdgcn_wave_barrier();
    for (int e = tid; e < WC_NB * WC_NB; e += 64) { const int rr = e % WC_NB, c = e / WC_NB; wc_st(Lkk_all + e, Li[c][rr]); }
  }
  wchol_grid_barrier(counter, target, nwg, info, spin_limit);
  for (int k0 = 0; k0 < n; k0 += WC_NB) {
    const int kb = n - k0 < WC_NB ? n - k0 : WC_NB;
    const int below = nrows - k0 - kb;                       // rows under the diagonal block (>= 1: the right-hand side row)
    double* Lk = Lkk_all + (size_t)(k0 / WC_NB) * WC_NB * WC_NB;
    const int nchunks = (below + 255) / 256;
    if ((int)wg < nchunks) {
      wchol_load_lkk(Lk, Li);
      __syncthreads();
      for (int chunk = wg; chunk < nchunks; chunk += nwg) wchol_panel_rows(A, lda, nrows, k0, kb, chunk, Li);
      __syncthreads();
    }
    wchol_grid_barrier(counter, target, nwg, info, spin_limit);
    const int tr = (below + WC_NB - 1) / WC_NB, tc = (n - k0 - kb + WC_NB - 1) / WC_NB;
    if (tc > 0) {
      for (int t = wg; t < tr * tc; t += nwg) {
        const int I = t / tc, J = t % tc;
        if (J <= I) wchol_trailing_tile(A, lda, nrows, n, k0, kb, I, J, Lk + WC_NB * WC_NB, info, Li, Lj, colbuf);
      }
      wchol_grid_barrier(counter, target, nwg, info, spin_limit);
    }
  }
  if (wg == 0) {
    wchol_backsolve(A, lda, n, Lkk_all, xbuf, xs);
    __threadfence_block();
    wide_q1(packed, xbuf, dvec, n, u, out, &Li[0][0]);
  }
}

struct DenseSolver {
  double *d_A = nullptr, *d_Lkk = nullptr, *d_x = nullptr, *d_dvec = nullptr, *d_out = nullptr, *h_out = nullptr;
  int* d_info = nullptr;
  int* h_info = nullptr;
  unsigned* d_counter = nullptr;
  int n = 0, nwg = 0;
};

void wide_solver_free(DenseSolver*& ds) {
  if (!ds) return;
  void* ptrs[] = {ds->d_A, ds->d_Lkk, ds->d_x, ds->d_dvec, ds->d_out, ds->d_info, ds->d_counter};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  if (ds->h_out) (void)hipHostFree(ds->h_out);
  if (ds->h_info) (void)hipHostFree(ds->h_info);
  delete ds;
  ds = nullptr;
}

// nullptr if the buffers cannot be allocated (caller falls back to the host solve)
size_t wide_solver_bytes(const DenseSolver* ds) {
  if (!ds) return 0;
  const size_t n = (size_t)ds->n;
  return 8 * ((n + 1) * n + n + ((n + WC_NB - 1) / WC_NB) * WC_NB * WC_NB + n + n + 2) + 4;
}
DenseSolver* wide_solver_create(int n, hipStream_t) {
  DenseSolver* ds = new DenseSolver();
  ds->n = n;
  const bool ok = hipMalloc((void**)&ds->d_A, sizeof(double) * (size_t)(n + 1) * n) == hipSuccess && hipMalloc((void**)&ds->d_x, sizeof(double) * n) == hipSuccess &&
                  hipMalloc((void**)&ds->d_Lkk, sizeof(double) * (size_t)((n + WC_NB - 1) / WC_NB) * WC_NB * WC_NB) == hipSuccess &&
                  hipMalloc((void**)&ds->d_dvec, sizeof(double) * n) == hipSuccess && hipMalloc((void**)&ds->d_out, sizeof(double) * (n + 2)) == hipSuccess &&
                  hipMalloc((void**)&ds->d_info, sizeof(int)) == hipSuccess && hipMalloc((void**)&ds->d_counter, 64) == hipSuccess && hipHostMalloc((void**)&ds->h_out, sizeof(double) * (n + 2), hipHostMallocDefault) == hipSuccess &&
                  hipHostMalloc((void**)&ds->h_info, sizeof(int), hipHostMallocDefault) == hipSuccess;
  if (!ok) { wide_solver_free(ds); return nullptr; }
  // dynamic LDS above 64 KB (block sizes over 32) needs the opt-in on THIS device; per solver, i.e. per factor: no process-wide flag
  if (WC_LDS_BYTES > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(&wchol_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WC_LDS_BYTES) != hipSuccess) {
    wide_solver_free(ds);
    return nullptr;
  }
  return ds;
}

// One damped step from the packed buffer on the device.  Host outputs: dxi (n), *q1, *residual1.  Returns 0, or 1 if the
// factorisation met a non-positive pivot / a call failed -- the caller then takes the host path for this step.
int wide_solver_step(DenseSolver* ds, const double* d_packed, double u, hipStream_t s, double* dxi, double* q1, double* residual1, bool debug_give_up) {
  const int n = ds->n;
  if (ds->nwg == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    const int tiles = ((n + WC_NB) / WC_NB) * ((n + WC_NB - 1) / WC_NB) / 2 + 1;     // lower tiles of the first trailing update
    int cap = 128;
    if (const char* e = getenv("VXBA_WIDE_NWG")) cap = std::max(1, atoi(e));     // development: workgroups of the persistent solve
    ds->nwg = std::max(1, std::min(std::min(prop.multiProcessorCount / 2, cap), tiles));
  }
  if (hipMemsetAsync(ds->d_counter, 0, sizeof(unsigned), s) != hipSuccess || hipMemsetAsync(ds->d_info, 0, sizeof(int), s) != hipSuccess) return 1;
  wchol_persistent_kernel<<<dim3((unsigned)ds->nwg), dim3(256), WC_LDS_BYTES, s>>>(d_packed, n, u, ds->d_A, ds->d_Lkk, ds->d_dvec, ds->d_x, ds->d_out, ds->d_info, ds->d_counter,
                                                                                          debug_give_up ? 0u : WC_SPIN_LIMIT);
  if (hipMemcpyAsync(ds->h_info, ds->d_info, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
  if (hipMemcpyAsync(ds->h_out, ds->d_out, sizeof(double) * (n + 2), hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
  if (hipStreamSynchronize(s) != hipSuccess) return 1;
  if (hipGetLastError() != hipSuccess) return 1;
  if (*ds->h_info != 0) return 1;
  std::memcpy(dxi, ds->h_out, sizeof(double) * n);
  *q1 = ds->h_out[n];
  *residual1 = ds->h_out[n + 1];
  for (int r = 0; r < n; r++) if (!(dxi[r] == dxi[r])) return 1;   // NaN guard
  return 0;
}

}  // namespace vxw
