// vxba_map.hip -- the incremental local map resident on the GPU: the reference's `surf_map` / `surf_map_slide`
// of OctoTree nodes with their sliding windows, driven scan by scan the way the local-mapping thread drives it
//     cut_voxel_multi  voxel_map.hpp:1545-1639      OctoTree::push / allocate   :969-1046
//     multi_recut      voxelslam.cpp:1396-1453      OctoTree::recut / fix_divide / subdivide   voxel_map.hpp:1074-1116, 1148-1194
//     tras_opt         voxel_map.hpp:1308-1333      (writes the factor's planes on the device, no host copy of the factor)
//     multi_margi      voxelslam.cpp:1321-1394      OctoTree::margi / plane_update   voxel_map.hpp:1118-1146, 1196-1305
//     ring shift       voxelslam.cpp:1683-1687
// from scratch as flat arrays and kernels -- no pointers, no per-voxel mutex, no host octree:
//
//   node pool (structure of arrays, zero-initialised, bump-allocated with one atomic cursor): layer, state (0 leaf / 1 subdivided),
//     8 child indices (+1, 0 = none), centre, quater_length (float, as upstream), isexist / has_sw / is_plane / last_num / opt_state,
//     pcr_add, pcr_fix, cov_add (9x9), eigen-decomposition, plane record, the W window clusters, and per window slot an index range;
//   roots are found through an open-addressing table of packed voxel coordinates (atomicCAS insert);
//   a node's `sw->points[slot]` is not a copy: scan points are immutable, so each window slot keeps its scan (body point + world
//     covariance as pvec_update left it) plus one permutation of its point indices in which every leaf owns a contiguous range in
//     scan order -- cut_voxel = leaf id per point -> stable radix sort -> one lane per touched leaf continues the leaf's running sums
//     over its range (the accumulator starts from the stored value, so the sums stay bit-identical to sequential push());
//   recut works layer by layer: a judge kernel (eigen-decomposition + plane_judge) marks the leaves that split; eight lanes per
//     splitting leaf (one per octant) then replay [fix points | scan 0 | scan 1 | ...] in the reference's order, each keeping the
//     points of its octant -- children get their sums in the order fix_divide / subdivide push them -- and re-bucket the parent's
//     index ranges by octant;
//   margi is one lane per leaf on clusters; the oldest slot's points move to a per-leaf world-frame fix pool until max_points.
// Arithmetic that decides tree structure or feeds running sums (world point, octant test, cluster push) is written unfused, as the
// reference's x86-64 build computes it; everything else follows vxba_math.hpp.
#include "vxba_wait.hpp"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include <atomic>
#include <type_traits>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/vxba.h"
#include "vxba_internal.h"
#include "vxba_math.hpp"

namespace vxmap {

constexpr int MAXW = 16;
constexpr unsigned long long EMPTY_KEY = ~0ull;
constexpr long long LOC_OFF = 32768;   // root coordinates within +-32768: the node ids of the batch voxeliser

struct Params {
  double voxel_size;
  int max_layer;
  double min_point[4];
  double min_eigen_value;
  double thre[4];
  int max_points, win_size, thread_num;
};

struct PoseArg { double Rp[MAXW * 12]; };   // R column-major | p, window order
struct RingArg { int mp[MAXW]; };

struct Nodes {
  int cap;
  int *layer, *state, *child, *root, *isexist, *has_sw, *is_plane, *last_num, *opt_state, *in_slide, *stamp, *path, *dirty;
  unsigned long long* key;     // root key (x,y,z offset by LOC_OFF, 16 bits each)
  double *center, *pcr_add, *pcr_fix, *cov_add, *eigval, *eigvec, *pl_center, *pl_normal, *pl_radius, *pl_var, *pcrs_local;
  double* jour;               // roots: the journey at the last multi_margi that saw the root in the slide map (OctoTree::jour, voxelslam.cpp:1349)
  float* ql;
  int *pt_start, *pt_count;    // [node][slot]
  long long* fix_start;
  int *fix_count, *fix_cap;
};

struct Counters {   // device block of counters, read back after each stage
  int n_nodes, n_roots, n_touched, n_slide_new, n_split, n_fac, n_removed, err;
  int n_list, pad_;
  int n_split_l[4];
  long long fix_need_l[4];
  long long fix_cursor, fix_need;
};

// The counter block goes to the device as a kernel argument and comes back through host memory the device writes directly (round 4): the two
// blit copies per stage (hipMemcpyAsync H2D / D2H: ~5.5 us of GPU time each, ~18 per scan) become two 64-thread kernels.  The host fills its pinned copy
// with 0xffffffff words before the publish kernel is queued and polls until none is left: no counter ever holds that pattern in a 32-bit half (counts
// and error codes are small non-negative ints, the two cursors are point counts), and a word-wise sentinel needs no ordering between the device's
// stores into host memory (which has none, see the LiDAR-inertial shell).  The stream is in order, so a landed publish also means every kernel
// queued before it has finished -- what the stream query of the copy told the host before.
constexpr int CNT_WORDS = (int)(sizeof(Counters) / 4);
static_assert(sizeof(Counters) % 4 == 0 && CNT_WORDS <= 64, "Counters: one wave publishes it word by word");
__global__ __launch_bounds__(64) void cnt_reset_kernel(Counters* d, Counters v) {
  if (threadIdx.x == 0) *d = v;   // (a lane-dependent index into the argument would go through scratch)
}
__global__ __launch_bounds__(64) void cnt_publish_kernel(const Counters* d, unsigned* h) {
  if (threadIdx.x < CNT_WORDS) h[threadIdx.x] = reinterpret_cast<const unsigned*>(d)[threadIdx.x];
}

// ---- unfused helpers --------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double madd_u(double acc, double a, double b) {
#pragma clang fp contract(off)
  const double p = a * b;
  return acc + p;
}
__device__ __forceinline__ double mul_u(double a, double b) {
#pragma clang fp contract(off)
  const double p = a * b;
  return p;
}
// xx.R * pv.pnt + xx.p (voxel_map.hpp:1100, 1264): (R(r,0) x + R(r,1) y) + R(r,2) z, then + p
__device__ __forceinline__ void to_world(const double* Rp, const double* x, double* w) {
#pragma unroll
  for (int r = 0; r < 3; r++) {
    double s = mul_u(Rp[r], x[0]);
    s = madd_u(s, Rp[3 + r], x[1]);
    s = madd_u(s, Rp[6 + r], x[2]);
    w[r] = s + Rp[9 + r];
  }
}
// PointCluster::push (tools.hpp:326-331) on the packed cluster [Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz N]
__device__ __forceinline__ void cl_push(double* c, const double* x) {
  c[9] += 1.0;
  c[0] = madd_u(c[0], x[0], x[0]); c[1] = madd_u(c[1], x[0], x[1]); c[2] = madd_u(c[2], x[0], x[2]);
  c[3] = madd_u(c[3], x[1], x[1]); c[4] = madd_u(c[4], x[1], x[2]); c[5] = madd_u(c[5], x[2], x[2]);
  c[6] += x[0]; c[7] += x[1]; c[8] += x[2];
}
// cov_add += Bf_var(pv, vec) (voxel_map.hpp:91-106), column-major 9x9; Biup = Bi * var evaluated as the text does
__device__ __forceinline__ void cov_add_point(double* acc, const double* x, const double* V /* column-major 3x3 */) {
  const double Bi[6][3] = {{2 * x[0], 0, 0}, {x[1], x[0], 0}, {x[2], 0, x[0]}, {0, 2 * x[1], 0}, {0, x[2], x[1]}, {0, 0, 2 * x[2]}};
  double Biup[6][3];
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) Biup[r][k] = madd_u(madd_u(mul_u(Bi[r][0], V[3 * k]), Bi[r][1], V[3 * k + 1]), Bi[r][2], V[3 * k + 2]);
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int k = 0; k < 6; k++) acc[9 * k + r] += madd_u(madd_u(mul_u(Biup[r][0], Bi[k][0]), Biup[r][1], Bi[k][1]), Biup[r][2], Bi[k][2]);
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) { acc[9 * (6 + k) + r] += Biup[r][k]; acc[9 * r + 6 + k] += Biup[r][k]; }
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) acc[9 * (6 + k) + 6 + r] += V[3 * k + r];
}
// the float-typed voxel index of cut_voxel (voxel_map.hpp:1553-1560)
__device__ __forceinline__ long long voxel_index(double w, double voxel_size) {
  float loc = (float)(w / voxel_size);
  if (loc < 0.0f) loc = __fsub_rn(loc, 1.0f);
  return (long long)loc;
}
__device__ __forceinline__ int octant_of(const double* w, const double* c) {
  return 4 * (w[0] > c[0] ? 1 : 0) + 2 * (w[1] > c[1] ? 1 : 0) + (w[2] > c[2] ? 1 : 0);
}
__host__ __device__ inline unsigned long long mix64(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
__device__ __forceinline__ void init_child(const Nodes& nd, int c, int parent, int oct) {
  nd.layer[c] = nd.layer[parent] + 1;
  nd.root[c] = nd.root[parent];
  nd.key[c] = nd.key[parent];
  nd.path[c] = nd.path[parent] | (oct << (3 * (2 - nd.layer[parent])));
  const float ql = nd.ql[parent];
  const int xyz[3] = {(oct >> 2) & 1, (oct >> 1) & 1, oct & 1};
#pragma unroll
  for (int k = 0; k < 3; k++) nd.center[3 * (size_t)c + k] = nd.center[3 * (size_t)parent + k] + (double)((float)(2 * xyz[k] - 1) * ql);
  nd.ql[c] = ql / 2;
  nd.opt_state[c] = -1;
}

// ---- cut_voxel ---------------------------------------------------------------------------------------------------------------
// A: root voxel of every point, find-or-insert (voxel_map.hpp:1553-1582)
__global__ void map_roots_kernel(Nodes nd, Params prm, unsigned long long* keys, int* vals, unsigned long long cap_mask, const double* __restrict__ pwld, int n,
                                 int* __restrict__ slot_of_point, Counters* cnt, int serial) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in = i < n;
  bool ok = false, created = false;
  long long x = 0, y = 0, z = 0;
  unsigned long long key = 0, h = 0;
  if (in) {
    x = voxel_index(pwld[3 * i], prm.voxel_size); y = voxel_index(pwld[3 * i + 1], prm.voxel_size); z = voxel_index(pwld[3 * i + 2], prm.voxel_size);
    const long long a = x + LOC_OFF, b = y + LOC_OFF, c = z + LOC_OFF;
    if ((a | b | c) < 0 || a >= 2 * LOC_OFF || b >= 2 * LOC_OFF || c >= 2 * LOC_OFF) cnt->err = 1;
    else {
      ok = true;
      key = ((unsigned long long)a << 32) | ((unsigned long long)b << 16) | (unsigned long long)c;
      h = mix64(key) & cap_mask;
      for (;;) {
        // a scan's 100k points fall into a few thousand root voxels, most of them known: look first, only an empty slot costs an atomic
        const unsigned long long seen = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == key) break;
        if (seen != EMPTY_KEY) { h = (h + 1) & cap_mask; continue; }
        const unsigned long long old = atomicCAS(&keys[h], EMPTY_KEY, key);
        if (old == EMPTY_KEY) { created = true; break; }       // new root (voxel_map.hpp:1571-1581), set up below
        if (old == key) break;
        h = (h + 1) & cap_mask;
      }
    }
  }
  // node slots for the wave's new roots with ONE atomic per counter (a fast-moving sensor opens thousands of roots per scan: one
  // atomic each on the same two addresses, plus a device-wide fence each, was this kernel: 80 us -> 58 without the fences)
  const unsigned long long m = __ballot(created);
  if (m) {
    const int leader = __ffsll((long long)m) - 1, cntm = __popcll(m);
    int base = 0;
    if (lane == leader) { base = atomicAdd(&cnt->n_nodes, cntm); atomicAdd(&cnt->n_roots, cntm); }
    base = __shfl(base, leader);
    if (created) {
      const int idx = base + __popcll(m & ((1ull << lane) - 1ull));
      nd.layer[idx] = 0; nd.root[idx] = idx; nd.key[idx] = key; nd.path[idx] = 0; nd.opt_state[idx] = -1;
      nd.center[3 * (size_t)idx] = (0.5 + (double)x) * prm.voxel_size;
      nd.center[3 * (size_t)idx + 1] = (0.5 + (double)y) * prm.voxel_size;
      nd.center[3 * (size_t)idx + 2] = (0.5 + (double)z) * prm.voxel_size;
      nd.ql[idx] = (float)(prm.voxel_size / 4.0);
      vals[h] = idx + 1;       // nobody reads vals (or the new node) inside this kernel: the next launch sees both
    }
  }
  if (in) slot_of_point[i] = ok ? (int)h : -1;
}
// B: every point walks down from its root: leaf -> stop, subdivided -> octant, no child there -> allocate it (OctoTree::allocate
// voxel_map.hpp:1021-1046).  A lane that finds the child being allocated by another lane reports (parent, octant) for kernel C.
__global__ void map_descend_kernel(Nodes nd, const int* __restrict__ vals, const double* __restrict__ pwld, int n, const int* __restrict__ slot_of_point, int* __restrict__ leaf,
                                   int* __restrict__ pend_parent, Counters* cnt, int serial) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = slot_of_point[i];
  if (s < 0) { leaf[i] = 0x7fffffff; pend_parent[i] = -1; return; }
  int node = vals[s] - 1;
  if (atomicExch(&nd.stamp[node], serial) != serial) atomicAdd(&cnt->n_touched, 1);      // distinct roots this scan touches (:1603-1605)
  if (atomicExch(&nd.in_slide[node], 1) == 0) atomicAdd(&cnt->n_slide_new, 1);         // feat_tem_map (:1566-1567, 1580)
  nd.dirty[node] = 1;      // (round 4, measured: a plain load in front of each exchange changes nothing -- 47.0 against 42.5 us; the bench's scans touch ~90k roots, one point each)
  const double w[3] = {pwld[3 * i], pwld[3 * i + 1], pwld[3 * i + 2]};
  pend_parent[i] = -1;
  while (nd.state[node] != 0) {
    const int oct = octant_of(w, nd.center + 3 * (size_t)node);
    int* slot = &nd.child[8 * (size_t)node + oct];
    int c = atomicCAS(slot, 0, -1);
    if (c == 0) {
      const int idx = atomicAdd(&cnt->n_nodes, 1);
      init_child(nd, idx, node, oct);
      __threadfence();
      atomicExch(slot, idx + 1);
      node = idx;
      break;
    }
    if (c < 0) { pend_parent[i] = node * 8 + oct; node = -1; break; }
    node = c - 1;
  }
  leaf[i] = node;
}
__global__ void map_resolve_kernel(Nodes nd, int n, int* __restrict__ leaf, const int* __restrict__ pend_parent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || pend_parent[i] < 0) return;
  leaf[i] = nd.child[pend_parent[i]] - 1;      // a freshly allocated child is a leaf
}
__global__ void map_iota_kernel(int* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}
// One lane per head of a run of equal leaf ids in the sorted order: OctoTree::push (voxel_map.hpp:969-993) for the run, in scan order.
// (Round 4, measured and rejected: eight lanes per run head with the ~110 running sums dealt out as in map_subdivide_wave_kernel -- 80.7 us
// against 67.8.  On the bench's scans a leaf receives one or two points per scan, so the kernel is the read-modify-write of ~100k node
// records of 808 B each = 162 MB at 2.4 TB/s, not the folding; eight lanes per record only coalesce worse.)
__global__ __launch_bounds__(64) void map_push_kernel(Nodes nd, Params prm, const int* __restrict__ leaf_sorted, const int* __restrict__ perm, int n,
                                                      const double* __restrict__ pnt, const double* __restrict__ var9, const double* __restrict__ pwld, int mord) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int node = leaf_sorted[q];
  if (node == 0x7fffffff || (q > 0 && leaf_sorted[q - 1] == node)) return;
  const int W = prm.win_size;
  const int layer_n = nd.layer[node];   // with the other loads: read behind the 101 stores below it waited for them
  double cl[10], ca[10], acc[81];
  double* g_cl = nd.pcrs_local + ((size_t)node * W + mord) * 10;
  double* g_ca = nd.pcr_add + (size_t)node * 10;
  double* g_acc = nd.cov_add + (size_t)node * 81;
#pragma unroll
  for (int k = 0; k < 10; k++) { cl[k] = g_cl[k]; ca[k] = g_ca[k]; }
#pragma unroll
  for (int k = 0; k < 81; k++) acc[k] = g_acc[k];
  int cntp = 0;
  for (int j = q; j < n && leaf_sorted[j] == node; j++) {
    const int i = perm[j];
    const double x[3] = {pnt[3 * (size_t)i], pnt[3 * (size_t)i + 1], pnt[3 * (size_t)i + 2]};
    const double w[3] = {pwld[3 * (size_t)i], pwld[3 * (size_t)i + 1], pwld[3 * (size_t)i + 2]};
    cl_push(cl, x);
    cl_push(ca, w);
    cov_add_point(acc, w, var9 + 9 * (size_t)i);
    cntp++;
  }
#pragma unroll
  for (int k = 0; k < 10; k++) { g_cl[k] = cl[k]; g_ca[k] = ca[k]; }
#pragma unroll
  for (int k = 0; k < 81; k++) g_acc[k] = acc[k];
  nd.has_sw[node] = 1;
  nd.isexist[node] = 1;
  if (layer_n < prm.max_layer) { nd.pt_start[(size_t)node * W + mord] = q; nd.pt_count[(size_t)node * W + mord] = cntp; }
}
// `iter->second->isexist = true` for roots that already existed (voxel_map.hpp:1565) -- marked through the stamp of this scan
__global__ void map_mark_existing_roots_kernel(Nodes nd, int n_nodes_before, int serial) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_nodes_before && nd.layer[i] == 0 && nd.root[i] == i && nd.stamp[i] == serial) nd.isexist[i] = 1;
}

// ---- recut -------------------------------------------------------------------------------------------------------------------
// OctoTree::recut's leaf branch (voxel_map.hpp:1150-1172) for every leaf of layer L under a root of the slide map
__global__ void map_judge_kernel(Nodes nd, Params prm, int n_bound, int L, int* __restrict__ split_list, Counters* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_bound) return;
  // Everything the decisions below read, requested TOGETHER (round 4): as a chain of short-circuit tests and a read of the cluster behind the store of opt_state
  // this was ~9 dependent round trips per thread.  (i < n_bound <= the arrays' capacity: a node the device has not created yet reads as garbage and is dropped below.)
  const int n_nodes = cnt->n_nodes;          // the previous layer's subdivision may have added nodes the host has not seen yet
  const int lay = nd.layer[i], stt = nd.state[i], rt = nd.root[i], ex = nd.isexist[i], sw = nd.has_sw[i];
  double c[10];
#pragma unroll
  for (int k = 0; k < 10; k++) c[k] = nd.pcr_add[(size_t)i * 10 + k];
  if (i >= n_nodes || lay != L || stt != 0) return;
  if (!nd.in_slide[rt]) return;
  nd.opt_state[i] = -1;
  if (c[9] <= prm.min_point[L]) { nd.is_plane[i] = 0; return; }
  if (!ex || !sw) return;
  double Cm[6], lam[3], U[9];
  vxm::cluster_cov(c, c + 6, c[9], Cm);
  vxm::eig_sym3(Cm, lam, U);
  for (int k = 0; k < 3; k++) nd.eigval[3 * (size_t)i + k] = lam[k];
  for (int col = 0; col < 3; col++)
    for (int row = 0; row < 3; row++) nd.eigvec[9 * (size_t)i + 3 * col + row] = U[3 * row + col];
  const int plane = (lam[0] < prm.min_eigen_value && (lam[0] / lam[2]) < prm.thre[L]) ? 1 : 0;
  nd.is_plane[i] = plane;
  if (plane || L >= prm.max_layer) return;
  const int k = atomicAdd(&cnt->n_split_l[L], 1);
  split_list[k] = i;
  if (nd.pcr_fix[(size_t)i * 10 + 9] != 0.0) atomicAdd((unsigned long long*)&cnt->fix_need_l[L], (unsigned long long)nd.fix_count[i]);
}
struct ScanSlot { const double* pnt; const double* var9; int* perm; int* tmp; int n; };
struct ScanSlots { ScanSlot s[MAXW]; };
// fix_divide + subdivide(0 .. win_count-1) of one splitting leaf (voxel_map.hpp:1074-1116, 1174-1188), one 64-lane workgroup per leaf:
// lane = (octant, part).  Every accumulator of an octant is one
// sequential sum over the leaf's points in the reference's push order -- that is what makes the sums bit-identical -- but the ~100
// accumulators of an octant (two world clusters, the per-scan local cluster, the 9x9 covariance sum) are dealt to eight lanes instead of
// sitting in one, and the classification (gather the point, transform it, find its octant) is done ONCE per point, cooperatively, 128
// points per round through LDS, instead of once per octant lane.  (Until the end of round 2 eight lanes per leaf each replayed every
// point with all the octant's accumulators in one lane: 160 us per call, the longest kernel of the scan cycle; now ~60.)
//   part 0..5  row r = part of the covariance sum's 6x6 block and of its two 6x3 borders (12 accumulators); part 0 also owns the child
//              node, the counts and the stable re-bucketing of the point lists
//   part 6     the 3x3 variance block (9) + the per-scan local cluster (10)
//   part 7     the two world clusters pcr_add / pcr_fix (20)
constexpr int SUB_TILE = 128;
struct SubStage {
  double w[SUB_TILE][3];     // world point (what the octant test and the world-frame sums see)
  double x[SUB_TILE][3];     // body-frame point (the per-scan local cluster)
  double V[SUB_TILE][9];     // its 3x3 variance, column-major
  int pi[SUB_TILE];
  int code[SUB_TILE];
  int out[SUB_TILE];         // a scan's point indices re-bucketed by octant (single-tile scans: straight from pi / code above)
  long long fdst[8];         // fix points: where each octant's region of the pool starts (-1: none)
};
// cl_push on a[OFF .. OFF + 9] of the lane's accumulator array, the offset a template constant (see sub_accumulate)
template <int OFF>
__device__ __forceinline__ void cl_push_at(double (&a)[20], const double* x) {
  a[OFF + 9] += 1.0;
  a[OFF + 0] = madd_u(a[OFF + 0], x[0], x[0]); a[OFF + 1] = madd_u(a[OFF + 1], x[0], x[1]); a[OFF + 2] = madd_u(a[OFF + 2], x[0], x[2]);
  a[OFF + 3] = madd_u(a[OFF + 3], x[1], x[1]); a[OFF + 4] = madd_u(a[OFF + 4], x[1], x[2]); a[OFF + 5] = madd_u(a[OFF + 5], x[2], x[2]);
  a[OFF + 6] += x[0]; a[OFF + 7] += x[1]; a[OFF + 8] += x[2];
}
// The pushes into a[0..9] (pcr_add: lane 7) and into a[10..19] (the per-scan local cluster: lane 6, body-frame point; pcr_fix: lane 7, world point) must not sit at the
// ends of sibling branches: the optimiser sinks the common tail `a[OFF + 8] += x[2]` of two such branches into their successor through a POINTER phi, and the two
// doubles it selects between then live in scratch memory -- a scratch load, a full s_waitcnt and a scratch store per folded point on the one wave that subdivides a
// leaf (round 4, from the IR: `phi ptr addrspace(5)`).  Hence ONE push into the upper half, behind the role branches, with the source selected instead.
__device__ __forceinline__ void sub_accumulate(int part, const double* w, const double* x, const double* V, bool with_local, bool with_fix, double (&a)[20]) {
  if (part < 6) {
    const double Bi[6][3] = {{2 * w[0], 0, 0}, {w[1], w[0], 0}, {w[2], 0, w[0]}, {0, 2 * w[1], 0}, {0, w[2], w[1]}, {0, 0, 2 * w[2]}};
    double br[3] = {0.0, 0.0, 0.0};      // row `part` of Bi
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) br[k] = (r == part) ? Bi[r][k] : br[k];
    double up[3];                         // Biup[part][k]
#pragma unroll
    for (int k = 0; k < 3; k++) up[k] = madd_u(madd_u(mul_u(br[0], V[3 * k]), br[1], V[3 * k + 1]), br[2], V[3 * k + 2]);
#pragma unroll
    for (int k = 0; k < 6; k++) a[k] += madd_u(madd_u(mul_u(up[0], Bi[k][0]), up[1], Bi[k][1]), up[2], Bi[k][2]);   // acc[9 k + part]
#pragma unroll
    for (int k = 0; k < 3; k++) { a[6 + k] += up[k]; a[9 + k] += up[k]; }      // acc[9 (6 + k) + part], acc[9 part + 6 + k]
  } else if (part == 6) {
#pragma unroll
    for (int k = 0; k < 9; k++) a[k] += V[k];                                  // acc[9 (6 + k/3) + 6 + k%3] += V[3 (k/3) + k%3]
  } else {
    cl_push_at<0>(a, w);                                                       // pcr_add
  }
  if ((part == 6 && with_local) || (part == 7 && with_fix)) cl_push_at<10>(a, part == 6 ? x : w);   // lane 6: this scan's local cluster; lane 7: pcr_fix
}
__global__ __launch_bounds__(64) void map_subdivide_wave_kernel(Nodes nd, Params prm, const int* __restrict__ split_list, int n_split, int win_count, PoseArg poses,
                                                                RingArg ring, ScanSlots scans, double* __restrict__ fix_pnt, double* __restrict__ fix_var, Counters* cnt) {
  __shared__ SubStage st;
  const int node = split_list[blockIdx.x];
  const int lane = threadIdx.x, oct = lane >> 3, part = lane & 7;
  const int W = prm.win_size;
  const int L = nd.layer[node];
  const bool keep_pts = (L + 1) < prm.max_layer;
  const double ctr[3] = {nd.center[3 * (size_t)node], nd.center[3 * (size_t)node + 1], nd.center[3 * (size_t)node + 2]};
  int child = -1;                 // meaningful in part 0 until it is broadcast
  double a[20];
#pragma unroll
  for (int k = 0; k < 20; k++) a[k] = 0.0;
  // (Measured and rejected, round 3: a pre-pass that classifies the node's points first, so that it takes its children and their fix
  // regions with one read-modify-write of each counter instead of up to sixteen: 74.4 against 72.8 us -- the folding, not the atomics.)
  // the parent's fields a child is made from, fetched ONCE (round 4): init_child read them from the node arrays between its stores -- eight load -> wait -> store
  // round trips -- every time an octant met its first point, i.e. up to eight times per leaf, in the middle of the fold: most of this kernel's time
  const int p_root = nd.root[node];
  const unsigned long long p_key = nd.key[node];
  const int p_path = nd.path[node];
  const float p_ql = nd.ql[node];
  auto get_child = [&]() {
    if (part == 0 && child < 0) {
      child = atomicAdd(&cnt->n_nodes, 1);
      const int c = child;
      nd.layer[c] = L + 1;
      nd.root[c] = p_root;
      nd.key[c] = p_key;
      nd.path[c] = p_path | (oct << (3 * (2 - L)));
      const int xyz[3] = {(oct >> 2) & 1, (oct >> 1) & 1, oct & 1};
#pragma unroll
      for (int k = 0; k < 3; k++) nd.center[3 * (size_t)c + k] = ctr[k] + (double)((float)(2 * xyz[k] - 1) * p_ql);
      nd.ql[c] = p_ql / 2;
      nd.opt_state[c] = -1;
      nd.child[8 * (size_t)node + oct] = child + 1;
    }
  };
  // one round: `count` staged points, every lane folds the ones of its octant, in order
  auto fold = [&](int count, bool with_local, bool with_fix, int& mine) {
    for (int j = 0; j < count; j++) {
      if (st.code[j] != oct) continue;
      get_child();
      sub_accumulate(part, st.w[j], st.x[j], st.V[j], with_local, with_fix, a);
      mine++;
    }
  };
  // fix_divide (voxel_map.hpp:1074-1094): only when pcr_fix.N != 0 (:1174)
  const bool has_fix = nd.pcr_fix[(size_t)node * 10 + 9] != 0.0;
  if (has_fix) {
    const long long f0 = nd.fix_start[node];
    const int fc = nd.fix_count[node];
    int nfix = 0;
    for (int t0 = 0; t0 < fc; t0 += SUB_TILE) {
      const int cntp = fc - t0 < SUB_TILE ? fc - t0 : SUB_TILE;
      for (int j = lane; j < cntp; j += 64) {
        const double* x = fix_pnt + 3 * (size_t)(f0 + t0 + j);
        const double* v = fix_var + 9 * (size_t)(f0 + t0 + j);
#pragma unroll
        for (int e = 0; e < 3; e++) { st.w[j][e] = x[e]; st.x[j][e] = x[e]; }
#pragma unroll
        for (int e = 0; e < 9; e++) st.V[j][e] = v[e];
        st.code[j] = octant_of(x, ctr);
      }
      __syncthreads();
      fold(cntp, false, true, nfix);
      __syncthreads();
    }
    if (keep_pts && fc <= SUB_TILE) {
      // One tile: st.code / st.w / st.V still hold the node's fix points, so every lane moves one point (its place = its octant's region + the number of
      // earlier points of that octant: the order the serial copy below produces).  Round 4: that serial copy -- lane `part 0` of an octant walking
      // all fc points with a dependent global load each and twelve load -> store pairs per kept point -- WAS the kernel: 15-35 leaves split per
      // scan, one wave each, ~70 us of dependent round trips.
      long long dst = -1;
      if (part == 0 && nfix > 0) dst = (long long)atomicAdd((unsigned long long*)&cnt->fix_cursor, (unsigned long long)nfix);
      if (part == 0) st.fdst[oct] = dst;
      __syncthreads();
      for (int j = lane; j < fc; j += 64) {
        const int o = st.code[j];
        int rank = 0;
        for (int q = 0; q < j; q++) rank += (st.code[q] == o) ? 1 : 0;
        const size_t d = (size_t)(st.fdst[o] + rank);
#pragma unroll
        for (int e = 0; e < 3; e++) fix_pnt[3 * d + e] = st.w[j][e];
#pragma unroll
        for (int e = 0; e < 9; e++) fix_var[9 * d + e] = st.V[j][e];
      }
      if (part == 0 && nfix > 0) { nd.fix_start[child] = dst; nd.fix_count[child] = nfix; nd.fix_cap[child] = nfix; }
      __syncthreads();
    } else if (part == 0 && nfix > 0 && keep_pts) {      // push_fix keeps the point when layer < max_layer (voxel_map.hpp:998-999)
      const long long dst = (long long)atomicAdd((unsigned long long*)&cnt->fix_cursor, (unsigned long long)nfix);
      int k = 0;
      for (int j = 0; j < fc; j++) {
        const double* x = fix_pnt + 3 * (size_t)(f0 + j);
        if (octant_of(x, ctr) != oct) continue;
        for (int e = 0; e < 3; e++) fix_pnt[3 * (size_t)(dst + k) + e] = x[e];
        for (int e = 0; e < 9; e++) fix_var[9 * (size_t)(dst + k) + e] = fix_var[9 * (size_t)(f0 + j) + e];
        k++;
      }
      nd.fix_start[child] = dst; nd.fix_count[child] = nfix; nd.fix_cap[child] = nfix;
    }
  }
  // subdivide(i) for the window's scans in order (voxel_map.hpp:1096-1116)
  // Round 4: a wave walks the window's scans one after the other and every global round trip of a scan is on its chain (~1 us each with one wave per
  // split leaf and nothing else to run): the point ranges of all W slots are fetched once, up front, and a scan that fits one tile (the rule: a
  // leaf holds a few points per scan) is re-bucketed out of the tile it has just folded -- no second pass over perm / pnt, no trip through tmp.
  int my_p0 = 0, my_pc = 0;
  if (lane < W) { my_p0 = nd.pt_start[(size_t)node * W + lane]; my_pc = nd.pt_count[(size_t)node * W + lane]; }
  for (int i = 0; i < win_count; i++) {
    int slot = 0;     // ring.mp[i] by a select chain: the run-time index into the kernel argument went through scratch memory -- a memory round trip at the head of every scan
#pragma unroll
    for (int q = 0; q < MAXW; q++) slot = (i == q) ? ring.mp[q] : slot;
    const ScanSlot sc = scans.s[slot];
    const int p0 = __shfl(my_p0, slot);
    const int pc = __shfl(my_pc, slot);
    const double* Rp = poses.Rp + 12 * i;
    if (part == 6) {
#pragma unroll
      for (int k = 10; k < 20; k++) a[k] = 0.0;      // this scan's local cluster
    }
    int mine = 0;
    for (int t0 = 0; t0 < pc; t0 += SUB_TILE) {
      const int cntp = pc - t0 < SUB_TILE ? pc - t0 : SUB_TILE;
      for (int j = lane; j < cntp; j += 64) {
        const int pi = sc.perm[p0 + t0 + j];
        const double* x = sc.pnt + 3 * (size_t)pi;
        const double* v = sc.var9 + 9 * (size_t)pi;
        double xb[3] = {x[0], x[1], x[2]}, w[3];
        to_world(Rp, xb, w);
#pragma unroll
        for (int e = 0; e < 3; e++) { st.w[j][e] = w[e]; st.x[j][e] = xb[e]; }
#pragma unroll
        for (int e = 0; e < 9; e++) st.V[j][e] = v[e];
        st.pi[j] = pi;
        st.code[j] = octant_of(w, ctr);
      }
      __syncthreads();
      fold(cntp, true, false, mine);
      __syncthreads();
    }
    const int child_o = __shfl(child, lane & ~7);            // the octant's node (allocated by part 0 at its first point)
    if (part == 6 && mine > 0) {
      double* g = nd.pcrs_local + ((size_t)child_o * W + slot) * 10;
#pragma unroll
      for (int k = 0; k < 10; k++) g[k] = a[10 + k];
    }
    if (keep_pts) {
      // stable re-bucketing of the parent's index range by octant: exclusive prefix of `mine` over the eight octants
      int before = 0;
#pragma unroll
      for (int o = 0; o < 8; o++) {
        const int v = __shfl(mine, 8 * o);
        if (o < oct) before += v;
      }
      int k = 0;
      if (pc <= SUB_TILE) {
        // st.pi / st.code still hold this scan's points (nothing has touched the tile since the fold's barrier)
        if (part == 0 && mine > 0) {
          for (int j = 0; j < pc; j++)
            if (st.code[j] == oct) { st.out[before + k] = st.pi[j]; k++; }
          nd.pt_start[(size_t)child * W + slot] = p0 + before;
          nd.pt_count[(size_t)child * W + slot] = mine;
        }
        __syncthreads();
        for (int j = lane; j < pc; j += 64) sc.perm[p0 + j] = st.out[j];
        __syncthreads();
        continue;
      }
      for (int t0 = 0; t0 < pc; t0 += SUB_TILE) {
        const int cntp = pc - t0 < SUB_TILE ? pc - t0 : SUB_TILE;
        for (int j = lane; j < cntp; j += 64) {
          const int pi = sc.perm[p0 + t0 + j];
          const double* x = sc.pnt + 3 * (size_t)pi;
          double xb[3] = {x[0], x[1], x[2]}, w[3];
          to_world(Rp, xb, w);
          st.pi[j] = pi;
          st.code[j] = octant_of(w, ctr);
        }
        __syncthreads();
        if (part == 0 && mine > 0)
          for (int j = 0; j < cntp; j++)
            if (st.code[j] == oct) { sc.tmp[p0 + before + k] = st.pi[j]; k++; }
        __syncthreads();
      }
      if (part == 0 && mine > 0) {
        nd.pt_start[(size_t)child * W + slot] = p0 + before;
        nd.pt_count[(size_t)child * W + slot] = mine;
      }
      __syncthreads();
      for (int j = lane; j < pc; j += 64) sc.perm[p0 + j] = sc.tmp[p0 + j];
      __syncthreads();
    }
  }
  const int child_o = __shfl(child, lane & ~7);
  if (child_o >= 0) {
    double* g_acc = nd.cov_add + (size_t)child_o * 81;
    if (part < 6) {
#pragma unroll
      for (int k = 0; k < 6; k++) g_acc[9 * k + part] = a[k];
#pragma unroll
      for (int k = 0; k < 3; k++) { g_acc[9 * (6 + k) + part] = a[6 + k]; g_acc[9 * part + 6 + k] = a[9 + k]; }
    } else if (part == 6) {
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int r = 0; r < 3; r++) g_acc[9 * (6 + k) + 6 + r] = a[3 * k + r];
    } else {
      double* g_ca = nd.pcr_add + (size_t)child_o * 10;
      double* g_cf = nd.pcr_fix + (size_t)child_o * 10;
#pragma unroll
      for (int k = 0; k < 10; k++) { g_ca[k] = a[k]; g_cf[k] = a[10 + k]; }
      if (a[9] != a[19]) { nd.has_sw[child_o] = 1; nd.isexist[child_o] = 1; }     // push() opened a window; push_fix alone does not
    }
  }
  __syncthreads();
  if (lane == 0) {      // sw->clear(); sws.push_back(sw); sw = nullptr; octo_state = 1  (voxel_map.hpp:1184-1187)
    for (int s2 = 0; s2 < W; s2++) {
      nd.pt_count[(size_t)node * W + s2] = 0;
      for (int k = 0; k < 10; k++) nd.pcrs_local[((size_t)node * W + s2) * 10 + k] = 0.0;
    }
    if (has_fix) { nd.fix_count[node] = 0; nd.fix_cap[node] = 0; }
    nd.has_sw[node] = 0;
    nd.state[node] = 1;
  }
}
// tras_opt's filter (voxel_map.hpp:1312-1314): candidates with their node ids, to be ordered by id
__global__ void map_factor_flag_kernel(Nodes nd, int n_bound, unsigned long long* __restrict__ ids, int* __restrict__ nodes, Counters* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_bound || i >= cnt->n_nodes || nd.state[i] != 0 || !nd.in_slide[nd.root[i]]) return;
  if (!(nd.isexist[i] && nd.is_plane[i] && nd.has_sw[i])) return;
  if (nd.eigval[3 * (size_t)i] / nd.eigval[3 * (size_t)i + 1] > 0.12) return;
  const int k = atomicAdd(&cnt->n_fac, 1);
  ids[k] = (nd.key[i] << 16) | ((unsigned long long)nd.path[i] << 7) | (unsigned long long)nd.layer[i];
  nodes[k] = i;
}
// pcrs[i] = sw->pcrs_local[mp[i]]; push_voxel(pcrs, pcr_fix, 1, eig_value, eig_vector, pcr_add)  (voxel_map.hpp:1316-1321)
// One WORKGROUP per factor voxel, one thread per copied double (10 W cluster entries + fix 10 + merged 10 + eigenvalues 3 + eigenvectors 9): every
// thread does ONE load and ONE store.  (Round 4, from the ISA: as one thread per voxel with `out[k] = nd.x[k]` loops the 133 copies were ~25
// dependent load -> wait -> store round trips -- the node arrays may alias the outputs for all the compiler knows -- and with ~17k
// voxels = 270 waves on 1024 SIMDs nothing hid them: 33 us.)
__global__ __launch_bounds__(192) void map_factor_gather_kernel(Nodes nd, int W, RingArg ring, const int* __restrict__ nodes, int n, double* __restrict__ clusters, double* __restrict__ fix,
                                                                double* __restrict__ coe, double* __restrict__ eigval, double* __restrict__ eigvec, double* __restrict__ merged) {
  const int a = blockIdx.x;
  if (a >= n) return;
  const int i = nodes[a];
  const int per = 10 * W + 32;
  for (int t = threadIdx.x; t < per; t += blockDim.x) {
    if (t < 10 * W) {
      const int s = t / 10, k = t - 10 * s;
      int slot = 0;     // ring.mp[s] by a select chain: a lane-dependent index into a kernel argument would go through scratch
#pragma unroll
      for (int q = 0; q < MAXW; q++) slot = (s == q) ? ring.mp[q] : slot;
      clusters[((size_t)a * W + s) * 10 + k] = nd.pcrs_local[((size_t)i * W + slot) * 10 + k];
    } else {
      const int u = t - 10 * W;
      if (u < 10) fix[10 * (size_t)a + u] = nd.pcr_fix[10 * (size_t)i + u];
      else if (u < 20) merged[10 * (size_t)a + (u - 10)] = nd.pcr_add[10 * (size_t)i + (u - 10)];
      else if (u < 23) eigval[3 * (size_t)a + (u - 20)] = nd.eigval[3 * (size_t)i + (u - 20)];
      else eigvec[9 * (size_t)a + (u - 23)] = nd.eigvec[9 * (size_t)i + (u - 23)];
    }
  }
  if (threadIdx.x == 0) { nd.opt_state[i] = a; coe[a] = 1.0; }
}

// ---- margi -------------------------------------------------------------------------------------------------------------------
// PointCluster::transform (tools.hpp:357-363) on packed clusters, pose R column-major | p
__device__ __forceinline__ void cl_transform(const double* s, const double* Rp, double* o) {
  const double N = s[9];
  const double P[9] = {s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5]};   // symmetric, row r col c = P[3r+c]
  double Rv[3], RP[9], out[9];
#pragma unroll
  for (int r = 0; r < 3; r++) Rv[r] = Rp[r] * s[6] + Rp[3 + r] * s[7] + Rp[6 + r] * s[8];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) RP[3 * r + c] = Rp[r] * P[c] + Rp[3 + r] * P[3 + c] + Rp[6 + r] * P[6 + c];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double rprt = RP[3 * r] * Rp[c] + RP[3 * r + 1] * Rp[3 + c] + RP[3 * r + 2] * Rp[6 + c];
      out[3 * r + c] = ((rprt + Rv[r] * Rp[9 + c]) + Rv[c] * Rp[9 + r]) + (N * Rp[9 + r]) * Rp[9 + c];
    }
  o[0] = out[0]; o[1] = out[1]; o[2] = out[2]; o[3] = out[4]; o[4] = out[5]; o[5] = out[8];
#pragma unroll
  for (int r = 0; r < 3; r++) o[6 + r] = Rv[r] + N * Rp[9 + r];
  o[9] = N;
}
// OctoTree::plane_update (voxel_map.hpp:1118-1146) of node i
// clv: the node's merged cluster (pcr_add), Uv / lamv: its eigenvectors (as stored: 3 * column + row) / eigenvalues -- the caller has just written them and
// hands over the registers (read back from the node arrays they would wait for those stores)
__device__ void plane_update_node(const Nodes& nd, int i, const double* clv, const double* Uv, const double* lamv) {
  // Inputs into registers first, outputs out last (round 4): the node arrays are plain pointers in a struct, so a load behind a store of this function
  // waits for that store -- the 36 entries of plane_var used to go out between the loads of cov_add, one round trip each.  Same expressions, same order.
  const double* CA = nd.cov_add + 81 * (size_t)i;
  double ca9[9];
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int r = 0; r < 3; r++) ca9[3 * q + r] = CA[9 * (6 + q) + 6 + r];
  const double N = clv[9];
  const double nv = 1.0 / N;
  const double c[3] = {clv[6] / N, clv[7] / N, clv[8] / N};
  const double* U = Uv;
  const double* lam = lamv;
  double u_c[3][9];
  for (int r = 0; r < 3; r++)
    for (int q = 0; q < 9; q++) u_c[r][q] = 0.0;
  const double* ul = U;
  for (int k = 1; k < 3; k++) {
    const double* uk = U + 3 * k;
    const double kc = uk[0] * c[0] + uk[1] * c[1] + uk[2] * c[2], lc = ul[0] * c[0] + ul[1] * c[1] + ul[2] * c[2];
    const double fkl[9] = {uk[0] * ul[0], uk[1] * ul[0] + uk[0] * ul[1], uk[2] * ul[0] + uk[0] * ul[2], uk[1] * ul[1], uk[1] * ul[2] + uk[2] * ul[1], uk[2] * ul[2],
                           -(kc * ul[0] + lc * uk[0]), -(kc * ul[1] + lc * uk[1]), -(kc * ul[2] + lc * uk[2])};
    const double sc = nv / (lam[0] - lam[k]);
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 9; q++) u_c[r][q] += sc * uk[r] * fkl[q];
  }
  double Jc[3][9];
  for (int r = 0; r < 3; r++)
    for (int q = 0; q < 9; q++) {
      double t = 0.0;
      for (int k = 0; k < 9; k++) t += u_c[r][k] * CA[9 * q + k];
      Jc[r][q] = t;
    }
  double Pv[36];
  for (int r = 0; r < 3; r++)
    for (int q = 0; q < 3; q++) {
      double t = 0.0;
      for (int k = 0; k < 9; k++) t += Jc[r][k] * u_c[q][k];
      Pv[6 * q + r] = t;
      const double jn = nv * Jc[r][6 + q];
      Pv[6 * (3 + q) + r] = jn;
      Pv[6 * r + 3 + q] = jn;
      Pv[6 * (3 + q) + 3 + r] = nv * nv * ca9[3 * q + r];
    }
  double* P = nd.pl_var + 36 * (size_t)i;
#pragma unroll
  for (int k = 0; k < 36; k++) P[k] = Pv[k];
  for (int k = 0; k < 3; k++) { nd.pl_center[3 * (size_t)i + k] = c[k]; nd.pl_normal[3 * (size_t)i + k] = ul[k]; }
  nd.pl_radius[i] = (double)(float)lam[2];
}
// OctoTree::margi's leaf branch with mgsize = 1 (voxel_map.hpp:1198-1290), one lane per leaf.  Pass 0 does the cluster work and
// sizes the fix-pool growth; pass 1 (after the host made room) moves the oldest scan's points into the pool and clears the slot.
__global__ __launch_bounds__(256) void map_margi_kernel(Nodes nd, Params prm, int n_nodes, int win_count, PoseArg poses, RingArg ring, const double* __restrict__ f_eigval,   // (without the bound: 128 registers, 129 spilled)
                                 const double* __restrict__ f_eigvec, const double* __restrict__ f_merged, int f_VS, int f_V, int* __restrict__ work, Counters* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  work[i] = 0;
  if (nd.state[i] != 0 || !nd.in_slide[nd.root[i]]) return;
  if (!nd.isexist[i] || !nd.has_sw[i]) return;
  const int W = prm.win_size;
  const int m0 = ring.mp[0];
  double* add = nd.pcr_add + 10 * (size_t)i;
  double* fixc = nd.pcr_fix + 10 * (size_t)i;
  // Round 4: the node's scalars and the two clusters this thread reads AND writes live in registers from here on (addv / fixv), every group of
  // loads is issued before the stores that follow it -- as `add[k] = f_merged[..]` loops and re-reads of add[9] / fixc[9] / last_num behind stores
  // the kernel was a chain of ~60 load -> wait -> store round trips per node (the arrays may alias, for all the compiler knows).  Same arithmetic.
  const int is_plane = nd.is_plane[i];
  const int last_num = nd.last_num[i];
  const int pc0 = nd.pt_count[(size_t)i * W + m0];
  const int fix_count = nd.fix_count[i], fix_cap = nd.fix_cap[i];
  const int os = nd.opt_state[i];
  double addv[10], fixv[10];
#pragma unroll
  for (int k = 0; k < 10; k++) fixv[k] = fixc[k];
  double w0[10];
  for (int k = 0; k < 10; k++) w0[k] = 0.0;
  if (os >= f_V) { cnt->err = 2; return; }
  double ev[3] = {0.0, 0.0, 0.0}, evec[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // the node's eigen-pairs as this thread leaves them (defined whenever plane_update_node runs)
  if (os >= 0) {                                     // adopt the optimiser's cache (:1217-1229)
    double l0v[10];
#pragma unroll
    for (int k = 0; k < 10; k++) addv[k] = f_merged[(size_t)k * f_VS + os];
#pragma unroll
    for (int k = 0; k < 3; k++) ev[k] = f_eigval[(size_t)k * f_VS + os];
#pragma unroll
    for (int k = 0; k < 9; k++) evec[k] = f_eigvec[(size_t)k * f_VS + os];
    const double* l0 = nd.pcrs_local + ((size_t)i * W + m0) * 10;
#pragma unroll
    for (int k = 0; k < 10; k++) l0v[k] = l0[k];
#pragma unroll
    for (int k = 0; k < 10; k++) add[k] = addv[k];
#pragma unroll
    for (int k = 0; k < 3; k++) nd.eigval[3 * (size_t)i + k] = ev[k];
#pragma unroll
    for (int k = 0; k < 9; k++) nd.eigvec[9 * (size_t)i + k] = evec[k];
    nd.opt_state[i] = -1;
    if (l0v[9] != 0.0) cl_transform(l0v, poses.Rp, w0);
  } else {                                           // :1230-1247
    double s[10];
    for (int k = 0; k < 10; k++) s[k] = fixv[k];
    for (int j = 0; j < win_count; j++) {
      const double* lj = nd.pcrs_local + ((size_t)i * W + ring.mp[j]) * 10;
      if (lj[9] == 0.0) continue;
      double wj[10];
      cl_transform(lj, poses.Rp + 12 * j, wj);
      for (int k = 0; k < 10; k++) s[k] += wj[k];
      if (j == 0) for (int k = 0; k < 10; k++) w0[k] = wj[k];
    }
    for (int k = 0; k < 10; k++) { addv[k] = s[k]; add[k] = s[k]; }
    if (is_plane) {
      double Cm[6], lam[3], U[9];
      vxm::cluster_cov(s, s + 6, s[9], Cm);
      vxm::eig_sym3(Cm, lam, U);
      for (int k = 0; k < 3; k++) { ev[k] = lam[k]; nd.eigval[3 * (size_t)i + k] = lam[k]; }
      for (int col = 0; col < 3; col++)
        for (int row = 0; row < 3; row++) { evec[3 * col + row] = U[3 * row + col]; nd.eigvec[9 * (size_t)i + 3 * col + row] = U[3 * row + col]; }
    }
  }
  const int Nadd = (int)addv[9], Nfix = (int)fixv[9];
  if (Nfix < prm.max_points && is_plane)
    if (Nadd - last_num >= 5 || last_num <= 10) {
      plane_update_node(nd, i, addv, evec, ev);
      nd.last_num[i] = Nadd;
    }
  int wk = 4;                                        // bit 2: the slot is cleared in pass 1
  if (Nfix < prm.max_points) {
    if (w0[9] != 0.0) {
      for (int k = 0; k < 10; k++) { fixv[k] += w0[k]; fixc[k] = fixv[k]; }
      if (pc0 > 0) {
        wk |= 1;                                     // append the slot's points to the fix pool
        if (fix_count + pc0 > fix_cap) {
          const int ncap = 2 * (fix_count + pc0);
          atomicAdd((unsigned long long*)&cnt->fix_need, (unsigned long long)ncap);
          wk |= 2;                                   // needs a new region
        }
      }
    }
  } else {
    if (w0[9] != 0.0) for (int k = 0; k < 10; k++) { addv[k] -= w0[k]; add[k] = addv[k]; }
    if (fix_count != 0) { nd.fix_count[i] = 0; nd.fix_cap[i] = 0; }
  }
  work[i] = wk;
  nd.isexist[i] = (fixv[9] >= addv[9]) ? 0 : 1;       // :1292-1295
}
// Two passes.  `list`: one thread per node (coalesced reads of the work words) appends the marginalised slot's points to the node's
// fix region when they fit (a handful of points per node) and clears the slot's running sums; nodes whose region has to MOVE first -- a
// copy of up to max_points points -- go onto a work list (one atomic per wave).  `points`: one 64-lane workgroup per listed node, strided
// over the list -- the points are independent of one another, so the lanes copy / transform them side by side.  (One thread per node
// for everything took 106 us per call at 120k nodes, the movers' serial copies being the tail; one workgroup per node for everything
// 240 us: most nodes have one or two points and 64 times the load instructions.)
__global__ void map_margi_list_kernel(Nodes nd, Params prm, int n_nodes, PoseArg poses, RingArg ring, ScanSlots scans, const int* __restrict__ work, int* __restrict__ list,
                                      long long* __restrict__ list_dst, double* __restrict__ fix_pnt, double* __restrict__ fix_var, Counters* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = prm.win_size, m0 = ring.mp[0];
  const int wk = i < n_nodes ? work[i] : 0;
  const bool heavy = (wk & 3) == 3;
  const unsigned long long mask = __ballot(heavy);
  if (mask) {
    // the movers of this wave: list slots and new regions from ONE atomic each (a mover per atomic -- ~20k read-modify-writes of one
    // word -- was what the kernel behind this one spent 200 us on)
    const int lane = threadIdx.x & 63;
    long long mine = 0;
    if (heavy) mine = 2ll * (nd.fix_count[i] + nd.pt_count[(size_t)i * W + m0]);
    long long incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const long long t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    const long long total = __shfl(incl, 63, 64);
    int base = 0;
    unsigned long long rbase = 0;
    if (lane == 0) { base = atomicAdd(&cnt->n_list, __popcll(mask)); rbase = atomicAdd((unsigned long long*)&cnt->fix_cursor, (unsigned long long)total); }
    base = __shfl(base, 0, 64);
    rbase = __shfl(rbase, 0, 64);
    if (heavy) {
      const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
      list[slot] = i;
      list_dst[slot] = (long long)rbase + incl - mine;
    }
  }
  if (wk == 0 || heavy) return;
  double* l0 = nd.pcrs_local + ((size_t)i * W + m0) * 10;      // :1283-1288
  const double l0n = l0[9];                                     // (read here, in front of the stores below: behind them it waited for every one of them)
  if (wk & 1) {
    const ScanSlot sc = scans.s[m0];
    const int p0 = nd.pt_start[(size_t)i * W + m0], pc = nd.pt_count[(size_t)i * W + m0];
    const int fc = nd.fix_count[i];
    const long long dst = nd.fix_start[i] + fc;
    for (int j = 0; j < pc; j++) {                   // pv.pnt = R * pv.pnt + p; point_fix.push_back(pv)  (:1262-1266)
      const int pi = sc.perm[p0 + j];
      double xb[3], v9[9], w[3];                     // the point's twelve values in registers before its twelve stores (a load behind a store waits for it: may alias)
#pragma unroll
      for (int e = 0; e < 3; e++) xb[e] = sc.pnt[3 * (size_t)pi + e];
#pragma unroll
      for (int e = 0; e < 9; e++) v9[e] = sc.var9[9 * (size_t)pi + e];
      to_world(poses.Rp, xb, w);
#pragma unroll
      for (int e = 0; e < 3; e++) fix_pnt[3 * (size_t)(dst + j) + e] = w[e];
#pragma unroll
      for (int e = 0; e < 9; e++) fix_var[9 * (size_t)(dst + j) + e] = v9[e];
    }
    nd.fix_count[i] = fc + pc;
  }
  if (l0n != 0.0) {
    for (int k = 0; k < 10; k++) l0[k] = 0.0;
    nd.pt_count[(size_t)i * W + m0] = 0;
  }
}
__global__ __launch_bounds__(64) void map_margi_points_kernel(Nodes nd, Params prm, PoseArg poses, RingArg ring, ScanSlots scans, const int* __restrict__ work,
                                                              const int* __restrict__ list, const long long* __restrict__ list_dst, double* __restrict__ fix_pnt,
                                                              double* __restrict__ fix_var, Counters* cnt) {
  const int lane = threadIdx.x;
  const int W = prm.win_size, m0 = ring.mp[0];
  const ScanSlot sc = scans.s[m0];
  const int n_list = cnt->n_list;
  for (int q = blockIdx.x; q < n_list; q += gridDim.x) {
    const int i = list[q];
    const int wk = work[i];
    const int p0 = nd.pt_start[(size_t)i * W + m0], pc = nd.pt_count[(size_t)i * W + m0];
    const int fc = nd.fix_count[i];
    long long base = nd.fix_start[i];
    double* l0 = nd.pcrs_local + ((size_t)i * W + m0) * 10;      // :1283-1288
    const bool had = l0[9] != 0.0;                  // read by every lane before any of them writes -- and before the stores below (behind them the load waits for them)
    if (wk & 2) {
      const long long dst = list_dst[q];
      const long long src = base;
      for (int k = lane; k < 3 * fc; k += 64) fix_pnt[3 * (size_t)dst + k] = fix_pnt[3 * (size_t)src + k];
      for (int k = lane; k < 9 * fc; k += 64) fix_var[9 * (size_t)dst + k] = fix_var[9 * (size_t)src + k];
      base = dst;
    }
    const long long dst = base + fc;
    for (int j = lane; j < pc; j += 64) {          // pv.pnt = R * pv.pnt + p; point_fix.push_back(pv)  (:1262-1266)
      const int pi = sc.perm[p0 + j];
      double xb[3], v9[9], w[3];                   // the point's twelve values in registers before its twelve stores
#pragma unroll
      for (int e = 0; e < 3; e++) xb[e] = sc.pnt[3 * (size_t)pi + e];
#pragma unroll
      for (int e = 0; e < 9; e++) v9[e] = sc.var9[9 * (size_t)pi + e];
      to_world(poses.Rp, xb, w);
#pragma unroll
      for (int e = 0; e < 3; e++) fix_pnt[3 * (size_t)(dst + j) + e] = w[e];
#pragma unroll
      for (int e = 0; e < 9; e++) fix_var[9 * (size_t)(dst + j) + e] = v9[e];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      if (wk & 2) { nd.fix_start[i] = base; nd.fix_cap[i] = 2 * (fc + pc); }
      nd.fix_count[i] = fc + pc;
      if (had) nd.pt_count[(size_t)i * W + m0] = 0;
    }
    if (had && lane < 10) l0[lane] = 0.0;
  }
}
// internal nodes, one layer at a time from the bottom: isexist = any child (voxel_map.hpp:1297-1304)
__global__ void map_margi_up_kernel(Nodes nd, int n_nodes, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes || nd.layer[i] != L || nd.state[i] != 1 || !nd.in_slide[nd.root[i]]) return;
  int e = 0;
  for (int o = 0; o < 8; o++) {
    const int c = nd.child[8 * (size_t)i + o] - 1;
    if (c >= 0) e |= nd.isexist[c];
  }
  nd.isexist[i] = e;
}
// roots without live content leave the slide map: clear_slwd on the subtree (voxelslam.cpp:1384-1393, voxel_map.hpp:1482-1500)
__global__ void map_release_kernel(Nodes nd, int W, int n_nodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const int r = nd.root[i];
  if (!nd.in_slide[r] || nd.isexist[r] || !nd.has_sw[i]) return;
  for (int s = 0; s < W; s++) {
    nd.pt_count[(size_t)i * W + s] = 0;
    for (int k = 0; k < 10; k++) nd.pcrs_local[((size_t)i * W + s) * 10 + k] = 0.0;
  }
  nd.has_sw[i] = 0;
}
__global__ void map_leave_slide_kernel(Nodes nd, int n_nodes, Counters* cnt, double jour) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes || nd.root[i] != i || nd.layer[i] != 0) return;
  if (nd.in_slide[i]) nd.jour[i] = jour;       // `iter->second->jour = jour` for every root of the slide map (voxelslam.cpp:1349), leaving or not
  if (nd.in_slide[i] && !nd.isexist[i]) { nd.in_slide[i] = 0; atomicAdd(&cnt->n_removed, 1); }
}

// ---- release of far-away roots (voxelslam.cpp:1503-1523, OctoTree::tras_ptr voxel_map.hpp:1394-1405) -------------------------------------
// keep[i] = 0 for every node under a root that has not been in a multi_margi for `min_age` journeys (int(jour_now - root.jour) >= min_age, the
// reference's test) -- and is not in the slide map: upstream would delete such a root under the sliding window's feet; here it stays.
__global__ void map_release_flag_kernel(Nodes nd, int n_nodes, double jour_now, int min_age, unsigned int* __restrict__ keep, Counters* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const int r = nd.root[i];
  const int dis = (int)(jour_now - nd.jour[r]);
  const bool gone = !nd.in_slide[r] && dis >= min_age;
  keep[i] = gone ? 0u : 1u;
  if (gone && r == i) atomicAdd(&cnt->n_removed, 1);
}
// "nothing here any more" for the odometry's plane map: one layer-0 non-plane entry per released root clears all of its cells
__global__ void map_release_export_kernel(Nodes nd, int n_nodes, const unsigned int* __restrict__ keep, long long* __restrict__ loc, int* __restrict__ layer, int* __restrict__ path,
                                          int* __restrict__ is_plane, int* n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes || keep[i] || nd.root[i] != i) return;
  const int k = atomicAdd(n_out, 1);
  const unsigned long long key = nd.key[i];
  loc[3 * k] = (long long)((key >> 32) & 0xffff) - LOC_OFF; loc[3 * k + 1] = (long long)((key >> 16) & 0xffff) - LOC_OFF; loc[3 * k + 2] = (long long)(key & 0xffff) - LOC_OFF;
  layer[k] = 0; path[k] = 0; is_plane[k] = 0;
}
// gather of one node array into its compacted copy: dst[new(i) * mult + k] = src[i * mult + k] for the nodes that stay
template <class T>
__global__ void map_compact_array_kernel(const T* __restrict__ src, T* __restrict__ dst, int n_nodes, int mult, const unsigned int* __restrict__ keep,
                                         const unsigned int* __restrict__ pos) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (long long)n_nodes * mult) return;
  const int i = (int)(q / mult), k = (int)(q - (long long)i * mult);
  if (keep[i]) dst[(size_t)pos[i] * mult + k] = src[q];
}
// node references of the compacted pool: children (index + 1, 0 = none) and the root of every node
__global__ void map_compact_refs_kernel(Nodes nd, int n_new, const unsigned int* __restrict__ pos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  nd.root[i] = (int)pos[nd.root[i]];
  for (int o = 0; o < 8; o++) {
    const int c = nd.child[8 * (size_t)i + o];
    if (c > 0) nd.child[8 * (size_t)i + o] = (int)pos[c - 1] + 1;
  }
}
__global__ void map_table_insert_roots_kernel(Nodes nd, int n_nodes, unsigned long long* keys, int* vals, unsigned long long cap_mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes || nd.root[i] != i || nd.layer[i] != 0) return;
  const unsigned long long key = nd.key[i];
  unsigned long long h = mix64(key) & cap_mask;
  for (;;) {
    if (atomicCAS(&keys[h], EMPTY_KEY, key) == EMPTY_KEY) { vals[h] = i + 1; return; }
    h = (h + 1) & cap_mask;
  }
}

// ---- export -----------------------------------------------------------------------------------------------------------------
__global__ void map_leaf_flag_kernel(Nodes nd, int n_nodes, int* __restrict__ list, int* n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes || nd.state[i] != 0) return;
  list[atomicAdd(n_out, 1)] = i;
}
// the record layout documented at vxba_map_leaves (include/vxba.h): ints 8, doubles 156 + 11 W
__global__ void map_leaf_export_kernel(Nodes nd, int W, RingArg ring, const int* __restrict__ list, int n, unsigned long long* __restrict__ ids, int* __restrict__ ints,
                                       double* __restrict__ dbl) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const int i = list[a];
  ids[a] = (nd.key[i] << 16) | ((unsigned long long)nd.path[i] << 7) | (unsigned long long)nd.layer[i];
  int* I = ints + 8 * (size_t)a;
  I[0] = nd.layer[i]; I[1] = nd.isexist[i]; I[2] = nd.is_plane[i]; I[3] = nd.has_sw[i]; I[4] = nd.opt_state[i]; I[5] = nd.last_num[i]; I[6] = nd.fix_count[i];
  I[7] = nd.in_slide[nd.root[i]];
  const size_t rec = 156 + 11 * (size_t)W;
  double* D = dbl + rec * a;
  for (int k = 0; k < 10; k++) { D[k] = nd.pcr_add[10 * (size_t)i + k]; D[10 + k] = nd.pcr_fix[10 * (size_t)i + k]; }
  for (int k = 0; k < 3; k++) D[20 + k] = nd.eigval[3 * (size_t)i + k];
  for (int k = 0; k < 9; k++) D[23 + k] = nd.eigvec[9 * (size_t)i + k];
  for (int k = 0; k < 3; k++) { D[32 + k] = nd.pl_center[3 * (size_t)i + k]; D[35 + k] = nd.pl_normal[3 * (size_t)i + k]; }
  D[38] = nd.pl_radius[i];
  for (int k = 0; k < 36; k++) D[39 + k] = nd.is_plane[i] ? nd.pl_var[36 * (size_t)i + k] : 0.0;
  for (int k = 0; k < 81; k++) D[75 + k] = nd.cov_add[81 * (size_t)i + k];
  for (int s = 0; s < W; s++) {
    const int m = ring.mp[s];
    for (int k = 0; k < 10; k++) D[156 + 10 * s + k] = nd.has_sw[i] ? nd.pcrs_local[((size_t)i * W + m) * 10 + k] : 0.0;
    D[156 + 10 * W + s] = nd.has_sw[i] ? (double)nd.pt_count[(size_t)i * W + m] : 0.0;
  }
}
// Plane records for the odometry's plane map (vxba_lio): every leaf under a root whose subtree may have changed since the last export,
// plus an explicit "no plane here" entry for every octant of a subdivided node that has no child -- together they tile the root, so
// the entries never overlap and one launch of the odometry's update kernel applies them race-free.
__global__ void map_plane_export_kernel(Nodes nd, int n_nodes, int max_layer, long long* __restrict__ loc, int* __restrict__ layer, int* __restrict__ path, int* __restrict__ is_plane,
                                        double* __restrict__ center, double* __restrict__ normal, double* __restrict__ plane_var, double* __restrict__ radius, int* n_out, int capacity) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes || !nd.dirty[nd.root[i]]) return;
  const unsigned long long key = nd.key[i];
  const long long lx = (long long)((key >> 32) & 0xffff) - LOC_OFF, ly = (long long)((key >> 16) & 0xffff) - LOC_OFF, lz = (long long)(key & 0xffff) - LOC_OFF;
  const int L = nd.layer[i], p = nd.path[i];
  auto lio_path = [](int p9, int lay) { int r = 0; for (int l = 0; l < lay; l++) r |= ((p9 >> (3 * (2 - l))) & 7) << (3 * l); return r; };
  if (nd.state[i] == 0) {
    // Every load of the record BEFORE the first store (round 4, from the ISA): written as `out[e] = nd.x[e]` the 42 copies became 42 load -> wait -> store
    // round trips, one after the other (the node arrays are plain pointers in a struct: for all the compiler knows a store may hit the next
    // load's address), and with ~100k threads on 1024 SIMDs nothing hides a round trip: 61 us for 40 MB.
    const int plane = (nd.is_plane[i] && nd.last_num[i] > 0) ? 1 : 0;     // a leaf that never saw plane_update holds an all-zero record: never matched
    double c3[3], n3[3], pv[36];
#pragma unroll
    for (int e = 0; e < 3; e++) { c3[e] = nd.pl_center[3 * (size_t)i + e]; n3[e] = nd.pl_normal[3 * (size_t)i + e]; }
#pragma unroll
    for (int e = 0; e < 36; e++) pv[e] = nd.pl_var[36 * (size_t)i + e];
    const double rad = nd.pl_radius[i];
    const int k = atomicAdd(n_out, 1);
    if (k >= capacity) return;
    loc[3 * k] = lx; loc[3 * k + 1] = ly; loc[3 * k + 2] = lz;
    layer[k] = L; path[k] = lio_path(p, L);
    is_plane[k] = plane;
#pragma unroll
    for (int e = 0; e < 3; e++) { center[3 * k + e] = c3[e]; normal[3 * k + e] = n3[e]; }
#pragma unroll
    for (int e = 0; e < 36; e++) plane_var[36 * (size_t)k + e] = pv[e];
    radius[k] = rad;
  } else if (L < max_layer) {
    for (int o = 0; o < 8; o++) {
      if (nd.child[8 * (size_t)i + o] != 0) continue;
      const int k = atomicAdd(n_out, 1);
      if (k >= capacity) continue;        // the counting pass (capacity 0) must still see every empty octant
      loc[3 * k] = lx; loc[3 * k + 1] = ly; loc[3 * k + 2] = lz;
      layer[k] = L + 1; path[k] = lio_path(p | (o << (3 * (2 - L))), L + 1);
      is_plane[k] = 0;
      for (int e = 0; e < 3; e++) { center[3 * k + e] = 0; normal[3 * k + e] = 0; }
      for (int e = 0; e < 36; e++) plane_var[36 * (size_t)k + e] = 0;
      radius[k] = 0;
    }
  }
}
__global__ void map_dirty_reset_kernel(Nodes nd, int n_nodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_nodes && nd.root[i] == i && nd.layer[i] == 0) nd.dirty[i] = nd.in_slide[i];     // roots still in the slide map keep changing
}
__global__ void map_gather_body_kernel(const double* __restrict__ soa, long long n, long long stride, double* __restrict__ pnt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 3; k++) pnt[3 * i + k] = soa[k * stride + i];
}
__global__ void map_rehash_kernel(const unsigned long long* __restrict__ old_keys, const int* __restrict__ old_vals, long long old_cap, unsigned long long* keys, int* vals,
                                  unsigned long long cap_mask) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= old_cap || old_keys[i] == EMPTY_KEY) return;
  unsigned long long h = mix64(old_keys[i]) & cap_mask;
  for (;;) {
    if (atomicCAS(&keys[h], EMPTY_KEY, old_keys[i]) == EMPTY_KEY) { vals[h] = old_vals[i]; return; }
    h = (h + 1) & cap_mask;
  }
}
__global__ void map_fill_u64_kernel(unsigned long long* p, long long n, unsigned long long v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}


// ---- fix-point pool compaction ----------------------------------------------------------------------------------------------
// The pool is a bump allocator: a leaf that outgrows its region gets a new one at the cursor (margi), a split leaf's points move to
// its children (subdivide), a leaf that reaches max_points drops its points -- the old regions are never handed out again.  When the
// cursor has run far ahead of what is live, the live regions (capacity included: the slack keeps the next append in place) are moved,
// in node order, to the front of a fresh pool.  Pure relocation: contents and order of every leaf's points are unchanged.
__global__ void map_fix_caps_kernel(Nodes nd, int n_nodes, long long* __restrict__ caps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const bool live = nd.fix_count[i] > 0;
  if (!live) nd.fix_cap[i] = 0;
  caps[i] = live ? (long long)nd.fix_cap[i] : 0ll;
}
__global__ void map_fix_move_kernel(Nodes nd, int n_nodes, const long long* __restrict__ new_start, const double* __restrict__ old_pnt, const double* __restrict__ old_var,
                                    double* __restrict__ new_pnt, double* __restrict__ new_var) {
  const int i = blockIdx.x;                       // one 64-lane workgroup per node
  if (i >= n_nodes) return;
  const int fc = nd.fix_count[i];
  if (fc <= 0) return;
  const long long src = nd.fix_start[i], dst = new_start[i];
  for (int k = threadIdx.x; k < 3 * fc; k += blockDim.x) new_pnt[3 * (size_t)dst + k] = old_pnt[3 * (size_t)src + k];
  for (int k = threadIdx.x; k < 9 * fc; k += blockDim.x) new_var[9 * (size_t)dst + k] = old_var[9 * (size_t)src + k];
  __syncthreads();
  if (threadIdx.x == 0) nd.fix_start[i] = dst;
}
}  // namespace vxmap

// =====================================================================================================================================
struct vxba_map {
  int device = 0;
  long long export_cap = 0;           // entries the plane-export buffers are sized for (from the last export)
  vxmap::Params prm{};
  hipStream_t stream = nullptr;
  vxmap::Nodes nd{};
  std::vector<void*> node_allocs;     // every array of `nd`, for growth / free
  unsigned long long* keys = nullptr;
  int* vals = nullptr;
  long long table_cap = 0;
  vxmap::Counters* d_cnt = nullptr;
  vxmap::Counters* h_cnt = nullptr;   // pinned, mapped: cnt_publish_kernel writes it
  unsigned* h_cnt_dev = nullptr;      // the device's address of h_cnt
  int n_nodes = 0, n_roots = 0, n_slide = 0;
  int serial = 0;
  int mp[vxmap::MAXW];
  // one resident scan per window slot
  struct Scan { double* pnt = nullptr; double* var9 = nullptr; int* perm = nullptr; int* tmp = nullptr; int n = 0, cap = 0; } scan[vxmap::MAXW];
  double* fix_pnt = nullptr; double* fix_var = nullptr; long long fix_cap = 0, fix_cursor = 0;
  long long fix_compact_min = 1ll << 22, fix_compact_at = 1ll << 22;   // compaction of the fix pool once the cursor passes fix_compact_at (points)
  long long n_fix_compactions = 0;
  char* scratch = nullptr; size_t scratch_cap = 0;
  char* stage = nullptr; size_t stage_cap = 0;    // second grow-only buffer: outputs that live next to the scratch of the same call
  std::string err;
  double jour = 0.0;                  // the caller's journey odometer (vxba_map_set_journey): what the next margi stamps on the slide map's roots
  long long n_released_roots = 0, n_releases = 0;
  bool broken = false;                // a failure left the device-side tree in an unknown state (see compact_fix): every stage refuses from then on
};

namespace {
using namespace vxmap;

int mfail(vxba_map* m, int code, const char* msg) { if (m) m->err = msg; return code; }
#define VM_HIP(m, call)                                                                     \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess) { (m)->err = std::string(#call ": ") + hipGetErrorString(e__); return VXBA_ERR_HIP; } \
  } while (0)
inline int grid_for(long long n, int b = 256) { return (int)std::max<long long>(1, (n + b - 1) / b); }   // never a zero-sized launch: the kernels bound-check
// Completion of the map's stream by polling: the stages between two counter read-backs are tens of microseconds of kernels, less than
// what waking up from hipStreamSynchronize costs (the same observation as in the LiDAR-inertial shell: ~25 us per wait).
inline hipError_t map_wait(hipStream_t s) {
  hipError_t q;
  q = vxwait::stream_wait(s);
  return q;
}

template <class T>
int grow_array(vxba_map* m, T** p, size_t old_n, size_t new_n) {
  T* q = nullptr;
  VM_HIP(m, hipMalloc((void**)&q, new_n * sizeof(T)));
  VM_HIP(m, hipMemsetAsync(q, 0, new_n * sizeof(T), m->stream));
  if (*p && old_n) VM_HIP(m, hipMemcpyAsync(q, *p, old_n * sizeof(T), hipMemcpyDeviceToDevice, m->stream));
  if (*p) { VM_HIP(m, map_wait(m->stream)); VM_HIP(m, hipFree(*p)); }
  *p = q;
  return VXBA_OK;
}
int ensure_nodes(vxba_map* m, long long want) {
  Nodes& nd = m->nd;
  if (want <= nd.cap) return VXBA_OK;
  long long ncap = std::max<long long>(nd.cap ? 2ll * nd.cap : 1 << 16, want + want / 2);
  const size_t o = nd.cap, n = (size_t)ncap, W = m->prm.win_size;
  int rc;
#define G(field, mult) if ((rc = grow_array(m, &nd.field, o * (mult), n * (mult)))) return rc;
  G(layer, 1) G(state, 1) G(child, 8) G(root, 1) G(isexist, 1) G(has_sw, 1) G(is_plane, 1) G(last_num, 1) G(opt_state, 1) G(in_slide, 1) G(stamp, 1) G(path, 1) G(dirty, 1)
  G(key, 1) G(center, 3) G(pcr_add, 10) G(pcr_fix, 10) G(cov_add, 81) G(eigval, 3) G(eigvec, 9) G(pl_center, 3) G(pl_normal, 3) G(pl_radius, 1) G(pl_var, 36)
  G(pcrs_local, 10 * W) G(ql, 1) G(pt_start, W) G(pt_count, W) G(fix_start, 1) G(fix_count, 1) G(fix_cap, 1) G(jour, 1)
#undef G
  nd.cap = (int)ncap;
  return VXBA_OK;
}
int ensure_table(vxba_map* m, long long roots) {
  if (m->table_cap >= 2 * roots && m->table_cap > 0) return VXBA_OK;
  long long ncap = 1 << 12;
  while (ncap < 4 * roots) ncap <<= 1;
  unsigned long long* k = nullptr; int* v = nullptr;
  VM_HIP(m, hipMalloc((void**)&k, ncap * sizeof(unsigned long long)));
  VM_HIP(m, hipMalloc((void**)&v, ncap * sizeof(int)));
  map_fill_u64_kernel<<<grid_for(ncap), 256, 0, m->stream>>>(k, ncap, EMPTY_KEY);
  VM_HIP(m, hipMemsetAsync(v, 0, ncap * sizeof(int), m->stream));
  if (m->keys) {
    map_rehash_kernel<<<grid_for(m->table_cap), 256, 0, m->stream>>>(m->keys, m->vals, m->table_cap, k, v, (unsigned long long)ncap - 1);
    VM_HIP(m, map_wait(m->stream));
    hipFree(m->keys); hipFree(m->vals);
  }
  m->keys = k; m->vals = v; m->table_cap = ncap;
  return VXBA_OK;
}
int ensure_fix(vxba_map* m, long long want) {
  if (want <= m->fix_cap) return VXBA_OK;
  const long long ncap = std::max<long long>(m->fix_cap ? 2 * m->fix_cap : 1 << 18, want + want / 4);
  int rc;
  if ((rc = grow_array(m, &m->fix_pnt, (size_t)m->fix_cap * 3, (size_t)ncap * 3))) return rc;
  if ((rc = grow_array(m, &m->fix_var, (size_t)m->fix_cap * 9, (size_t)ncap * 9))) return rc;
  m->fix_cap = ncap;
  return VXBA_OK;
}
int ensure_scratch(vxba_map* m, size_t bytes);
// Move the live regions of the fix-point pool to the front of a fresh pool when the cursor has passed m->fix_compact_at (see the kernels).
// Called between stages (the host's view of the counters is current, nothing is in flight that allocates).
int compact_fix(vxba_map* m) {
  if (m->fix_cursor <= m->fix_compact_at || m->n_nodes == 0 || !m->fix_pnt) return VXBA_OK;
  const int n = m->n_nodes;
  size_t tb = 0;
  VM_HIP(m, rocprim::exclusive_scan(nullptr, tb, (long long*)nullptr, (long long*)nullptr, 0ll, (size_t)n, rocprim::plus<long long>(), m->stream));
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t b_l = up((size_t)n * sizeof(long long));
  int rc = ensure_scratch(m, 2 * b_l + up(tb));
  if (rc) return rc;
  long long* d_caps = (long long*)m->scratch;
  long long* d_start = (long long*)(m->scratch + b_l);
  void* d_tmp = m->scratch + 2 * b_l;
  map_fix_caps_kernel<<<grid_for(n), 256, 0, m->stream>>>(m->nd, n, d_caps);
  VM_HIP(m, rocprim::exclusive_scan(d_tmp, tb, d_caps, d_start, 0ll, (size_t)n, rocprim::plus<long long>(), m->stream));
  long long last[2] = {0, 0};
  VM_HIP(m, hipMemcpyAsync(&last[0], d_start + (n - 1), sizeof(long long), hipMemcpyDeviceToHost, m->stream));
  VM_HIP(m, hipMemcpyAsync(&last[1], d_caps + (n - 1), sizeof(long long), hipMemcpyDeviceToHost, m->stream));
  VM_HIP(m, map_wait(m->stream));
  const long long live = last[0] + last[1];
  const long long ncap = std::max<long long>(1ll << 18, live + live / 2 + (1ll << 16));
  double *np = nullptr, *nv = nullptr;
  VM_HIP(m, hipMalloc((void**)&np, (size_t)ncap * 3 * sizeof(double)));
  if (hipMalloc((void**)&nv, (size_t)ncap * 9 * sizeof(double)) != hipSuccess) { hipFree(np); return mfail(m, VXBA_ERR_HIP, "vxba_map: out of memory compacting the fix-point pool"); }
  map_fix_move_kernel<<<dim3((unsigned)n), 64, 0, m->stream>>>(m->nd, n, d_start, m->fix_pnt, m->fix_var, np, nv);
  {
    // The kernel rewrites every node's fix_start as it moves the node's region: once it is launched, a failure leaves starts that may point
    // into either pool.  Nothing is committed on the host side then -- the new pool is released and the map is marked unusable, so that
    // no later stage reads regions through stale starts (a map_create + replay of the window is the caller's way out).
    hipError_t e1 = map_wait(m->stream);
    if (e1 == hipSuccess) e1 = hipGetLastError();
    if (e1 != hipSuccess) {
      hipFree(np); hipFree(nv);
      m->broken = true;
      m->err = std::string("vxba_map: compacting the fix-point pool failed (") + hipGetErrorString(e1) + "); the map is unusable from here on";
      return VXBA_ERR_HIP;
    }
  }
  hipFree(m->fix_pnt); hipFree(m->fix_var);
  m->fix_pnt = np; m->fix_var = nv; m->fix_cap = ncap; m->fix_cursor = live;
  m->fix_compact_at = std::max(m->fix_compact_min, 3 * live);
  m->n_fix_compactions++;
  return VXBA_OK;
}
int ensure_scratch(vxba_map* m, size_t bytes) {
  if (bytes <= m->scratch_cap) return VXBA_OK;
  if (m->scratch) { VM_HIP(m, map_wait(m->stream)); hipFree(m->scratch); m->scratch = nullptr; m->scratch_cap = 0; }
  const size_t want = bytes + bytes / 2;
  VM_HIP(m, hipMalloc((void**)&m->scratch, want));
  m->scratch_cap = want;
  return VXBA_OK;
}
int ensure_stage(vxba_map* m, size_t bytes) {
  if (bytes <= m->stage_cap) return VXBA_OK;
  if (m->stage) { VM_HIP(m, map_wait(m->stream)); hipFree(m->stage); m->stage = nullptr; m->stage_cap = 0; }
  const size_t want = bytes + bytes / 2;
  VM_HIP(m, hipMalloc((void**)&m->stage, want));
  m->stage_cap = want;
  return VXBA_OK;
}
// counters: push the host view to the device before a stage, pull it back after
int cnt_push(vxba_map* m) {
  Counters c;
  c.n_nodes = m->n_nodes; c.n_roots = m->n_roots; c.n_touched = 0; c.n_slide_new = 0; c.n_split = 0; c.n_fac = 0; c.n_removed = 0; c.err = 0; c.n_list = 0; c.pad_ = 0; for (int k = 0; k < 4; k++) { c.n_split_l[k] = 0; c.fix_need_l[k] = 0; } c.fix_cursor = m->fix_cursor; c.fix_need = 0;
  vxmap::cnt_reset_kernel<<<1, 64, 0, m->stream>>>(m->d_cnt, c);
  VM_HIP(m, hipGetLastError());
  return VXBA_OK;
}
int cnt_pull(vxba_map* m) {
  volatile unsigned* hw = reinterpret_cast<volatile unsigned*>(m->h_cnt);
  for (int k = 0; k < vxmap::CNT_WORDS; k++) hw[k] = 0xffffffffu;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  vxmap::cnt_publish_kernel<<<1, 64, 0, m->stream>>>(m->d_cnt, m->h_cnt_dev);
  VM_HIP(m, hipGetLastError());
  // poll the words; every few thousand rounds ask the stream as well (a failed launch or a device fault would otherwise never end the wait)
  for (unsigned spins = 1;; spins++) {
    bool all = true;
    for (int k = 0; k < vxmap::CNT_WORDS; k++) all = all && hw[k] != 0xffffffffu;
    if (all) break;
    if ((spins & 0xfff) == 0) {
      const hipError_t q = hipStreamQuery(m->stream);
      if (q == hipSuccess) {     // the kernel has ended: its stores are on their way, give them a moment, then give up
        bool ok2 = false;
        for (int t = 0; t < 100000 && !ok2; t++) { ok2 = true; for (int k = 0; k < vxmap::CNT_WORDS; k++) ok2 = ok2 && hw[k] != 0xffffffffu; }
        if (ok2) break;
        return mfail(m, VXBA_ERR_HIP, "vxba_map: the counter block did not arrive in host memory");
      }
      if (q != hipErrorNotReady) { m->err = std::string("vxba_map: ") + hipGetErrorString(q); return VXBA_ERR_HIP; }
    }
  }
  std::atomic_thread_fence(std::memory_order_seq_cst);
  m->n_nodes = m->h_cnt->n_nodes; m->n_roots = m->h_cnt->n_roots; m->fix_cursor = m->h_cnt->fix_cursor;
  return VXBA_OK;
}
PoseArg make_poses(const double* Rp, int n) {
  PoseArg p;
  std::memset(&p, 0, sizeof p);
  std::memcpy(p.Rp, Rp, sizeof(double) * 12 * n);
  return p;
}
RingArg make_ring(const vxba_map* m) {
  RingArg r;
  for (int i = 0; i < MAXW; i++) r.mp[i] = i < m->prm.win_size ? m->mp[i] : 0;
  return r;
}
ScanSlots make_scans(const vxba_map* m) {
  ScanSlots s;
  for (int i = 0; i < MAXW; i++) s.s[i] = ScanSlot{m->scan[i].pnt, m->scan[i].var9, m->scan[i].perm, m->scan[i].tmp, m->scan[i].n};
  return s;
}
int check_err(vxba_map* m, const char* what) {     // after cnt_pull
  const int e = m->h_cnt->err;
  if (e == 0) return VXBA_OK;
  if (e == 1) return mfail(m, VXBA_ERR_UNSUPPORTED, "vxba_map: a point lies outside the +-32768 voxel range");
  if (e == 2) return mfail(m, VXBA_ERR_STATE, "vxba_map_margi: opt_state beyond the factor (the factor is not the one recut filled)");
  return mfail(m, VXBA_ERR_STATE, what);
}
}  // namespace

extern "C" {

int vxba_map_create(const vxba_map_params* p, int device, vxba_map** out) {
  if (!out || !p) return VXBA_ERR_ARG;
  *out = nullptr;
  if (!(p->voxel_size > 0) || p->max_layer < 0 || p->max_layer > 2 || p->win_size < 1 || p->win_size > vxmap::MAXW || p->thread_num < 1) return VXBA_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VXBA_ERR_NODEV;
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_HIP;
  vxba_map* m = new vxba_map();
  m->device = device;
  m->prm.voxel_size = p->voxel_size; m->prm.max_layer = p->max_layer; m->prm.min_eigen_value = p->min_eigen_value;
  for (int k = 0; k < 4; k++) { m->prm.min_point[k] = p->min_point[k]; m->prm.thre[k] = p->plane_eigen_value_thre[k]; }
  m->prm.max_points = p->max_points; m->prm.win_size = p->win_size; m->prm.thread_num = p->thread_num;
  for (int i = 0; i < vxmap::MAXW; i++) m->mp[i] = i;
  if (const char* e = getenv("VXBA_MAP_FIX_COMPACT_AT")) {   // points; development / tests (the default compacts from 4M abandoned + live points on)
    const long long v = atoll(e);
    if (v > 0) m->fix_compact_min = m->fix_compact_at = v;
  }
  bool ok = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipMalloc((void**)&m->d_cnt, sizeof(vxmap::Counters)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&m->h_cnt, sizeof(vxmap::Counters), hipHostMallocMapped) == hipSuccess;
  ok = ok && hipHostGetDevicePointer((void**)&m->h_cnt_dev, m->h_cnt, 0) == hipSuccess;
  if (!ok || ensure_nodes(m, 1 << 16) != VXBA_OK || ensure_table(m, 1 << 10) != VXBA_OK || ensure_fix(m, 1 << 18) != VXBA_OK) { vxba_map_destroy(m); return VXBA_ERR_HIP; }
  *out = m;
  return VXBA_OK;
}

int vxba_map_destroy(vxba_map* m) {
  if (!m) return VXBA_ERR_ARG;
  hipSetDevice(m->device);
  if (m->stream) hipStreamSynchronize(m->stream);
  vxmap::Nodes& nd = m->nd;
  void* arrs[] = {nd.layer, nd.state, nd.child, nd.root, nd.isexist, nd.has_sw, nd.is_plane, nd.last_num, nd.opt_state, nd.in_slide, nd.stamp, nd.path, nd.dirty, nd.key, nd.center,
                  nd.pcr_add, nd.pcr_fix, nd.cov_add, nd.eigval, nd.eigvec, nd.pl_center, nd.pl_normal, nd.pl_radius, nd.pl_var, nd.pcrs_local, nd.ql, nd.pt_start,
                  nd.pt_count, nd.fix_start, nd.fix_count, nd.fix_cap, nd.jour};
  for (void* a : arrs) if (a) hipFree(a);
  for (auto& s : m->scan) { hipFree(s.pnt); hipFree(s.var9); hipFree(s.perm); hipFree(s.tmp); }
  hipFree(m->keys); hipFree(m->vals); hipFree(m->d_cnt); hipFree(m->fix_pnt); hipFree(m->fix_var); hipFree(m->scratch); hipFree(m->stage);
  if (m->h_cnt) hipHostFree(m->h_cnt);
  if (m->stream) hipStreamDestroy(m->stream);
  delete m;
  return VXBA_OK;
}

const char* vxba_map_last_error(const vxba_map* m) { return m ? m->err.c_str() : "null map"; }

// cut_voxel_multi(surf_map, pvec, ord, surf_map_slide, win_size, pwld, sws)   (voxel_map.hpp:1545-1639; voxelslam.cpp:1609)
static int map_cut_voxel_impl(vxba_map* m, int ord, int64_t n64, const double* pnt_body, const double* var_world, const double* pwld, bool on_device) {
  if (!m || ord < 0 || ord >= m->prm.win_size || n64 < 0 || n64 > 0x3fffffff || (n64 > 0 && (!pnt_body || !var_world || !pwld))) return mfail(m, VXBA_ERR_ARG, "vxba_map_cut_voxel: bad argument");
  if (m->broken) return VXBA_ERR_STATE;   // m->err still names the failure that broke it
  hipSetDevice(m->device);
  const int n = (int)n64;
  const int slot = m->mp[ord];
  vxba_map::Scan& sc = m->scan[slot];
  if (n > sc.cap) {
    VM_HIP(m, map_wait(m->stream));
    hipFree(sc.pnt); hipFree(sc.var9); hipFree(sc.perm); hipFree(sc.tmp);
    sc.pnt = sc.var9 = nullptr; sc.perm = sc.tmp = nullptr;
    const size_t cap = (size_t)n + n / 4 + 64;
    VM_HIP(m, hipMalloc((void**)&sc.pnt, cap * 3 * sizeof(double)));
    VM_HIP(m, hipMalloc((void**)&sc.var9, cap * 9 * sizeof(double)));
    VM_HIP(m, hipMalloc((void**)&sc.perm, cap * sizeof(int)));
    VM_HIP(m, hipMalloc((void**)&sc.tmp, cap * sizeof(int)));
    sc.cap = (int)cap;
  }
  sc.n = n;
  if (n == 0) return VXBA_OK;
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  VM_HIP(m, hipMemcpyAsync(sc.pnt, pnt_body, (size_t)n * 3 * sizeof(double), kind, m->stream));
  VM_HIP(m, hipMemcpyAsync(sc.var9, var_world, (size_t)n * 9 * sizeof(double), kind, m->stream));
  // scratch: world points, table slot / leaf / pending / sorted leaf per point, radix-sort temporaries
  size_t tb = 0;
  rocprim::radix_sort_pairs(nullptr, tb, (int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)n, 0, 32, m->stream);
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t b_w = up((size_t)n * 3 * sizeof(double)), b_i = up((size_t)n * sizeof(int));
  int rc = ensure_scratch(m, b_w + 5 * b_i + up(tb));
  if (rc) return rc;
  char* q = m->scratch;
  double* d_w = (double*)q; q += b_w;
  int* d_slot = (int*)q; q += b_i;
  int* d_leaf = (int*)q; q += b_i;
  int* d_pend = (int*)q; q += b_i;
  int* d_leaf_s = (int*)q; q += b_i;
  int* d_iota = (int*)q; q += b_i;
  void* d_tmp = q;
  if (on_device) d_w = const_cast<double*>(pwld);      // only read during this call
  else VM_HIP(m, hipMemcpyAsync(d_w, pwld, (size_t)n * 3 * sizeof(double), kind, m->stream));
  if ((rc = ensure_table(m, (long long)m->n_roots + n))) return rc;
  if ((rc = ensure_nodes(m, (long long)m->n_nodes + 2ll * n))) return rc;     // at most one new root and one new child per point
  m->serial++;
  const int nodes_before = m->n_nodes;
  if ((rc = cnt_push(m))) return rc;
  map_roots_kernel<<<grid_for(n), 256, 0, m->stream>>>(m->nd, m->prm, m->keys, m->vals, (unsigned long long)m->table_cap - 1, d_w, n, d_slot, m->d_cnt, m->serial);
  map_descend_kernel<<<grid_for(n), 256, 0, m->stream>>>(m->nd, m->vals, d_w, n, d_slot, d_leaf, d_pend, m->d_cnt, m->serial);
  if ((rc = cnt_pull(m))) return rc;
  if ((rc = check_err(m, "vxba_map_cut_voxel"))) return rc;
  m->n_slide += m->h_cnt->n_slide_new;
  map_mark_existing_roots_kernel<<<grid_for(nodes_before), 256, 0, m->stream>>>(m->nd, nodes_before, m->serial);
  // upstream quirk (voxel_map.hpp:1603-1605): fewer touched roots than threads -> nothing is pushed.  The children the descent above
  // allocated are empty leaves then, which changes nothing observable (no window, no points, never a factor).
  if (m->h_cnt->n_touched < m->prm.thread_num) { VM_HIP(m, map_wait(m->stream)); return VXBA_OK; }
  map_resolve_kernel<<<grid_for(n), 256, 0, m->stream>>>(m->nd, n, d_leaf, d_pend);
  map_iota_kernel<<<grid_for(n), 256, 0, m->stream>>>(d_iota, n);
  VM_HIP(m, rocprim::radix_sort_pairs(d_tmp, tb, d_leaf, d_leaf_s, d_iota, sc.perm, (size_t)n, 0, 32, m->stream));
  map_push_kernel<<<grid_for(n, 64), 64, 0, m->stream>>>(m->nd, m->prm, d_leaf_s, sc.perm, n, sc.pnt, sc.var9, d_w, slot);
  VM_HIP(m, map_wait(m->stream));
  VM_HIP(m, hipGetLastError());
  return VXBA_OK;
}
int vxba_map_cut_voxel(vxba_map* m, int ord, int64_t n, const double* pnt_body, const double* var_world, const double* pwld) {
  return map_cut_voxel_impl(m, ord, n, pnt_body, var_world, pwld, false);
}
int vxba_map_cut_voxel_device(vxba_map* m, int ord, int64_t n, const double* d_pnt_body, const double* d_var_world, const double* d_pwld) {
  return map_cut_voxel_impl(m, ord, n, d_pnt_body, d_var_world, d_pwld, true);
}

// multi_recut (voxelslam.cpp:1396-1453): recut of every root of the slide map, then tras_opt into `factor` (cleared by the caller like
// voxhess.clear(); its win_size must be the map's)
int vxba_map_recut(vxba_map* m, int win_count, const double* Rp, vxba_factor* factor, int64_t* n_pushed) {
  if (!m || !Rp || !factor || win_count < 1 || win_count > m->prm.win_size) return mfail(m, VXBA_ERR_ARG, "vxba_map_recut: bad argument");
  if (m->broken) return VXBA_ERR_STATE;   // m->err still names the failure that broke it
  if (vxba_win_size(factor) != m->prm.win_size) return mfail(m, VXBA_ERR_ARG, "vxba_map_recut: the factor's win_size differs from the map's");
  if (vxba_internal_factor_device(factor) != m->device) return mfail(m, VXBA_ERR_ARG, "vxba_map_recut: the factor lives on another device than the map (raw device pointers are exchanged)");
  hipSetDevice(m->device);
  if (n_pushed) *n_pushed = 0;
  if (m->n_slide < m->prm.thread_num) return VXBA_OK;                      // `if(g_size < thd_num) return;`
  const PoseArg poses = make_poses(Rp, win_count);
  const RingArg ring = make_ring(m);
  int rc;
  if ((rc = compact_fix(m))) return rc;
  if ((rc = cnt_push(m))) return rc;
  int bound = m->n_nodes;                      // upper bound of the node count on the device
  for (int L = 0; L <= m->prm.max_layer; L++) {
    if ((rc = ensure_scratch(m, (size_t)bound * sizeof(int)))) return rc;
    int* d_split = (int*)m->scratch;
    map_judge_kernel<<<grid_for(bound), 256, 0, m->stream>>>(m->nd, m->prm, bound, L, d_split, m->d_cnt);
    if (L >= m->prm.max_layer) break;          // leaves of the finest layer never split: nothing to read back
    if ((rc = cnt_pull(m))) return rc;         // one read-back per layer: how many leaves split
    const int n_split = m->h_cnt->n_split_l[L];
    { static const bool dbg = getenv("VXBA_MAP_DEBUG") != nullptr; if (dbg) fprintf(stderr, "vxba_map_recut: layer %d: %d of %d nodes split, fix points needed %lld\n", L, n_split, bound, (long long)m->h_cnt->fix_need_l[L]); }
    if (n_split == 0) continue;
    if ((rc = ensure_nodes(m, (long long)m->n_nodes + 8ll * n_split))) return rc;
    if ((rc = ensure_fix(m, m->fix_cursor + m->h_cnt->fix_need_l[L]))) return rc;
    map_subdivide_wave_kernel<<<dim3((unsigned)n_split), 64, 0, m->stream>>>(m->nd, m->prm, d_split, n_split, win_count, poses, ring, make_scans(m), m->fix_pnt, m->fix_var, m->d_cnt);
    bound = m->n_nodes + 8 * n_split;
  }
  // tras_opt, ordered by node id so that the factor is the same from run to run
  const size_t b_id = (size_t)bound * sizeof(unsigned long long), b_nd = (size_t)bound * sizeof(int);
  size_t tb = 0;
  rocprim::radix_sort_pairs(nullptr, tb, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)bound, 0, 64, m->stream);
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  if ((rc = ensure_scratch(m, 2 * up(b_id) + 2 * up(b_nd) + up(tb)))) return rc;
  char* q = m->scratch;
  unsigned long long* d_ids = (unsigned long long*)q; q += up(b_id);
  unsigned long long* d_ids_s = (unsigned long long*)q; q += up(b_id);
  int* d_nodes = (int*)q; q += up(b_nd);
  int* d_nodes_s = (int*)q; q += up(b_nd);
  void* d_tmp = q;
  map_factor_flag_kernel<<<grid_for(bound), 256, 0, m->stream>>>(m->nd, bound, d_ids, d_nodes, m->d_cnt);
  if ((rc = cnt_pull(m))) return rc;           // also brings n_nodes / fix_cursor up to date
  const int nf = m->h_cnt->n_fac;
  if (nf == 0) return VXBA_OK;
  VM_HIP(m, rocprim::radix_sort_pairs(d_tmp, tb, d_ids, d_ids_s, d_nodes, d_nodes_s, (size_t)nf, 0, 64, m->stream));
  const int W = m->prm.win_size;
  const size_t per = (size_t)W * 10 + 10 + 1 + 3 + 9 + 10;
  if ((rc = ensure_stage(m, (size_t)nf * per * sizeof(double)))) return rc;
  double* d_stage = (double*)m->stage;
  double* d_cl = d_stage; double* d_fix = d_cl + (size_t)nf * W * 10; double* d_coe = d_fix + (size_t)nf * 10; double* d_ev = d_coe + nf;
  double* d_evec = d_ev + (size_t)nf * 3; double* d_mg = d_evec + (size_t)nf * 9;
  map_factor_gather_kernel<<<dim3((unsigned)nf), 192, 0, m->stream>>>(m->nd, W, ring, d_nodes_s, nf, d_cl, d_fix, d_coe, d_ev, d_evec, d_mg);
  hipError_t e = map_wait(m->stream);
  if (e == hipSuccess) rc = vxba_internal_push_voxels_device(factor, nf, d_cl, d_fix, d_coe, d_ev, d_evec, d_mg);
  if (e != hipSuccess) return mfail(m, VXBA_ERR_HIP, "vxba_map_recut: gather failed");
  if (rc != VXBA_OK) return mfail(m, rc, vxba_last_error(factor));
  if (n_pushed) *n_pushed = nf;
  return VXBA_OK;
}

// multi_margi (voxelslam.cpp:1321-1394) with mgsize = 1: the factor is the one recut filled; its cache (pcr_adds, eig_values,
// eig_vectors as the optimiser left them) is read on the device
int vxba_map_margi(vxba_map* m, int win_count, const double* Rp, vxba_factor* factor) {
  if (!m || !Rp || !factor || win_count < 1 || win_count > m->prm.win_size) return mfail(m, VXBA_ERR_ARG, "vxba_map_margi: bad argument");
  if (m->broken) return VXBA_ERR_STATE;   // m->err still names the failure that broke it
  if (vxba_internal_factor_device(factor) != m->device) return mfail(m, VXBA_ERR_ARG, "vxba_map_margi: the factor lives on another device than the map (its cache planes are read in place)");
  hipSetDevice(m->device);
  if (m->n_slide < m->prm.thread_num) return VXBA_OK;
  const double *f_ev = nullptr, *f_evec = nullptr, *f_mg = nullptr;
  int VS = 0, V = 0;
  int rc = vxba_internal_cache_view(factor, &f_ev, &f_evec, &f_mg, &VS, &V);     // synchronises the factor's stream
  if (rc != VXBA_OK) return mfail(m, rc, vxba_last_error(factor));
  const PoseArg poses = make_poses(Rp, win_count);
  const RingArg ring = make_ring(m);
  if ((rc = compact_fix(m))) return rc;
  if ((rc = ensure_scratch(m, (size_t)m->n_nodes * (sizeof(long long) + 2 * sizeof(int))))) return rc;
  long long* d_list_dst = (long long*)m->scratch;
  int* d_work = (int*)(d_list_dst + m->n_nodes);
  int* d_list = d_work + m->n_nodes;
  if ((rc = cnt_push(m))) return rc;
  map_margi_kernel<<<grid_for(m->n_nodes), 256, 0, m->stream>>>(m->nd, m->prm, m->n_nodes, win_count, poses, ring, f_ev, f_evec, f_mg, VS, V, d_work, m->d_cnt);
  if ((rc = cnt_pull(m))) return rc;
  if ((rc = check_err(m, "vxba_map_margi"))) return rc;
  if ((rc = ensure_fix(m, m->fix_cursor + m->h_cnt->fix_need))) return rc;
  if ((rc = cnt_push(m))) return rc;
  map_margi_list_kernel<<<grid_for(m->n_nodes), 256, 0, m->stream>>>(m->nd, m->prm, m->n_nodes, poses, ring, make_scans(m), d_work, d_list, d_list_dst, m->fix_pnt, m->fix_var, m->d_cnt);
  map_margi_points_kernel<<<dim3((unsigned)std::min(m->n_nodes, 4096)), 64, 0, m->stream>>>(m->nd, m->prm, poses, ring, make_scans(m), d_work, d_list, d_list_dst, m->fix_pnt, m->fix_var, m->d_cnt);
  for (int L = m->prm.max_layer - 1; L >= 0; L--) map_margi_up_kernel<<<grid_for(m->n_nodes), 256, 0, m->stream>>>(m->nd, m->n_nodes, L);
  map_release_kernel<<<grid_for(m->n_nodes), 256, 0, m->stream>>>(m->nd, m->prm.win_size, m->n_nodes);
  map_leave_slide_kernel<<<grid_for(m->n_nodes), 256, 0, m->stream>>>(m->nd, m->n_nodes, m->d_cnt, m->jour);
  if ((rc = cnt_pull(m))) return rc;
  m->n_slide -= m->h_cnt->n_removed;
  return VXBA_OK;
}

// The caller's journey odometer (`jour += spat`, voxelslam.cpp:1677): stamped on every root of the slide map by the next vxba_map_margi.
int vxba_map_set_journey(vxba_map* m, double jour) {
  if (!m) return VXBA_ERR_ARG;
  m->jour = jour;
  return VXBA_OK;
}

// The release branch of the local-mapping loop (voxelslam.cpp:1503-1523): every root voxel whose last multi_margi lies `min_age` journeys back
// (700 upstream) leaves surf_map with its whole subtree (OctoTree::tras_ptr + delete).  Here: the node pool is COMPACTED -- surviving nodes
// move to the front of freshly sized arrays in their old order (so every later id-ordered step sees the order it would have seen), child /
// root references and the voxel table are rebuilt, the fix-point pool is compacted behind it -- and device memory shrinks with the map.
// `lio` (optional): the odometry handle whose plane map mirrors this map; the released roots are cleared there as well.
int vxba_map_release(vxba_map* m, double jour_now, int min_age, vxba_lio* lio, int64_t* n_roots_released, int64_t* n_nodes_released) {
  if (n_roots_released) *n_roots_released = 0;
  if (n_nodes_released) *n_nodes_released = 0;
  if (!m || min_age < 0) return mfail(m, VXBA_ERR_ARG, "vxba_map_release: bad argument");
  if (m->broken) return VXBA_ERR_STATE;
  if (lio) {
    double vs = 0; int ml = 0, dev = 0;
    vxba_internal_lio_geometry(lio, &vs, &ml, &dev);
    if (vs != m->prm.voxel_size || ml != m->prm.max_layer || dev != m->device) return mfail(m, VXBA_ERR_ARG, "vxba_map_release: voxel_size / max_layer / device of the two handles differ");
  }
  hipSetDevice(m->device);
  const int n = m->n_nodes;
  if (n == 0) return VXBA_OK;
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  size_t tb = 0;
  VM_HIP(m, rocprim::exclusive_scan(nullptr, tb, (unsigned int*)nullptr, (unsigned int*)nullptr, 0u, (size_t)n, rocprim::plus<unsigned int>(), m->stream));
  const size_t b_u = up((size_t)n * sizeof(unsigned int));
  int rc = ensure_scratch(m, 2 * b_u + up(tb) + 256);
  if (rc) return rc;
  unsigned int* d_keep = (unsigned int*)m->scratch;
  unsigned int* d_pos = (unsigned int*)(m->scratch + b_u);
  void* d_tmp = m->scratch + 2 * b_u;
  int* d_n = (int*)(m->scratch + 2 * b_u + up(tb));
  if ((rc = cnt_push(m))) return rc;
  map_release_flag_kernel<<<grid_for(n), 256, 0, m->stream>>>(m->nd, n, jour_now, min_age, d_keep, m->d_cnt);
  VM_HIP(m, rocprim::exclusive_scan(d_tmp, tb, d_keep, d_pos, 0u, (size_t)n, rocprim::plus<unsigned int>(), m->stream));
  unsigned int last[2] = {0, 0};
  VM_HIP(m, hipMemcpyAsync(&last[0], d_pos + (n - 1), sizeof(unsigned int), hipMemcpyDeviceToHost, m->stream));
  VM_HIP(m, hipMemcpyAsync(&last[1], d_keep + (n - 1), sizeof(unsigned int), hipMemcpyDeviceToHost, m->stream));
  if ((rc = cnt_pull(m))) return rc;
  const int n_gone_roots = m->h_cnt->n_removed;
  const int n_new = (int)(last[0] + last[1]);
  if (n_gone_roots == 0) return VXBA_OK;
  if (lio) {
    // stage (second grow-only buffer): the "no plane here" entries of the released roots; the record arrays of a non-plane entry are never read
    const size_t c_loc = up((size_t)n_gone_roots * 3 * 8), c_i = up((size_t)n_gone_roots * 4);
    if ((rc = ensure_stage(m, c_loc + 3 * c_i + 256))) return rc;
    char* q = m->stage;
    long long* d_loc = (long long*)q; q += c_loc;
    int* d_layer = (int*)q; q += c_i;
    int* d_path = (int*)q; q += c_i;
    int* d_isp = (int*)q; q += c_i;
    VM_HIP(m, hipMemsetAsync(d_n, 0, sizeof(int), m->stream));
    map_release_export_kernel<<<grid_for(n), 256, 0, m->stream>>>(m->nd, n, d_keep, d_loc, d_layer, d_path, d_isp, d_n);
    VM_HIP(m, map_wait(m->stream));
    VM_HIP(m, hipGetLastError());
    const double* dummy = (const double*)m->stage;
    rc = vxba_internal_lio_map_update_device(lio, n_gone_roots, d_loc, d_layer, d_path, d_isp, dummy, dummy, dummy, dummy);
    if (rc != VXBA_OK) return mfail(m, rc, vxba_lio_last_error(lio));
  }
  // every node array into a fresh one sized for what stays.  From the first swapped array on the pool is in a mixed state: a failure
  // (out of memory) leaves the map unusable -- say so rather than go on.
  Nodes& nd = m->nd;
  const long long ncap = std::max<long long>(1 << 12, (long long)n_new + n_new / 2);   // sized for what stays: the next scan's ensure_nodes grows it as before
  const size_t W = m->prm.win_size;
  bool failed = false;
  auto gather = [&](auto** arr, size_t mult) {
    using T = std::remove_pointer_t<std::remove_pointer_t<decltype(arr)>>;
    if (failed) return;
    T* q = nullptr;
    if (hipMalloc((void**)&q, (size_t)ncap * mult * sizeof(T)) != hipSuccess) { failed = true; return; }
    hipMemsetAsync(q, 0, (size_t)ncap * mult * sizeof(T), m->stream);
    map_compact_array_kernel<T><<<grid_for((long long)n * (long long)mult), 256, 0, m->stream>>>(*arr, q, n, (int)mult, d_keep, d_pos);
    if (map_wait(m->stream) != hipSuccess || hipGetLastError() != hipSuccess) { hipFree(q); failed = true; return; }
    hipFree(*arr);
    *arr = q;
  };
#define G(field, mult) gather(&nd.field, (size_t)(mult));
  G(layer, 1) G(state, 1) G(child, 8) G(root, 1) G(isexist, 1) G(has_sw, 1) G(is_plane, 1) G(last_num, 1) G(opt_state, 1) G(in_slide, 1) G(stamp, 1) G(path, 1) G(dirty, 1)
  G(key, 1) G(center, 3) G(pcr_add, 10) G(pcr_fix, 10) G(cov_add, 81) G(eigval, 3) G(eigvec, 9) G(pl_center, 3) G(pl_normal, 3) G(pl_radius, 1) G(pl_var, 36)
  G(pcrs_local, 10 * W) G(ql, 1) G(pt_start, W) G(pt_count, W) G(fix_start, 1) G(fix_count, 1) G(fix_cap, 1) G(jour, 1)
#undef G
  if (failed) { m->broken = true; return mfail(m, VXBA_ERR_HIP, "vxba_map_release: out of memory while compacting the node pool; the map is unusable from here on"); }
  nd.cap = (int)ncap;
  map_compact_refs_kernel<<<grid_for(n_new), 256, 0, m->stream>>>(nd, n_new, d_pos);
  // the voxel table, from scratch (released keys must not linger: open addressing has no delete)
  map_fill_u64_kernel<<<grid_for(m->table_cap), 256, 0, m->stream>>>(m->keys, m->table_cap, EMPTY_KEY);
  hipError_t te = hipMemsetAsync(m->vals, 0, (size_t)m->table_cap * sizeof(int), m->stream);
  map_table_insert_roots_kernel<<<grid_for(n_new), 256, 0, m->stream>>>(nd, n_new, m->keys, m->vals, (unsigned long long)m->table_cap - 1);
  if (te == hipSuccess) te = map_wait(m->stream);
  if (te == hipSuccess) te = hipGetLastError();
  if (te != hipSuccess) {   // the nodes have moved and the table may not point at them: as unusable as a half-compacted pool
    m->broken = true;
    return mfail(m, VXBA_ERR_HIP, "vxba_map_release: rebuilding the voxel table failed after the node pool was compacted; the map is unusable from here on");
  }
  m->n_nodes = n_new;
  m->n_roots -= n_gone_roots;
  m->n_released_roots += n_gone_roots;
  m->n_releases++;
  // the released leaves' regions of the fix-point pool are dead weight now: compact it behind the nodes
  const long long saved_at = m->fix_compact_at;
  m->fix_compact_at = -1;
  rc = compact_fix(m);
  if (m->fix_compact_at < 0) m->fix_compact_at = saved_at;      // nothing live: compact_fix returned early
  if (rc) return rc;
  if (n_roots_released) *n_roots_released = n_gone_roots;
  if (n_nodes_released) *n_nodes_released = n - n_new;
  return VXBA_OK;
}

// Device memory held by the map, bytes: [0] node pool (capacity x ~2.3 KB), [1] fix-point pool, [2] resident scans of the window, [3] voxel table +
// scratch + stage, [4] total.
int vxba_map_device_bytes(vxba_map* m, int64_t out[5]) {
  if (!m || !out) return VXBA_ERR_ARG;
  const size_t W = m->prm.win_size;
  const size_t per_node = 13 * sizeof(int) + sizeof(unsigned long long) + sizeof(double) * (3 + 10 + 10 + 81 + 3 + 9 + 3 + 3 + 1 + 36 + 10 * W + 1) + sizeof(float) + 2 * W * sizeof(int) + sizeof(long long) + 2 * sizeof(int);
  out[0] = (int64_t)((size_t)m->nd.cap * per_node);
  out[1] = (int64_t)((size_t)m->fix_cap * 12 * sizeof(double));
  size_t sc = 0;
  for (const auto& s : m->scan) sc += (size_t)s.cap * (12 * sizeof(double) + 2 * sizeof(int));
  out[2] = (int64_t)sc;
  out[3] = (int64_t)((size_t)m->table_cap * (sizeof(unsigned long long) + sizeof(int)) + m->scratch_cap + m->stage_cap);
  out[4] = out[0] + out[1] + out[2] + out[3];
  return VXBA_OK;
}

// voxelslam.cpp:1683-1687
int vxba_map_slide(vxba_map* m, int mgsize) {
  if (!m || mgsize < 0) return VXBA_ERR_ARG;
  for (int i = 0; i < m->prm.win_size; i++) {
    m->mp[i] += mgsize;
    if (m->mp[i] >= m->prm.win_size) m->mp[i] -= m->prm.win_size;
  }
  return VXBA_OK;
}

int vxba_map_fix_pool(vxba_map* m, int64_t out[3]) {
  if (!m || !out) return VXBA_ERR_ARG;
  out[0] = m->fix_cursor; out[1] = m->fix_cap; out[2] = m->n_fix_compactions;
  return VXBA_OK;
}
int vxba_map_counts(vxba_map* m, int64_t out[4]) {
  if (!m || !out) return VXBA_ERR_ARG;
  hipSetDevice(m->device);
  int* d_n = nullptr;
  int rc = ensure_scratch(m, (size_t)(m->n_nodes + 64) * sizeof(int));
  if (rc) return rc;
  d_n = (int*)m->scratch;
  VM_HIP(m, hipMemsetAsync(d_n, 0, sizeof(int), m->stream));
  vxmap::map_leaf_flag_kernel<<<grid_for(m->n_nodes), 256, 0, m->stream>>>(m->nd, m->n_nodes, d_n + 64, d_n);
  int nl = 0;
  VM_HIP(m, hipMemcpyAsync(&nl, d_n, sizeof(int), hipMemcpyDeviceToHost, m->stream));
  VM_HIP(m, map_wait(m->stream));
  out[0] = m->n_roots; out[1] = m->n_slide; out[2] = nl; out[3] = m->mp[0];
  return VXBA_OK;
}

// Every leaf (octo_state == 0), in no particular order; record layout as documented in include/vxba.h.  Returns the count in *n_out.
int vxba_map_leaves(vxba_map* m, int64_t capacity, uint64_t* ids, int32_t* ints, double* dbl, int64_t* n_out) {
  if (!m || !n_out) return VXBA_ERR_ARG;
  hipSetDevice(m->device);
  const int W = m->prm.win_size;
  const size_t rec = 156 + 11 * (size_t)W;
  int rc = ensure_scratch(m, (size_t)(m->n_nodes + 64) * sizeof(int));
  if (rc) return rc;
  int* d_n = (int*)m->scratch;
  int* d_list = d_n + 64;
  VM_HIP(m, hipMemsetAsync(d_n, 0, sizeof(int), m->stream));
  vxmap::map_leaf_flag_kernel<<<grid_for(m->n_nodes), 256, 0, m->stream>>>(m->nd, m->n_nodes, d_list, d_n);
  int nl = 0;
  VM_HIP(m, hipMemcpyAsync(&nl, d_n, sizeof(int), hipMemcpyDeviceToHost, m->stream));
  VM_HIP(m, map_wait(m->stream));
  *n_out = nl;
  if (!ids || !ints || !dbl || capacity <= 0) return VXBA_OK;
  const int n = (int)std::min<int64_t>(nl, capacity);
  if (n == 0) return VXBA_OK;
  unsigned long long* d_ids = nullptr; int* d_ints = nullptr; double* d_dbl = nullptr;
  VM_HIP(m, hipMalloc((void**)&d_ids, (size_t)n * sizeof(unsigned long long)));
  VM_HIP(m, hipMalloc((void**)&d_ints, (size_t)n * 8 * sizeof(int)));
  VM_HIP(m, hipMalloc((void**)&d_dbl, (size_t)n * rec * sizeof(double)));
  vxmap::map_leaf_export_kernel<<<grid_for(n), 256, 0, m->stream>>>(m->nd, W, make_ring(m), d_list, n, d_ids, d_ints, d_dbl);
  hipError_t e = hipMemcpyAsync(ids, d_ids, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(ints, d_ints, (size_t)n * 8 * sizeof(int), hipMemcpyDeviceToHost, m->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dbl, d_dbl, (size_t)n * rec * sizeof(double), hipMemcpyDeviceToHost, m->stream);
  if (e == hipSuccess) e = map_wait(m->stream);
  hipFree(d_ids); hipFree(d_ints); hipFree(d_dbl);
  return e == hipSuccess ? VXBA_OK : mfail(m, VXBA_ERR_HIP, "vxba_map_leaves: copy failed");
}

// cut_voxel_multi on the scan resident in an odometry handle after vxba_lio_pvec_update: nothing crosses PCIe.
int vxba_map_cut_voxel_lio(vxba_map* m, int ord, vxba_lio* lio) {
  if (!m || !lio) return mfail(m, VXBA_ERR_ARG, "vxba_map_cut_voxel_lio: null argument");
  if (m->broken) return VXBA_ERR_STATE;   // m->err still names the failure that broke it
  {
    double vs_ = 0; int ml_ = 0, dev_ = -1;
    vxba_internal_lio_geometry(lio, &vs_, &ml_, &dev_);
    if (dev_ != m->device) return mfail(m, VXBA_ERR_ARG, "vxba_map_cut_voxel_lio: the odometry handle lives on another device than the map (its scan arrays are read in place)");
  }
  const double *soa = nullptr, *world = nullptr;
  long long n = 0, stride = 0;
  int valid = 0;
  int rc = vxba_internal_lio_scan_view(lio, &soa, &n, &stride, &world, &valid);
  if (rc != VXBA_OK) return mfail(m, rc, "vxba_map_cut_voxel_lio: cannot read the odometry handle");
  if (n > 0 && !valid) return mfail(m, VXBA_ERR_STATE, "vxba_map_cut_voxel_lio: no world points on the device (call vxba_lio_pvec_update first)");
  if (n == 0) return map_cut_voxel_impl(m, ord, 0, nullptr, nullptr, nullptr, true);
  hipSetDevice(m->device);
  if ((rc = ensure_stage(m, (size_t)n * 3 * sizeof(double)))) return rc;
  double* d_pnt = (double*)m->stage;
  vxmap::map_gather_body_kernel<<<grid_for(n), 256, 0, m->stream>>>(soa, n, stride, d_pnt);
  return map_cut_voxel_impl(m, ord, n, d_pnt, world + 3 * n, world, true);
}

// The odometry's plane map (vxba_lio) brought up to date with the tree: the leaves (and empty octants) under every root that was in
// the slide map since the last export.  Call after vxba_map_recut / vxba_map_margi, before the next scan is matched.
int vxba_map_export_planes(vxba_map* m, vxba_lio* lio, int64_t* n_exported) {
  if (!m || !lio) return mfail(m, VXBA_ERR_ARG, "vxba_map_export_planes: null argument");
  if (m->broken) return VXBA_ERR_STATE;
  double vs = 0; int ml = 0, dev = 0;
  vxba_internal_lio_geometry(lio, &vs, &ml, &dev);
  if (vs != m->prm.voxel_size || ml != m->prm.max_layer || dev != m->device) return mfail(m, VXBA_ERR_ARG, "vxba_map_export_planes: voxel_size / max_layer / device of the two handles differ");
  hipSetDevice(m->device);
  if (n_exported) *n_exported = 0;
  if (m->n_nodes == 0) return VXBA_OK;
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  // The worst case (every node exports eight entries) is far above what a scan changes, so the buffers are sized from the LAST
  // export (x 1.5): one pass that writes what fits and counts everything; only a first call or a jump in the number of changed leaves
  // pays a second pass.  (Until the end of round 2 every export was a counting pass + a host round trip + the real pass.)
  int rc = VXBA_OK, n = 0;
  long long* d_loc = nullptr; int *d_layer = nullptr, *d_path = nullptr, *d_isp = nullptr, *d_n = nullptr;
  double *d_center = nullptr, *d_normal = nullptr, *d_pvar = nullptr, *d_radius = nullptr;
  for (int attempt = 0; attempt < 2; attempt++) {
    const long long capn = m->export_cap;
    const size_t c_loc = up((size_t)capn * 3 * 8), c_i = up((size_t)capn * 4), c_3 = up((size_t)capn * 3 * 8), c_36 = up((size_t)capn * 36 * 8), c_1 = up((size_t)capn * 8);
    if ((rc = ensure_scratch(m, 256 + c_loc + 3 * c_i + 2 * c_3 + c_36 + c_1))) return rc;
    char* q = m->scratch;
    d_n = (int*)q; q += 256;
    d_loc = (long long*)q; q += c_loc;
    d_layer = (int*)q; q += c_i;
    d_path = (int*)q; q += c_i;
    d_isp = (int*)q; q += c_i;
    d_center = (double*)q; q += c_3;
    d_normal = (double*)q; q += c_3;
    d_pvar = (double*)q; q += c_36;
    d_radius = (double*)q;
    // the entry count rides in the counter block (n_list): a reset kernel and a publish kernel instead of a memset and a blit copy
    if ((rc = cnt_push(m))) return rc;
    d_n = &m->d_cnt->n_list;
    vxmap::map_plane_export_kernel<<<grid_for(m->n_nodes), 256, 0, m->stream>>>(m->nd, m->n_nodes, m->prm.max_layer, d_loc, d_layer, d_path, d_isp, d_center, d_normal, d_pvar, d_radius, d_n,
                                                                                  (int)capn);
    if ((rc = cnt_pull(m))) return rc;
    n = m->h_cnt->n_list;
    if (n <= capn) break;
    m->export_cap = (long long)n + n / 2 + 1024;      // everything was counted, not everything written: once more with room
  }
  if (m->export_cap < (long long)n + n / 2) m->export_cap = (long long)n + n / 2 + 1024;
  if (n == 0) return VXBA_OK;
  vxmap::map_dirty_reset_kernel<<<grid_for(m->n_nodes), 256, 0, m->stream>>>(m->nd, m->n_nodes);
  VM_HIP(m, hipGetLastError());
  rc = vxba_internal_lio_map_update_device(lio, n, d_loc, d_layer, d_path, d_isp, d_center, d_normal, d_pvar, d_radius);
  if (rc != VXBA_OK) return mfail(m, rc, vxba_lio_last_error(lio));
  if (n_exported) *n_exported = n;
  return VXBA_OK;
}

}  // extern "C"
