// Batch factor construction on the GPU: points of a window -> voxel hash -> octree subdivision -> plane test -> BA factor.
// The reference does this with an unordered_map of octree nodes walked on the host (OctreeGBA::cut_voxel / subdivide /
// recut, loop_refine.hpp:273-476; call site voxelslam.cpp:2374-2379).  Here it is a sort:
//
//   every point gets one 64-bit key  [ x:16 | y:16 | z:16 | octant path: 3 bits per layer, 9 | frame: 7 ]
//   (root voxel exactly as upstream: float quotient, "-1 if negative", truncation; octants by strict '>' against the node
//   centres, which are carried along with upstream's float quarter lengths), and for each octree layer l one stable radix
//   sort of (key truncated to layer l, frame) makes every (node, frame) cell and every node a contiguous point range, in
//   the order upstream pushes the points (frame by frame, cloud order) -- so the cluster sums are bit-identical to
//   PointCluster::push.  Nodes are then judged top-down (N > 10, plane_judge, >= 2 observing frames, lambda0/lambda1 <=
//   0.12; non-planes are subdivided while layer < max_layer), and the accepted ones are appended to the factor's planes
//   on the device, cache seeded with (lambda, U, world cluster) like recut's push_voxel does.
//
// Round 6: only layer 0 is a sort.  The cloud arrives frame by frame in cloud order, so the order a layer needs -- (node of layer l, frame,
// cloud index) -- is the stable sort of the ORIGINAL sequence by the node key alone; and since a node of layer l is (node of layer l-1,
// one octant), the order of layer l is the order of layer l-1 with every node's point range stably partitioned by that octant: one pass of
// partition_kernel (a workgroup per node of layer l-1: eight ballots per 64 points) in place of a 64-bit radix sort of all n keys, which
// at the sizes of a window (<= 1M points) rocPRIM runs as a block sort + eight merge passes, each launch-bound (a quarter of the GPU time
// of a hierarchical-BA pass, profiles/r05_cfg5).  Same permutation, hence the same cells, nodes and cluster sums bit for bit.  The counts
// the host needs between launches are written by the kernels that produce them straight into mapped pinned memory (rocPRIM's run-length
// encode takes the pinned word as its count output; the others through a one-thread kernel): no copy engine round trips.
#include "vxba_wait.hpp"
#include <hip/hip_runtime.h>
#include <atomic>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "vxba_kernels.h"
#include "vxba_math.hpp"
#include "vxba_voxelize.h"

namespace vxv {

constexpr int FRAME_BITS = 7, PATH_BITS = 9;

// bbox (round 6): per-axis minimum and maximum of the root-voxel coordinates over the cloud (block-wide in LDS, then one global atomic per
// block and component) -- the host turns it into a compact layer-0 sort key (relkey_kernel)
__global__ void init_err_bbox_kernel(int* __restrict__ err, int* __restrict__ bbox) {
  if (threadIdx.x == 0) err[0] = 0;
  if (threadIdx.x < 3) bbox[threadIdx.x] = 0x7fffffff;
  else if (threadIdx.x < 6) bbox[threadIdx.x] = (int)0x80000000;
}
__global__ void key_kernel(const double* __restrict__ xyz, const long long* __restrict__ frame_ptr, int W, const double* __restrict__ poses, VoxelizeParams p,
                           long long n, double* __restrict__ world, unsigned long long* __restrict__ key, int* __restrict__ err, int* __restrict__ bbox) {
#pragma clang fp contract(off)
  __shared__ int bb[6];
  if (threadIdx.x < 3) bb[threadIdx.x] = 0x7fffffff;
  else if (threadIdx.x < 6) bb[threadIdx.x] = (int)0x80000000;
  __syncthreads();
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) {
    int f = 0;
    while (f + 1 < W && q >= frame_ptr[f + 1]) f++;
    const double* Rp = poses + 12 * f;
    const double x = xyz[3 * q], y = xyz[3 * q + 1], z = xyz[3 * q + 2];
    double w[3];
    for (int r = 0; r < 3; r++) w[r] = Rp[r] * x + Rp[3 + r] * y + Rp[6 + r] * z + Rp[9 + r];   // R * local + p, left to right
    unsigned long long root = 0;
    double centre[3];
    bool ok = true;
    for (int j = 0; j < 3; j++) {
      float loc = (float)(w[j] / p.voxel_size);
      if (loc < 0) loc -= 1.0f;
      const long long pos = (long long)loc;
      if (pos < -32768 || pos > 32767) { ok = false; break; }
      atomicMin(&bb[j], (int)pos);
      atomicMax(&bb[3 + j], (int)pos);
      root = (root << 16) | (unsigned long long)(pos + 32768);
      centre[j] = (0.5 + (double)pos) * p.voxel_size;
    }
    if (!ok) *err = 1;
    else {
      float quarter = (float)(p.voxel_size / 4.0);
      unsigned long long path = 0;
      for (int l = 0; l < 3; l++) {
        int b[3];
        for (int k = 0; k < 3; k++) b[k] = w[k] > centre[k] ? 1 : 0;
        path = (path << 3) | (unsigned long long)(4 * b[0] + 2 * b[1] + b[2]);
        for (int k = 0; k < 3; k++) centre[k] = centre[k] + (double)((float)(2 * b[k] - 1) * quarter);
        quarter = quarter / 2;
      }
      world[3 * q] = w[0]; world[3 * q + 1] = w[1]; world[3 * q + 2] = w[2];
      key[q] = (root << (PATH_BITS + FRAME_BITS)) | (path << FRAME_BITS) | (unsigned long long)f;
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&bbox[threadIdx.x], bb[threadIdx.x]);
  else if (threadIdx.x < 6) atomicMax(&bbox[threadIdx.x], bb[threadIdx.x]);
}
// Layer 0 through a COMPACT key (round 6): the root coordinates relative to the cloud's bounding box, packed x | y | z in bx + by + bz bits
// (<= 30; a ten-keyframe window: ~20).  The cloud is in (frame, cloud index) order already, so a stable sort by the root alone is the order
// the 64-bit (root, frame) sort produced -- and with ~20 key bits and rocPRIM's onesweep forced (its default below 1 M keys is a block sort
// + eight merge passes whatever the key) it is a histogram + three passes.  idx = the identity; the layer keys follow by a gather.
__global__ void relkey_kernel(const unsigned long long* __restrict__ key, long long n, int xmin, int ymin, int zmin, int by, int bz, unsigned int* __restrict__ rk,
                              unsigned int* __restrict__ idx) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const unsigned long long root = key[q] >> (PATH_BITS + FRAME_BITS);
  const unsigned int x = (unsigned int)((int)((root >> 32) & 0xffffull) - 32768 - xmin), y = (unsigned int)((int)((root >> 16) & 0xffffull) - 32768 - ymin),
                     z = (unsigned int)((int)(root & 0xffffull) - 32768 - zmin);
  rk[q] = (x << (by + bz)) | (y << bz) | z;
  idx[q] = (unsigned int)q;
}
__global__ void lkey_gather_kernel(const unsigned long long* __restrict__ key, const unsigned int* __restrict__ idx_s, long long n, int layer, unsigned long long* __restrict__ lkey_s) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const unsigned long long keep = ~(((1ull << (3 * (3 - layer))) - 1ull) << FRAME_BITS);
  lkey_s[q] = key[idx_s[q]] & keep;
}
__global__ void post_err_bbox_kernel(const int* __restrict__ err, const int* __restrict__ bbox, unsigned long long* __restrict__ host_err, unsigned long long* __restrict__ host_bbox) {
  if (threadIdx.x == 0) host_err[0] = (unsigned long long)(unsigned int)err[0];
  if (threadIdx.x < 6) host_bbox[threadIdx.x] = (unsigned long long)(long long)bbox[threadIdx.x];
}

// voxel-sharded windows: 1 for the points of this shard's root voxels, then their compaction to the front (order kept)
__global__ void shard_flag_kernel(const unsigned long long* __restrict__ key, long long n, int shard_index, int shard_count, unsigned int* __restrict__ flag) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) flag[q] = root_shard(key[q] >> (PATH_BITS + FRAME_BITS), shard_count) == shard_index ? 1u : 0u;
}
__global__ void shard_compact_kernel(const unsigned int* __restrict__ flag, const unsigned int* __restrict__ pos, long long n, const double* __restrict__ loc,
                                     const double* __restrict__ wld, const unsigned long long* __restrict__ key, double* __restrict__ loc_c,
                                     double* __restrict__ wld_c, unsigned long long* __restrict__ key_c) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n || !flag[q]) return;
  const unsigned int o = pos[q];
  for (int k = 0; k < 3; k++) { loc_c[3 * o + k] = loc[3 * q + k]; wld_c[3 * o + k] = wld[3 * q + k]; }
  key_c[o] = key[q];
}

// key of the (node at layer l, frame) cell: the octant bits below layer l are cleared
__global__ void layer_key_kernel(const unsigned long long* __restrict__ key, long long n, int layer, unsigned long long* __restrict__ out,
                                 unsigned int* __restrict__ idx) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const unsigned long long keep = ~(((1ull << (3 * (3 - layer))) - 1ull) << FRAME_BITS);
  out[q] = key[q] & keep;
  idx[q] = (unsigned int)q;
}
__global__ void gather3_kernel(const double* __restrict__ src, const unsigned int* __restrict__ idx, long long n, double* __restrict__ dst) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const unsigned int s = idx[q];
  dst[3 * q] = src[3 * s]; dst[3 * q + 1] = src[3 * s + 1]; dst[3 * q + 2] = src[3 * s + 2];
}
// Layer l >= 1 from layer l-1: node j of layer l-1 owns points node_ptr[j] .. node_ptr[j+1] of the sequence idx_in, ordered (frame, cloud
// index); its points are moved, order kept, into eight runs by the octant they fall into at layer l.  One workgroup per node; a node of up to
// 64 points (most of them) is done by wave 0 alone with eight ballots, longer ones in two passes over chunks of PART_BLOCK points (counts,
// then places).  Writes the new sequence and its layer keys (what layer_key_kernel + the sort produced).
constexpr int PART_BLOCK = 256;
__global__ __launch_bounds__(PART_BLOCK) void partition_kernel(const unsigned long long* __restrict__ key, const unsigned int* __restrict__ idx_in,
                                                               const long long* __restrict__ node_ptr, int layer, unsigned int* __restrict__ idx_out,
                                                               unsigned long long* __restrict__ lkey_out) {
  const long long a = node_ptr[blockIdx.x], b = node_ptr[blockIdx.x + 1];
  const long long m = b - a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int shift = FRAME_BITS + 3 * (3 - layer);
  const unsigned long long keep = ~(((1ull << (3 * (3 - layer))) - 1ull) << FRAME_BITS);
  const unsigned long long lt = (1ull << lane) - 1ull;
  if (m <= 64) {
    if (wave != 0) return;
    const bool valid = lane < m;
    unsigned int i = 0;
    unsigned long long k = 0;
    if (valid) { i = idx_in[a + lane]; k = key[i]; }
    const int o = (int)((k >> shift) & 7ull);
    unsigned int base = 0, dest = 0;
#pragma unroll
    for (int oo = 0; oo < 8; oo++) {
      const unsigned long long mask = __ballot(valid && o == oo);
      if (o == oo) dest = base + (unsigned int)__popcll(mask & lt);
      base += (unsigned int)__popcll(mask);
    }
    if (valid) { idx_out[a + dest] = i; lkey_out[a + dest] = k & keep; }
    return;
  }
  __shared__ unsigned int wcnt[PART_BLOCK / 64][8];
  __shared__ long long run[8];
  // pass 1: points per octant
  unsigned int mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long long e = a + tid; e < b; e += PART_BLOCK) {
    const int o = (int)((key[idx_in[e]] >> shift) & 7ull);
#pragma unroll
    for (int oo = 0; oo < 8; oo++) mine[oo] += (o == oo) ? 1u : 0u;
  }
#pragma unroll
  for (int oo = 0; oo < 8; oo++) {
    unsigned int v = mine[oo];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) wcnt[wave][oo] = v;
  }
  __syncthreads();
  if (tid == 0) {
    long long at = a;
    for (int oo = 0; oo < 8; oo++) {
      run[oo] = at;
      for (int w = 0; w < PART_BLOCK / 64; w++) at += wcnt[w][oo];
    }
  }
  __syncthreads();
  // pass 2: places, chunk by chunk in sequence order
  for (long long c = a; c < b; c += PART_BLOCK) {
    const long long e = c + tid;
    const bool valid = e < b;
    unsigned int i = 0;
    unsigned long long k = 0;
    if (valid) { i = idx_in[e]; k = key[i]; }
    const int o = (int)((k >> shift) & 7ull);
    unsigned int below = 0;
#pragma unroll
    for (int oo = 0; oo < 8; oo++) {
      const unsigned long long mask = __ballot(valid && o == oo);
      if (o == oo) below = (unsigned int)__popcll(mask & lt);
      if (lane == 0) wcnt[wave][oo] = (unsigned int)__popcll(mask);
    }
    __syncthreads();
    if (valid) {
      long long d = run[o] + below;
      for (int w = 0; w < wave; w++) d += wcnt[w][o];
      idx_out[d] = i;
      lkey_out[d] = k & keep;
    }
    __syncthreads();
    if (tid < 8) {
      long long add = 0;
      for (int w = 0; w < PART_BLOCK / 64; w++) add += wcnt[w][tid];
      run[tid] += add;
    }
    __syncthreads();
  }
}
// Cells and nodes of a layer from the sorted layer keys in ONE pass (round 6; rounds 1-5: run-length encode of the keys, widen, scan, shift,
// run-length encode of the cell keys, widen, scan, gather -- fourteen launches and two host round trips per layer).  A point starts a (node,
// frame) cell when its key differs from its predecessor's, a node when the key above the frame bits does; both flags ride one 64-bit scan
// (low word: cells, high word: nodes -- counts stay below 2^32), and the heads then write their own entries: cell_key / cell_ptr (first point
// of the cell), node_key / node_cell_ptr (first cell of the node) / node_ptr (first point of the node), the closing entries of the three offset
// arrays and both counts straight into the mapped pinned words the host polls.  The arrays are the ones the encodes + scans produced.
__global__ void heads_kernel(const unsigned long long* __restrict__ lkey_s, long long n, unsigned long long* __restrict__ flags) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const unsigned long long k = lkey_s[q];
  const unsigned long long pk = q ? lkey_s[q - 1] : ~k;
  flags[q] = (k != pk ? 1ull : 0ull) | ((k >> FRAME_BITS) != (pk >> FRAME_BITS) ? (1ull << 32) : 0ull);
}
__global__ void place_heads_kernel(const unsigned long long* __restrict__ lkey_s, const unsigned long long* __restrict__ flags, const unsigned long long* __restrict__ pos,
                                   long long n, unsigned long long* __restrict__ cell_key, long long* __restrict__ cell_ptr, unsigned long long* __restrict__ node_key,
                                   long long* __restrict__ node_cell_ptr, long long* __restrict__ node_ptr, unsigned long long* __restrict__ host_cells,
                                   unsigned long long* __restrict__ host_nodes) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const unsigned long long f = flags[q], ps = pos[q], k = lkey_s[q];
  const long long c = (long long)(ps & 0xffffffffull), j = (long long)(ps >> 32);
  if (f & 1ull) { cell_key[c] = k; cell_ptr[c] = q; }
  if (f >> 32) { node_key[j] = k >> FRAME_BITS; node_cell_ptr[j] = c; node_ptr[j] = q; }
  if (q == n - 1) {
    const long long nc = c + (long long)(f & 1ull), nn = j + (long long)(f >> 32);
    cell_ptr[nc] = n; node_cell_ptr[nn] = nc; node_ptr[nn] = n;
    host_cells[0] = (unsigned long long)nc; host_nodes[0] = (unsigned long long)nn;
  }
}

// a count the host is waiting for, written where it polls (mapped pinned memory): dst[0] = src_a[0] (+ src_b[0])
__global__ void post_count_kernel(const unsigned int* __restrict__ src_a, const unsigned int* __restrict__ src_b, unsigned long long* __restrict__ dst) {
  dst[0] = (unsigned long long)src_a[0] + (src_b ? (unsigned long long)src_b[0] : 0ull);
}
__global__ void post_count64_kernel(const long long* __restrict__ src, unsigned long long* __restrict__ dst) { dst[0] = (unsigned long long)src[0]; }


enum NodeState : unsigned char { DEAD = 0, FACTOR = 1, SUBDIVIDE = 2 };

// One thread per node of layer l: parent must have been subdivided; then loop_refine.hpp:358-405.
// (n_frames of node j = its number of (node, frame) cells = node_cell_ptr[j + 1] - node_cell_ptr[j]; flag[j] = 1 for an accepted node: what
// flag_kernel wrote in its own launch before round 6)
__global__ void judge_kernel(const unsigned long long* __restrict__ node_key, const double* __restrict__ node_cluster, const long long* __restrict__ node_cell_ptr,
                             long long n_nodes, int layer, VoxelizeParams p, const unsigned long long* __restrict__ parent_key,
                             const unsigned char* __restrict__ parent_state, long long n_parents, unsigned char* __restrict__ state,
                             double* __restrict__ eigval, double* __restrict__ eigvec, unsigned int* __restrict__ flag) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_nodes) return;
  unsigned char st = DEAD;
  bool alive = true;
  if (layer > 0) {
    // the parent's key: same root, octant bits of this layer cleared
    const unsigned long long pk = node_key[j] & ~(7ull << (3 * (3 - layer)));
    long long lo = 0, hi = n_parents - 1;
    alive = false;
    while (lo <= hi) {
      const long long mid = (lo + hi) >> 1;
      const unsigned long long v = parent_key[mid];
      if (v == pk) { alive = parent_state[mid] == SUBDIVIDE; break; }
      if (v < pk) lo = mid + 1; else hi = mid - 1;
    }
  }
  const double* c = node_cluster + 10 * j;
  const int minp = p.min_points_layer[layer] > 0 ? p.min_points_layer[layer] : p.min_points;
  if (alive && (int)c[9] > minp) {
    double Cm[6], lam[3], U[9];
    vxm::cluster_cov(c, c + 6, c[9], Cm);
    vxm::eig_sym3(Cm, lam, U);
    const bool is_plane = lam[0] < p.min_eigen_value && (lam[0] / lam[2]) < p.eigen_ratio[layer];
    if (is_plane) {
      if ((int)(node_cell_ptr[j + 1] - node_cell_ptr[j]) >= p.min_frames && !(lam[0] / lam[1] > p.factor_ratio_max)) st = FACTOR;
    } else if (layer < p.max_layer) {
      st = SUBDIVIDE;
    }
    for (int k = 0; k < 3; k++) eigval[3 * j + k] = lam[k];
    for (int col = 0; col < 3; col++)
      for (int row = 0; row < 3; row++) eigvec[9 * j + 3 * col + row] = U[3 * row + col];
  }
  state[j] = st;
  flag[j] = st == FACTOR ? 1u : 0u;
}

// accepted node j -> slot pos[j] of the AoS staging arrays [n][W][10], [n][3], [n][9], [n][10], ids
__global__ void emit_kernel(const unsigned char* __restrict__ state, const unsigned int* __restrict__ pos, long long n_nodes, int W, int layer,
                            const unsigned long long* __restrict__ node_key, const long long* __restrict__ node_cell_ptr,
                            const unsigned long long* __restrict__ cell_key, const double* __restrict__ cell_cluster,
                            const double* __restrict__ node_cluster, const double* __restrict__ eigval, const double* __restrict__ eigvec,
                            long long* d_total, long long capacity, double* __restrict__ o_clusters, double* __restrict__ o_eigval,
                            double* __restrict__ o_eigvec, double* __restrict__ o_merged, unsigned long long* __restrict__ o_id) {
  // d_total[0]: factor voxels emitted by the layers before this one (kept on the device since round 6: the host no longer waits for every
  // layer's count -- advance_total_kernel adds this layer's behind this kernel); d_total[1] != 0: the caller's capacity was exceeded
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_nodes || state[j] != FACTOR) return;
  const long long o = d_total[0] + pos[j];
  if (o >= capacity) { d_total[1] = 1; return; }
  double* oc = o_clusters + (size_t)o * W * 10;
  for (int k = 0; k < W * 10; k++) oc[k] = 0.0;
  for (long long cc = node_cell_ptr[j]; cc < node_cell_ptr[j + 1]; cc++) {
    const int f = (int)(cell_key[cc] & ((1ull << FRAME_BITS) - 1ull));
    for (int k = 0; k < 10; k++) oc[f * 10 + k] = cell_cluster[10 * cc + k];
  }
  for (int k = 0; k < 3; k++) o_eigval[3 * o + k] = eigval[3 * j + k];
  for (int k = 0; k < 9; k++) o_eigvec[9 * o + k] = eigvec[9 * j + k];
  for (int k = 0; k < 10; k++) o_merged[10 * o + k] = node_cluster[10 * j + k];
  const unsigned long long nk = node_key[j];                 // [root48 | path9]
  o_id[o] = ((nk >> PATH_BITS) << 16) | ((nk & 511ull) << 7) | (unsigned long long)layer;
}

__global__ void advance_total_kernel(const unsigned int* __restrict__ pos, const unsigned int* __restrict__ flag, long long n_nodes, long long* __restrict__ d_total) {
  d_total[0] += (long long)pos[n_nodes - 1] + (long long)flag[n_nodes - 1];
}
// compressed-row variant: entries of accepted node j at e_base + epos[j] ..
__global__ void entry_count_kernel(const unsigned char* __restrict__ state, const long long* __restrict__ node_cell_ptr, long long n_nodes, long long* __restrict__ cnt) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j <= n_nodes) cnt[j] = (j < n_nodes && state[j] == FACTOR) ? node_cell_ptr[j + 1] - node_cell_ptr[j] : 0ll;
}
__global__ void emit_csr_kernel(const unsigned char* __restrict__ state, const unsigned int* __restrict__ pos, const long long* __restrict__ epos, long long n_nodes,
                                int layer, const unsigned long long* __restrict__ node_key, const long long* __restrict__ node_cell_ptr,
                                const unsigned long long* __restrict__ cell_key, const double* __restrict__ cell_cluster,
                                const double* __restrict__ node_cluster, const double* __restrict__ eigval, const double* __restrict__ eigvec,
                                long long out_base, long long e_base, double* __restrict__ o_ecl, int* __restrict__ o_eframe, long long* __restrict__ o_row_ptr,
                                double* __restrict__ o_eigval, double* __restrict__ o_eigvec, double* __restrict__ o_merged, unsigned long long* __restrict__ o_id) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_nodes || state[j] != FACTOR) return;
  const long long o = out_base + pos[j];
  long long e = e_base + epos[j];
  for (long long cc = node_cell_ptr[j]; cc < node_cell_ptr[j + 1]; cc++, e++) {   // cells of a node: ascending frame (the key's low bits)
    o_eframe[e] = (int)(cell_key[cc] & ((1ull << FRAME_BITS) - 1ull));
    for (int k = 0; k < 10; k++) o_ecl[10 * e + k] = cell_cluster[10 * cc + k];
  }
  o_row_ptr[o + 1] = e;
  for (int k = 0; k < 3; k++) o_eigval[3 * o + k] = eigval[3 * j + k];
  for (int k = 0; k < 9; k++) o_eigvec[9 * o + k] = eigvec[9 * j + k];
  for (int k = 0; k < 10; k++) o_merged[10 * o + k] = node_cluster[10 * j + k];
  const unsigned long long nk = node_key[j];
  o_id[o] = ((nk >> PATH_BITS) << 16) | ((nk & 511ull) << 7) | (unsigned long long)layer;
}

__global__ void fill_kernel(double* __restrict__ p, long long n, double v) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) p[q] = v;
}
void fill(double* d, long long n, double v, hipStream_t s) {
  if (n > 0) fill_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(d, n, v);
}

// One device allocation for all temporaries of a call (hipMalloc/hipFree cost ~0.1-1 ms each; there are ~30 arrays), kept per
// device between calls and only ever grown (a window is re-voxelised every BA round); calls on one device take turns on it.
struct ArenaSlot {
  std::mutex mtx;
  char* base = nullptr;
  size_t cap = 0;
  unsigned long long* pinned = nullptr;   // READBACK_WORDS words of pinned host memory: where the counts a call needs on the host come down
};
constexpr int READBACK_WORDS = 16;
constexpr int ARENA_SLOTS = 8;   // per device: vxba_hba_pass drives up to eight host threads / streams that voxelise concurrently
static ArenaSlot g_arena[16][ARENA_SLOTS];
// vxba_voxelize_profile: time of / bytes moved by the cluster-build kernel inside the voxeliser (the dominant kernel of a hierarchical-BA pass)
struct K1Prof { std::atomic<int> on{0}; std::mutex m; double ms = 0, bytes = 0; long long launches = 0; };
static K1Prof g_k1prof;
// development: VXBA_VOXELIZE_WAITS=1 prints, at exit, how long the voxeliser's host threads spent waiting for the counts
struct WaitProf {
  bool on = false;
  std::atomic<long long> us{0}, n{0}, slow{0}, max_us{0}, alloc_us{0}, allocs{0}, alloc_max_us{0}, call_us{0}, calls{0};
  WaitProf() { const char* e = getenv("VXBA_VOXELIZE_WAITS"); on = e && e[0] == '1'; }
  ~WaitProf() {
    if (on) fprintf(stderr, "[vxba voxelize] %lld calls %.1f ms | %lld waits, %.1f ms in them, %lld over 1 ms, longest %lld us | %lld (re)allocations, %.1f ms in them, longest %lld us\n",
                    calls.load(), call_us.load() / 1e3, n.load(), us.load() / 1e3, slow.load(), max_us.load(), allocs.load(), alloc_us.load() / 1e3, alloc_max_us.load());
  }
  void note_alloc(long long t) { alloc_us.fetch_add(t); allocs.fetch_add(1); long long mx = alloc_max_us.load(); while (t > mx && !alloc_max_us.compare_exchange_weak(mx, t)) {} }
};
struct ScopeUs {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  long long us() const { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
static WaitProf g_wait_prof;
struct DevBuf {
  char* base = nullptr;
  size_t cap = 0, used = 0;
  ArenaSlot* slot = nullptr;
  bool own = false;
  unsigned long long* host = nullptr;     // pinned read-back words (the slot's, or this call's own)
  ~DevBuf() {
    ScopeUs tm;
    if (own && base) hipFree(base);
    if (own && host) hipHostFree(host);
    if (own && g_wait_prof.on) g_wait_prof.note_alloc(tm.us());
    if (slot) slot->mtx.unlock();
  }
  hipError_t reserve(size_t bytes) {
    int dev = 0;
    // Round 6: the limit above which a call allocates privately went from 2 GiB to 64: the top level of a hierarchical pass asks for 6.2 GB
    // four times per pass, and one hipMalloc / hipFree pair of that size in ~30 takes SECONDS on the gpurun boxes (3.55 s measured, against
    // 0.5 ms for the others: VXBA_VOXELIZE_WAITS=1) -- most of the "slow box" passes of rounds 5-6.  A slot that has seen such a call keeps
    // its 7.8 GB; 288 GB of HBM can afford it.
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || bytes > ((size_t)64 << 30)) {   // no cache slot / too big to keep: private to this call
      own = true; cap = bytes;
      ScopeUs tm;
      hipError_t e = hipHostMalloc((void**)&host, READBACK_WORDS * sizeof(unsigned long long), hipHostMallocDefault);
      if (e != hipSuccess) return e;
      e = hipMalloc((void**)&base, bytes);
      if (g_wait_prof.on) g_wait_prof.note_alloc(tm.us());
      return e;
    }
    for (int k = 0; k < ARENA_SLOTS && !slot; k++)
      if (g_arena[dev][k].mtx.try_lock()) slot = &g_arena[dev][k];
    if (!slot) { slot = &g_arena[dev][0]; slot->mtx.lock(); }
    if (bytes > slot->cap) {
      ScopeUs tm;
      if (slot->base) { hipDeviceSynchronize(); hipFree(slot->base); }
      slot->base = nullptr; slot->cap = 0;
      const size_t want = bytes + bytes / 4;
      hipError_t e = hipMalloc((void**)&slot->base, want);
      if (g_wait_prof.on) g_wait_prof.note_alloc(tm.us());
      if (e != hipSuccess) return e;
      slot->cap = want;
    }
    if (!slot->pinned) {
      hipError_t e = hipHostMalloc((void**)&slot->pinned, READBACK_WORDS * sizeof(unsigned long long), hipHostMallocDefault);
      if (e != hipSuccess) return e;
    }
    base = slot->base; cap = slot->cap; host = slot->pinned;
    return hipSuccess;
  }
  // A count the host needs (launch sizes, capacities): copied into pinned word k behind the work queued on s -- a copy into PAGEABLE memory
  // (a stack variable) is staged and waited for inside the runtime, and hipStreamSynchronize parks the thread; a hierarchical-BA pass makes
  // ~2600 such round trips, from several host threads.  wait(): spin on the stream's state instead.
  hipError_t fetch(int k, const void* d_src, size_t bytes, hipStream_t s) {
    host[k] = 0;
    return hipMemcpyAsync(host + k, d_src, bytes, hipMemcpyDeviceToHost, s);
  }
  // Round 6: the same counts WITHOUT the copy engine -- the kernel that produces a count (or a one-thread kernel behind it) stores it into
  // the mapped pinned word itself; a hierarchical-BA pass made 1 870 four-byte copies of ~4 us each.  (hipHostMalloc'd memory is mapped into
  // the device's address space at the same address; the store is visible to the host when the stream has drained, which is what wait() polls.)
  unsigned int* word_u32(int k) { host[k] = 0; return reinterpret_cast<unsigned int*>(host + k); }
  unsigned long long* word_u64(int k) { host[k] = 0; return host + k; }
  hipError_t post_sum_u32(int k, const unsigned int* a, const unsigned int* b, hipStream_t s) {
    host[k] = 0;
    post_count_kernel<<<1, 1, 0, s>>>(a, b, host + k);
    return hipGetLastError();
  }
  hipError_t post_i64(int k, const long long* a, hipStream_t s) {
    host[k] = 0;
    post_count64_kernel<<<1, 1, 0, s>>>(a, host + k);
    return hipGetLastError();
  }
  static hipError_t wait(hipStream_t s) {
    hipError_t q;
    if (g_wait_prof.on) {
      const auto t0 = std::chrono::steady_clock::now();
      q = vxwait::stream_wait(s);
      const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
      g_wait_prof.us.fetch_add(us); g_wait_prof.n.fetch_add(1);
      if (us > 1000) g_wait_prof.slow.fetch_add(1);
      long long mx = g_wait_prof.max_us.load();
      while (us > mx && !g_wait_prof.max_us.compare_exchange_weak(mx, us)) {}
      return q;
    }
    q = vxwait::stream_wait(s);
    return q;
  }
  unsigned int u32(int k) const { return (unsigned int)(*(volatile unsigned long long*)(host + k) & 0xffffffffull); }
  long long i64(int k) const { return (long long)*(volatile unsigned long long*)(host + k); }
  template <class T>
  hipError_t alloc(T** p, size_t n) {
    const size_t bytes = (((n > 0 ? n : 1) * sizeof(T)) + 255) & ~(size_t)255;
    if (used + bytes > cap) return hipErrorOutOfMemory;
    *p = (T*)(base + used);
    used += bytes;
    return hipSuccess;
  }
};

#define VV(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { *err_out = hipGetErrorString(e_); return -1; } } while (0)

static inline unsigned grid_for(long long n, int b = 256) { return (unsigned)((n + b - 1) / b); }
// rocPRIM's radix sort with the merge-sort path switched off (MergeSortLimit = 0): below 1 M keys its default is a block sort + log2(n / block)
// merge passes whatever the key width; the compact layer-0 key (relkey_kernel) wants a histogram + ceil(bits / 8) onesweep passes
using OnesweepAlways = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

long long voxelize(int W, long long n_points, const double* d_xyz_local, const long long* d_frame_ptr, const double* d_poses, const VoxelizeParams& p,
                   hipStream_t s, VoxelizeOutput* out, const char** err_out) {
  static const char* range_msg = "voxelize: a point lies outside the +-32768-voxel range of the 16-bit voxel coordinates";
  static const char* cap_msg = "voxelize: more factor voxels than the caller's capacity";
  DevBuf B;
  struct CallTimer { ScopeUs tm; ~CallTimer() { if (g_wait_prof.on) { g_wait_prof.call_us.fetch_add(tm.us()); g_wait_prof.calls.fetch_add(1); } } } call_timer;
  long long n = n_points;
  const bool sharded = p.shard_count > 1;
  if (sharded && (p.shard_index < 0 || p.shard_index >= p.shard_count)) { *err_out = "voxelize: shard_index outside 0 .. shard_count-1"; return -1; }
  // rocPRIM temporary storage: query the largest need first (size queries do not touch the pointers)
  size_t tb = 0;
  {
    size_t tb_sort = 0, tb_scan = 0, tb_scan32 = 0;
    unsigned long long* k0 = nullptr; unsigned int* v0 = nullptr; long long* l0 = nullptr;
    VV(rocprim::radix_sort_pairs(nullptr, tb_sort, k0, k0, v0, v0, (size_t)n, 0, 64, s));
    VV(rocprim::exclusive_scan(nullptr, tb_scan, l0, l0, 0ll, (size_t)n + 1, rocprim::plus<long long>(), s));
    VV(rocprim::exclusive_scan(nullptr, tb_scan32, v0, v0, 0u, (size_t)n, rocprim::plus<unsigned int>(), s));
    { size_t tb_scan64 = 0; VV(rocprim::exclusive_scan(nullptr, tb_scan64, k0, k0, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), s)); if (tb_scan64 > tb_scan) tb_scan = tb_scan64; }
    { size_t tb_rel = 0; VV(rocprim::radix_sort_pairs<OnesweepAlways>(nullptr, tb_rel, v0, v0, v0, v0, (size_t)n, 0, 30, s)); if (tb_rel > tb_sort) tb_sort = tb_rel; }
    tb = tb_sort;
    if (tb_scan > tb) tb = tb_scan;
    if (tb_scan32 > tb) tb = tb_scan32;
  }
  // worst case (every point its own cell): ~0.5 KB of scratch per point
  VV(B.reserve((size_t)(n + 64) * (sharded ? 608 : 544) + tb + (size_t)(1 << 20)));   // (+32 B per point in round 6: a layer's node keys are sized by the points, the node count arrives with them)
  double *d_world, *d_loc_s, *d_wld_s;
  unsigned long long *d_key, *d_lkey, *d_lkey_s;
  unsigned int *d_idx, *d_idx_s;
  int* d_err;
  VV(B.alloc(&d_world, 3 * n)); VV(B.alloc(&d_loc_s, 3 * n)); VV(B.alloc(&d_wld_s, 3 * n));
  VV(B.alloc(&d_key, n)); VV(B.alloc(&d_lkey, n)); VV(B.alloc(&d_lkey_s, n));
  VV(B.alloc(&d_idx, n)); VV(B.alloc(&d_idx_s, n)); VV(B.alloc(&d_err, 1));
  int* d_bbox;
  VV(B.alloc(&d_bbox, 6));
  init_err_bbox_kernel<<<1, 64, 0, s>>>(d_err, d_bbox);
  if (n > 0) key_kernel<<<grid_for(n), 256, 0, s>>>(d_xyz_local, d_frame_ptr, W, d_poses, p, n, d_world, d_key, d_err, d_bbox);

  // per-layer scratch (sized for the worst case: every point its own cell)
  unsigned long long *d_cell_key, *d_cell_node, *d_node_key[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned int *d_flag, *d_pos;
  long long *d_cell_ptr, *d_node_cell_ptr, *d_node_ptr, *d_tmp64;
  double *d_cell_cl, *d_node_cl, *d_eigval, *d_eigvec;
  unsigned char* d_state[4] = {nullptr, nullptr, nullptr, nullptr};
  VV(B.alloc(&d_cell_key, n)); VV(B.alloc(&d_cell_node, n));   // (d_cell_node: the scanned head flags of a layer -- cell position in the low word, node position in the high one)
  VV(B.alloc(&d_flag, n)); VV(B.alloc(&d_pos, n));
  VV(B.alloc(&d_cell_ptr, n + 1)); VV(B.alloc(&d_node_cell_ptr, n + 1)); VV(B.alloc(&d_node_ptr, n + 1)); VV(B.alloc(&d_tmp64, n + 1));
  VV(B.alloc(&d_cell_cl, 10 * n)); VV(B.alloc(&d_node_cl, 10 * n)); VV(B.alloc(&d_eigval, 3 * n)); VV(B.alloc(&d_eigvec, 9 * n));
  long long n_nodes_l[4] = {0, 0, 0, 0};

  char* d_temp;
  VV(B.alloc(&d_temp, tb));

  B.word_u64(0);
  for (int k = 8; k < 14; k++) B.word_u64(k);
  post_err_bbox_kernel<<<1, 64, 0, s>>>(d_err, d_bbox, B.host, B.host + 8);
  VV(hipGetLastError());
  VV(B.wait(s));
  if (B.u32(0)) { *err_out = range_msg; return -1; }
  // the compact layer-0 key: bits per axis from the cloud's bounding box (all points' box: a shard's points lie inside it)
  int rel_min[3] = {0, 0, 0}, rel_bits[3] = {0, 0, 0}, rel_total = 64;
  if (n > 0) {
    rel_total = 0;
    for (int j = 0; j < 3; j++) {
      const long long lo = B.i64(8 + j), hi = B.i64(11 + j);
      rel_min[j] = (int)lo;
      int b = 1;
      while (((long long)1 << b) < hi - lo + 1) b++;
      rel_bits[j] = b;
      rel_total += b;
    }
  }
  if (sharded && n > 0) {
    // this shard's points to the front, cloud order kept (the cluster sums below are in point order): everything behind works on the n_keep
    // points of whole root voxels, so each of its factor voxels is bit for bit the one the unsharded run produces
    double *d_loc_c, *d_wld_c;
    unsigned long long* d_key_c;
    VV(B.alloc(&d_loc_c, 3 * n)); VV(B.alloc(&d_wld_c, 3 * n)); VV(B.alloc(&d_key_c, n));
    shard_flag_kernel<<<grid_for(n), 256, 0, s>>>(d_key, n, p.shard_index, p.shard_count, d_flag);
    size_t t = tb;
    VV(rocprim::exclusive_scan(d_temp, t, d_flag, d_pos, 0u, (size_t)n, rocprim::plus<unsigned int>(), s));
    VV(B.post_sum_u32(1, d_pos + (n - 1), d_flag + (n - 1), s));
    shard_compact_kernel<<<grid_for(n), 256, 0, s>>>(d_flag, d_pos, n, d_xyz_local, d_world, d_key, d_loc_c, d_wld_c, d_key_c);
    VV(B.wait(s));
    n = B.i64(1);
    d_xyz_local = d_loc_c; d_world = d_wld_c; d_key = d_key_c;
    if (n == 0) { out->n_entries = 0; return 0; }
  }

  // development knob (same-box A/B): VXBA_VOXELIZE_PARTITION=0 sorts every layer as rounds 1-5 did
  static const bool use_partition = [] { const char* e = getenv("VXBA_VOXELIZE_PARTITION"); return !(e && e[0] == '0'); }();
  static const bool use_compact = [] { const char* e = getenv("VXBA_VOXELIZE_COMPACT_KEY"); return !(e && e[0] == '0'); }();
  constexpr long long PARTITION_MAX_POINTS = 1ll << 20;   // beyond this rocPRIM sorts with onesweep, and one node can be a workgroup's millisecond
  long long total = 0, total_entries = 0;
  long long* d_epos = nullptr;
  long long* d_total = nullptr;   // [emitted so far | capacity exceeded]: the dense-row path's running total (see emit_kernel)
  VV(B.alloc(&d_total, 2));
  VV(hipMemsetAsync(d_total, 0, 2 * sizeof(long long), s));
  if (out->d_row_ptr) {
    VV(B.alloc(&d_epos, n + 1));
    VV(hipMemsetAsync(out->d_row_ptr, 0, sizeof(long long), s));
  }
  for (int layer = 0; layer <= p.max_layer && n > 0; layer++) {
    size_t t = tb;
    if (layer > 0 && use_partition && n <= PARTITION_MAX_POINTS) {
      // this layer's order = the previous layer's with every node's range stably partitioned by the new octant (see the header)
      partition_kernel<<<(unsigned)n_nodes_l[layer - 1], PART_BLOCK, 0, s>>>(d_key, d_idx_s, d_node_ptr, layer, d_idx, d_lkey);
      { unsigned int* ti = d_idx; d_idx = d_idx_s; d_idx_s = ti; }
      { unsigned long long* tk = d_lkey; d_lkey = d_lkey_s; d_lkey_s = tk; }
    } else if (layer == 0 && use_compact && rel_total <= 30 && n > 4096) {
      // layer 0 by the compact root key (relkey_kernel): d_flag / d_pos are free until the plane test and hold the keys
      relkey_kernel<<<grid_for(n), 256, 0, s>>>(d_key, n, rel_min[0], rel_min[1], rel_min[2], rel_bits[1], rel_bits[2], d_flag, d_idx);
      VV(rocprim::radix_sort_pairs<OnesweepAlways>(d_temp, t, d_flag, d_pos, d_idx, d_idx_s, (size_t)n, 0, (unsigned)rel_total, s));
      lkey_gather_kernel<<<grid_for(n), 256, 0, s>>>(d_key, d_idx_s, n, layer, d_lkey_s);
    } else {
      layer_key_kernel<<<grid_for(n), 256, 0, s>>>(d_key, n, layer, d_lkey, d_idx);
      VV(rocprim::radix_sort_pairs(d_temp, t, d_lkey, d_lkey_s, d_idx, d_idx_s, (size_t)n, 0, 64, s));
    }
    // (node, frame) cells and nodes: head flags, one scan, the heads place themselves (see heads_kernel)
    heads_kernel<<<grid_for(n), 256, 0, s>>>(d_lkey_s, n, (unsigned long long*)d_tmp64);
    t = tb;
    VV(rocprim::exclusive_scan(d_temp, t, (unsigned long long*)d_tmp64, d_cell_node, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), s));
    VV(B.alloc(&d_node_key[layer], n));
    {
      unsigned long long* hc = B.word_u64(3);
      unsigned long long* hn = B.word_u64(4);
      place_heads_kernel<<<grid_for(n), 256, 0, s>>>(d_lkey_s, (const unsigned long long*)d_tmp64, d_cell_node, n, d_cell_key, d_cell_ptr, d_node_key[layer], d_node_cell_ptr,
                                                     d_node_ptr, hc, hn);
    }
    VV(B.wait(s));
    const unsigned int n_cells = B.u32(3);
    const unsigned int n_nodes = B.u32(4);
    n_nodes_l[layer] = n_nodes;
    // clusters: body-frame per cell, world per node -- sequential sums in upstream's push order
    gather3_kernel<<<grid_for(n), 256, 0, s>>>(d_xyz_local, d_idx_s, n, d_loc_s);
    gather3_kernel<<<grid_for(n), 256, 0, s>>>(d_world, d_idx_s, n, d_wld_s);
    if (g_k1prof.on.load(std::memory_order_relaxed)) {
      // measurement mode (vxba_voxelize_profile): the two cluster builds of this layer bracketed by events bound to the dispatches themselves.
      // Algorithmic bytes of a build: 24 B per point + 8 B per cell offset read, 80 B per cluster written.
      hipEvent_t ev[4];
      for (hipEvent_t& e : ev) VV(hipEventCreate(&e));
      vxk::launch_k1_build_aos(d_loc_s, (const int64_t*)d_cell_ptr, n_cells, d_cell_cl, s, ev[0], ev[1]);
      vxk::launch_k1_build_aos(d_wld_s, (const int64_t*)d_node_ptr, n_nodes, d_node_cl, s, ev[2], ev[3]);
      VV(hipStreamSynchronize(s));
      float ms0 = 0, ms1 = 0;
      VV(hipEventElapsedTime(&ms0, ev[0], ev[1]));
      VV(hipEventElapsedTime(&ms1, ev[2], ev[3]));
      for (hipEvent_t& e : ev) hipEventDestroy(e);
      std::lock_guard<std::mutex> lk(g_k1prof.m);
      g_k1prof.ms += (double)ms0 + (double)ms1;
      g_k1prof.launches += 2;
      g_k1prof.bytes += 2.0 * 24.0 * (double)n + 8.0 * ((double)n_cells + (double)n_nodes + 2.0) + 80.0 * ((double)n_cells + (double)n_nodes);
    } else {
      vxk::launch_k1_build_aos(d_loc_s, (const int64_t*)d_cell_ptr, n_cells, d_cell_cl, s);
      vxk::launch_k1_build_aos(d_wld_s, (const int64_t*)d_node_ptr, n_nodes, d_node_cl, s);
    }
    // verdicts
    VV(B.alloc(&d_state[layer], n_nodes));
    judge_kernel<<<grid_for(n_nodes), 256, 0, s>>>(d_node_key[layer], d_node_cl, d_node_cell_ptr, n_nodes, layer, p, layer ? d_node_key[layer - 1] : nullptr,
                                                  layer ? d_state[layer - 1] : nullptr, layer ? n_nodes_l[layer - 1] : 0, d_state[layer], d_eigval, d_eigvec, d_flag);
    t = tb;
    VV(rocprim::exclusive_scan(d_temp, t, d_flag, d_pos, 0u, (size_t)n_nodes, rocprim::plus<unsigned int>(), s));
    if (!out->d_row_ptr) {
      // dense rows: no host round trip for this layer's count -- the running total stays on the device, the emit kernel guards the capacity
      if (n_nodes > 0) {
        emit_kernel<<<grid_for(n_nodes), 256, 0, s>>>(d_state[layer], d_pos, n_nodes, W, layer, d_node_key[layer], d_node_cell_ptr, d_cell_key, d_cell_cl, d_node_cl,
                                                     d_eigval, d_eigvec, d_total, out->capacity, out->d_clusters, out->d_eigval, out->d_eigvec, out->d_merged, out->d_node_id);
        advance_total_kernel<<<1, 1, 0, s>>>(d_pos, d_flag, n_nodes, d_total);
      }
      continue;
    }
    long long n_acc = 0;
    if (n_nodes > 0) {
      VV(B.post_sum_u32(5, d_pos + n_nodes - 1, d_flag + n_nodes - 1, s));
      VV(B.wait(s));
      n_acc = B.i64(5);
    }
    if (total + n_acc > out->capacity) { *err_out = cap_msg; return -1; }
    if (n_acc > 0) {
      entry_count_kernel<<<grid_for((long long)n_nodes + 1), 256, 0, s>>>(d_state[layer], d_node_cell_ptr, n_nodes, d_tmp64);
      t = tb;
      VV(rocprim::exclusive_scan(d_temp, t, d_tmp64, d_epos, 0ll, (size_t)n_nodes + 1, rocprim::plus<long long>(), s));
      VV(B.post_i64(7, d_epos + n_nodes, s));
      VV(B.wait(s));
      const long long n_ent = B.i64(7);
      if (total_entries + n_ent > out->ecap) { *err_out = cap_msg; return -1; }
      emit_csr_kernel<<<grid_for(n_nodes), 256, 0, s>>>(d_state[layer], d_pos, d_epos, n_nodes, layer, d_node_key[layer], d_node_cell_ptr, d_cell_key, d_cell_cl,
                                                       d_node_cl, d_eigval, d_eigvec, total, total_entries, out->d_clusters, out->d_eframe, out->d_row_ptr,
                                                       out->d_eigval, out->d_eigvec, out->d_merged, out->d_node_id);
      total_entries += n_ent;
    }
    total += n_acc;
  }
  out->n_entries = total_entries;
  if (!out->d_row_ptr && n > 0) {
    VV(B.post_i64(5, d_total, s));
    VV(B.post_i64(6, d_total + 1, s));
    VV(B.wait(s));
    if (B.i64(6)) { *err_out = cap_msg; return -1; }
    total = B.i64(5);
  } else {
    VV(B.wait(s));
  }
  VV(hipGetLastError());
  return total;
}

}  // namespace vxv

// Measurement (bench.py --config cfg5): enable != 0 starts a fresh measurement of the cluster-build kernel inside the voxeliser (every launch bracketed by
// events bound to its dispatch, one stream synchronisation per layer: NOT for timed runs); enable == 0 stops it and returns the sums.
extern "C" int vxba_voxelize_profile(int enable, double* ms_sum, long long* launches, double* algorithmic_bytes) {
  std::lock_guard<std::mutex> lk(vxv::g_k1prof.m);
  if (enable) { vxv::g_k1prof.ms = 0; vxv::g_k1prof.bytes = 0; vxv::g_k1prof.launches = 0; vxv::g_k1prof.on.store(1); return 0; }
  vxv::g_k1prof.on.store(0);
  if (ms_sum) *ms_sum = vxv::g_k1prof.ms;
  if (launches) *launches = vxv::g_k1prof.launches;
  if (algorithmic_bytes) *algorithmic_bytes = vxv::g_k1prof.bytes;
  return 0;
}
