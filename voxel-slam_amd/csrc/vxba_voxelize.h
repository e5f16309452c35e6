// Launch interface of the batch factor construction (vxba_voxelize.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "vxba_kernels.h"

namespace vxv {

struct VoxelizeParams {
  double voxel_size;
  int max_layer;               // 0..3
  int min_points;              // a node needs N > min_points
  double min_eigen_value;      // plane_judge: lambda0 < min_eigen_value ...
  double eigen_ratio[4];       // ... && lambda0 / lambda2 < eigen_ratio[layer]
  double factor_ratio_max;     // factor filter: lambda0 / lambda1 <= factor_ratio_max
  int min_points_layer[4];     // > 0: per-layer override of min_points (OctoTree's min_point[layer])
  int min_frames;              // observing frames a factor needs (OctreeGBA: 2, OctoTree: 0)
  int shard_index, shard_count; // shard_count > 1: keep only the points whose ROOT voxel hashes to shard_index (vxv::root_shard) -- a rank's share of a
                                // voxel-sharded window (whole root voxels, so every factor voxel of the unsharded run appears on exactly one shard, bit for bit)
};

// Device staging arrays for the accepted voxels (AoS, the formats of vxba_push_voxels), capacity in voxels.
struct VoxelizeOutput {
  long long capacity;
  double* d_clusters;          // [n][W][10]
  double* d_eigval;            // [n][3]
  double* d_eigvec;            // [n][9] column-major
  double* d_merged;            // [n][10]
  unsigned long long* d_node_id;   // [n] canonical id: [x:16 | y:16 | z:16 | path:9 | pad:4 | layer:3]
  // Compressed rows instead of the dense [n][W][10] block (wide windows: a voxel is seen from a handful of ~100 frames): when
  // d_row_ptr is set, d_clusters is [ecap][10] over the observed (voxel, frame) entries, voxel by voxel with ascending frames,
  // d_row_ptr[n + 1] (d_row_ptr[0] == 0) and d_eframe[ecap] describe them, and n_entries receives their number.  A point belongs to
  // at most one factor voxel, so ecap = n_points is always enough.
  long long* d_row_ptr = nullptr;
  int* d_eframe = nullptr;
  long long ecap = 0;
  long long n_entries = 0;
};

// Returns the number of factor voxels written (grouped by layer, ascending node key inside a layer) or -1 (*err set).
// shard of a root voxel: [x:16 | y:16 | z:16] (coordinates offset by 32768) -> 0 .. count-1.  Fibonacci hash of the 48-bit root, so that the
// spatially contiguous roots of a scene spread evenly.  voxel_slam_amd.dist.root_shard is the same function in numpy.
inline __host__ __device__ int root_shard(unsigned long long root48, int count) {
  return (int)(((root48 * 0x9E3779B97F4A7C15ull) >> 32) % (unsigned long long)count);
}
long long voxelize(int W, long long n_points, const double* d_xyz_local, const long long* d_frame_ptr, const double* d_poses /* W*12 */, const VoxelizeParams& p,
                   hipStream_t s, VoxelizeOutput* out, const char** err);

void fill(double* d, long long n, double v, hipStream_t s);

}  // namespace vxv
