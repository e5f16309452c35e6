// Voxel-grid down-sampling on the GPU: `down_sampling_voxel` (VoxelSLAM/src/tools.hpp:201-238), the filter the reference runs on
// every raw scan before the odometry (voxelslam.cpp:1236, 1577-1583) and on every merged submap of the hierarchical BA (:2447).
// Upstream: an unordered_map from voxel index to a running mean, updated point by point in cloud order, in float.  Here: one key
// per point (upstream's float-typed voxel index), one stable radix sort, run-length encode + scan for the cell table, and one lane
// per occupied voxel that replays the running-mean recurrence over its points in cloud order with unfused float arithmetic -- the
// means are bit-identical to upstream's.  Output order is ascending (x, y, z) voxel index (upstream: hash-map iteration order,
// which is implementation-defined).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/vxba.h"

namespace vxd {

constexpr long long OFF = 1ll << 20;   // voxel indices in [-2^20, 2^20)

__global__ void ds_key_kernel(const float* __restrict__ xyz, long long n, double voxel_size, unsigned long long* __restrict__ key, unsigned int* __restrict__ idx,
                              int* __restrict__ err) {
#pragma clang fp contract(off)
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  unsigned long long k = 0;
  for (int j = 0; j < 3; j++) {
    float loc = (float)((double)xyz[3 * q + j] / voxel_size);   // loc_xyz[j] = p_c.data[j] / voxel_size   (:211)
    if (loc < 0) loc = (float)((double)loc - 1.0);              // loc_xyz[j] -= 1.0                         (:212-213)
    const long long pos = (long long)loc;
    if (pos < -OFF || pos >= OFF) { *err = 1; return; }
    k = (k << 21) | (unsigned long long)(pos + OFF);
  }
  key[q] = k;
  idx[q] = (unsigned int)q;
}

// pp = (pp * curvature + p) / (curvature + 1), curvature += 1   (:227-230), float, unfused, in cloud order
__global__ void ds_mean_kernel(const float* __restrict__ xyz, const unsigned int* __restrict__ idx_sorted, const long long* __restrict__ cell_ptr, long long n_cells,
                               float* __restrict__ out) {
#pragma clang fp contract(off)
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cells) return;
  long long q = cell_ptr[c];
  const long long end = cell_ptr[c + 1];
  unsigned int i = idx_sorted[q];
  float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2], cur = 1.0f;
  for (q++; q < end; q++) {
    i = idx_sorted[q];
    const float d = cur + 1.0f;
    x = (x * cur + xyz[3 * (size_t)i]) / d;
    y = (y * cur + xyz[3 * (size_t)i + 1]) / d;
    z = (z * cur + xyz[3 * (size_t)i + 2]) / d;
    cur = d;
  }
  out[3 * c] = x; out[3 * c + 1] = y; out[3 * c + 2] = z;
}

__global__ void ds_widen_kernel(const unsigned int* __restrict__ cnt, long long n, long long* __restrict__ out) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) out[q] = (long long)cnt[q];
  if (q == n) out[q] = 0;
}

}  // namespace vxd

extern "C" int vxba_down_sampling_voxel(int device, int64_t n, const float* xyz, double voxel_size, float* out_xyz, int64_t* n_out) {
  if (n < 0 || !n_out || (n > 0 && (!xyz || !out_xyz)) || n >= (1ll << 32)) return VXBA_ERR_ARG;
  *n_out = 0;
  if (n == 0) return VXBA_OK;
  if (voxel_size < 0.001) {   // upstream returns the cloud untouched (:203)
    std::memcpy(out_xyz, xyz, (size_t)n * 3 * sizeof(float));
    *n_out = n;
    return VXBA_OK;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_HIP;
  hipStream_t s = nullptr;
  float *d_xyz = nullptr, *d_out = nullptr;
  unsigned long long *d_key = nullptr, *d_key_s = nullptr, *d_ukey = nullptr;
  unsigned int *d_idx = nullptr, *d_idx_s = nullptr, *d_cnt = nullptr, *d_runs = nullptr;
  long long* d_ptr = nullptr;
  int* d_err = nullptr;
  void* d_temp = nullptr;
  size_t t_sort = 0, t_rle = 0, t_scan = 0;
  int rc = VXBA_OK;
  unsigned int runs = 0;
  int err = 0;
  const unsigned grid = (unsigned)((n + 255) / 256), grid1 = (unsigned)((n + 256) / 256);
  // one grow-only scratch per device for this stand-alone entry point (it runs once per scan: a dozen hipMalloc / hipFree pairs per
  // call cost more than the sort); calls are serialised on it
  static std::mutex mtx;
  static char* scratch[16] = {nullptr};
  static size_t scratch_cap[16] = {0};
  std::lock_guard<std::mutex> lock(mtx);
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
#define DS(call) do { if ((call) != hipSuccess) { rc = VXBA_ERR_HIP; goto done; } } while (0)
  DS(rocprim::radix_sort_pairs(nullptr, t_sort, d_key, d_key_s, d_idx, d_idx_s, (size_t)n, 0, 63, s));
  DS(rocprim::run_length_encode(nullptr, t_rle, d_key_s, (size_t)n, d_ukey, d_cnt, d_runs, s));
  DS(rocprim::exclusive_scan(nullptr, t_scan, d_ptr, d_ptr, 0ll, (size_t)n + 1, rocprim::plus<long long>(), s));
  {
    size_t tmax = t_sort > t_rle ? t_sort : t_rle;
    if (t_scan > tmax) tmax = t_scan;
    const size_t b_f = up((size_t)n * 3 * sizeof(float)), b_k = up((size_t)n * 8), b_i = up((size_t)n * 4), b_p = up((size_t)(n + 1) * 8), b_t = up(tmax ? tmax : 8);
    const size_t need = 2 * b_f + 3 * b_k + 3 * b_i + 256 + b_p + 256 + b_t;
    const int dv = device < 16 ? device : 15;
    if (need > scratch_cap[dv] || device >= 16) {
      if (scratch[dv]) { hipDeviceSynchronize(); hipFree(scratch[dv]); }
      scratch[dv] = nullptr; scratch_cap[dv] = 0;
      DS(hipMalloc((void**)&scratch[dv], need + need / 4));
      scratch_cap[dv] = need + need / 4;
    }
    char* q = scratch[dv];
    auto carve = [&](size_t bytes) { char* r = q; q += bytes; return r; };
    d_xyz = (float*)carve(b_f); d_out = (float*)carve(b_f);
    d_key = (unsigned long long*)carve(b_k); d_key_s = (unsigned long long*)carve(b_k); d_ukey = (unsigned long long*)carve(b_k);
    d_idx = (unsigned int*)carve(b_i); d_idx_s = (unsigned int*)carve(b_i); d_cnt = (unsigned int*)carve(b_i);
    d_runs = (unsigned int*)carve(256); d_ptr = (long long*)carve(b_p); d_err = (int*)carve(256); d_temp = carve(b_t);
  }
  DS(hipMemset(d_err, 0, 4));
  {
    size_t t = t_sort > t_rle ? t_sort : t_rle;
    if (t_scan > t) t = t_scan;
    if (!t) t = 8;
    DS(hipMemcpy(d_xyz, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
    vxd::ds_key_kernel<<<grid, 256, 0, s>>>(d_xyz, n, voxel_size, d_key, d_idx, d_err);
    DS(hipGetLastError());
    DS(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
    if (err) { rc = VXBA_ERR_ARG; goto done; }   // a voxel index outside [-2^20, 2^20)
    size_t tt = t;
    DS(rocprim::radix_sort_pairs(d_temp, tt, d_key, d_key_s, d_idx, d_idx_s, (size_t)n, 0, 63, s));
    tt = t;
    DS(rocprim::run_length_encode(d_temp, tt, d_key_s, (size_t)n, d_ukey, d_cnt, d_runs, s));
    DS(hipMemcpy(&runs, d_runs, 4, hipMemcpyDeviceToHost));
    vxd::ds_widen_kernel<<<grid1, 256, 0, s>>>(d_cnt, (long long)runs, d_ptr);
    DS(hipGetLastError());
    tt = t;
    DS(rocprim::exclusive_scan(d_temp, tt, d_ptr, d_ptr, 0ll, (size_t)runs + 1, rocprim::plus<long long>(), s));
    vxd::ds_mean_kernel<<<(unsigned)((runs + 63) / 64), 64, 0, s>>>(d_xyz, d_idx_s, d_ptr, (long long)runs, d_out);
    DS(hipGetLastError());
    DS(hipMemcpy(out_xyz, d_out, (size_t)runs * 3 * sizeof(float), hipMemcpyDeviceToHost));
    *n_out = (int64_t)runs;
  }
done:
#undef DS
  return rc;
}
