// Voxel-grid down-sampling on the GPU: `down_sampling_voxel` (VoxelSLAM/src/tools.hpp:201-238), the filter the reference runs on
// every raw scan before the odometry (voxelslam.cpp:1236, 1577-1583) and on every merged submap of the hierarchical BA (:2447).
// Upstream: an unordered_map from voxel index to a running mean, updated point by point in cloud order, in float.  Here: one key
// per point (upstream's float-typed voxel index), one stable radix sort, run-length encode + scan for the cell table, and one lane
// per occupied voxel that replays the running-mean recurrence over its points in cloud order with unfused float arithmetic -- the
// means are bit-identical to upstream's.  Output order is ascending (x, y, z) voxel index (upstream: hash-map iteration order,
// which is implementation-defined).
#include "vxba_wait.hpp"
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/vxba.h"
#include "vxba_downsample.h"

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

namespace vxd {

int downsample_device(Scratch& sc, hipStream_t s, const float* d_in, int64_t n, double voxel_size, float* d_out, int64_t* n_out) {
  *n_out = 0;
  if (n == 0) return VXBA_OK;
  if (n < 0 || n >= (1ll << 32)) return VXBA_ERR_ARG;
  if (voxel_size < 0.001) {
    if (hipMemcpyAsync(d_out, d_in, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return VXBA_ERR_HIP;
    *n_out = n;
    return VXBA_OK;
  }
  unsigned long long *d_key = nullptr, *d_key_s = nullptr, *d_ukey = nullptr;
  unsigned int *d_idx = nullptr, *d_idx_s = nullptr, *d_cnt = nullptr, *d_runs = nullptr;
  long long* d_ptr = nullptr;
  int* d_err = nullptr;
  void* d_temp = nullptr;
  size_t t_sort = 0, t_rle = 0, t_scan = 0;
  unsigned int runs = 0;
  int err = 0;
  const unsigned grid = (unsigned)((n + 255) / 256), grid1 = (unsigned)((n + 256) / 256);
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
#define DS(call) do { if ((call) != hipSuccess) return VXBA_ERR_HIP; } while (0)
  DS(rocprim::radix_sort_pairs(nullptr, t_sort, d_key, d_key_s, d_idx, d_idx_s, (size_t)n, 0, 63, s));
  DS(rocprim::run_length_encode(nullptr, t_rle, d_key_s, (size_t)n, d_ukey, d_cnt, d_runs, s));
  DS(rocprim::exclusive_scan(nullptr, t_scan, d_ptr, d_ptr, 0ll, (size_t)n + 1, rocprim::plus<long long>(), s));
  size_t t = t_sort > t_rle ? t_sort : t_rle;
  if (t_scan > t) t = t_scan;
  if (!t) t = 8;
  {
    const size_t b_k = up((size_t)n * 8), b_i = up((size_t)n * 4), b_p = up((size_t)(n + 1) * 8), b_t = up(t);
    const size_t need = 3 * b_k + 3 * b_i + 256 + b_p + 256 + b_t;
    if (need > sc.cap) {
      if (sc.base) { DS(hipStreamSynchronize(s)); hipFree(sc.base); }
      sc.base = nullptr; sc.cap = 0;
      DS(hipMalloc((void**)&sc.base, need + need / 4));
      sc.cap = need + need / 4;
    }
    char* q = sc.base;
    auto carve = [&](size_t bytes) { char* r = q; q += bytes; return r; };
    d_key = (unsigned long long*)carve(b_k); d_key_s = (unsigned long long*)carve(b_k); d_ukey = (unsigned long long*)carve(b_k);
    d_idx = (unsigned int*)carve(b_i); d_idx_s = (unsigned int*)carve(b_i); d_cnt = (unsigned int*)carve(b_i);
    d_runs = (unsigned int*)carve(256); d_ptr = (long long*)carve(b_p); d_err = (int*)carve(256); d_temp = carve(b_t);
  }
  // The two words the host needs -- the range-error flag and the number of occupied voxels -- are written by the kernels that produce them
  // straight into mapped pinned memory (round 6: no memset, no copy-engine round trips; hipHostMalloc'd memory is mapped at the same address).
  if (!sc.pinned) DS(hipHostMalloc((void**)&sc.pinned, 64, hipHostMallocDefault));
  ((volatile unsigned int*)sc.pinned)[0] = 0;
  ((volatile unsigned int*)sc.pinned)[1] = 0;
  d_err = (int*)sc.pinned;
  d_runs = (unsigned int*)sc.pinned + 1;
  ds_key_kernel<<<grid, 256, 0, s>>>(d_in, n, voxel_size, d_key, d_idx, d_err);
  DS(hipGetLastError());
  size_t tt = t;
  DS(rocprim::radix_sort_pairs(d_temp, tt, d_key, d_key_s, d_idx, d_idx_s, (size_t)n, 0, 63, s));
  tt = t;
  DS(rocprim::run_length_encode(d_temp, tt, d_key_s, (size_t)n, d_ukey, d_cnt, d_runs, s));
  {   // by polling: a blocking wait parks the thread (~25 us to wake up from); pinned destination: a copy into pageable memory is staged and waited for
    hipError_t q;
    q = vxwait::stream_wait(s);
    DS(q);
  }
  err = (int)((volatile unsigned int*)sc.pinned)[0];
  runs = ((volatile unsigned int*)sc.pinned)[1];
  if (err) return VXBA_ERR_ARG;   // a voxel index outside [-2^20, 2^20)
  ds_widen_kernel<<<grid1, 256, 0, s>>>(d_cnt, (long long)runs, d_ptr);
  DS(hipGetLastError());
  tt = t;
  DS(rocprim::exclusive_scan(d_temp, tt, d_ptr, d_ptr, 0ll, (size_t)runs + 1, rocprim::plus<long long>(), s));
  ds_mean_kernel<<<(unsigned)((runs + 63) / 64), 64, 0, s>>>(d_in, d_idx_s, d_ptr, (long long)runs, d_out);
  DS(hipGetLastError());
#undef DS
  *n_out = (int64_t)runs;
  return VXBA_OK;
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
  // one grow-only scratch + in / out buffers per device for this stand-alone entry point (it runs once per scan: a dozen hipMalloc / hipFree
  // pairs per call cost more than the sort); calls are serialised on it
  static std::mutex mtx;
  static vxd::Scratch scratch[16];
  static float* io[16] = {nullptr};
  static size_t io_cap[16] = {0};
  std::lock_guard<std::mutex> lock(mtx);
  const int dv = device < 16 ? device : 15;
  if ((size_t)n > io_cap[dv] || device >= 16) {
    if (io[dv]) { hipDeviceSynchronize(); hipFree(io[dv]); }
    io[dv] = nullptr; io_cap[dv] = 0;
    const size_t want = (size_t)n + (size_t)n / 4;
    if (hipMalloc((void**)&io[dv], 2 * want * 3 * sizeof(float)) != hipSuccess) return VXBA_ERR_HIP;
    io_cap[dv] = want;
  }
  float* d_in = io[dv];
  float* d_out = io[dv] + 3 * io_cap[dv];
  if (hipMemcpy(d_in, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return VXBA_ERR_HIP;
  int64_t kept = 0;
  int rc;
  if (device >= 16) {
    // devices beyond the per-device table share slot 15 for the in / out buffers (re-allocated on every call, above); the sort's scratch must
    // not be shared: it would keep memory owned by the previous such device (round-5 advisor) -- a scratch of the call's own instead
    vxd::Scratch own;
    rc = vxd::downsample_device(own, nullptr, d_in, n, voxel_size, d_out, &kept);
    (void)hipDeviceSynchronize();
    own.release();
  } else {
    rc = vxd::downsample_device(scratch[dv], nullptr, d_in, n, voxel_size, d_out, &kept);
  }
  if (rc != VXBA_OK) return rc;
  if (hipMemcpy(out_xyz, d_out, (size_t)kept * 3 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return VXBA_ERR_HIP;
  *n_out = kept;
  return VXBA_OK;
}
