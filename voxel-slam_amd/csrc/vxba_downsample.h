// Device-to-device core of the voxel-grid filter (vxba_downsample.hip), for callers inside the library (vxba_hba.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vxd {

// grow-only temporaries of one caller (one stream): never shared between threads
struct Scratch {
  char* base = nullptr;
  size_t cap = 0;
  unsigned int* pinned = nullptr;   // two words of pinned host memory: range check and voxel count come down here (polled, not waited for)
  void release() { if (base) hipFree(base); if (pinned) hipHostFree(pinned); base = nullptr; pinned = nullptr; cap = 0; }
};
// d_in: n x 3 floats on the device (unchanged), d_out: room for n x 3 floats; *n_out = occupied voxels.  Two small device-to-host copies
// (range check, voxel count) are waited for by polling `s`.  voxel_size < 0.001: the cloud is copied through (tools.hpp:203).
int downsample_device(Scratch& sc, hipStream_t s, const float* d_in, int64_t n, double voxel_size, float* d_out, int64_t* n_out);

}  // namespace vxd
