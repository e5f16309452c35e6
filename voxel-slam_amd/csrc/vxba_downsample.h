// Device-to-device core of the voxel-grid filter (vxba_downsample.hip), for callers inside the library (vxba_hba.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vxd {

// grow-only temporaries of one caller (one stream): never shared between threads
struct Scratch {
  char* base = nullptr;
  size_t cap = 0;
  void release() { if (base) hipFree(base); base = nullptr; cap = 0; }
};
// d_in: n x 3 floats on the device (unchanged), d_out: room for n x 3 floats; *n_out = occupied voxels.  Two small device-to-host copies
// (range check, voxel count) synchronise with `s`.  voxel_size < 0.001: the cloud is copied through (tools.hpp:203).
int downsample_device(Scratch& sc, hipStream_t s, const float* d_in, int64_t n, double voxel_size, float* d_out, int64_t* n_out);

}  // namespace vxd
