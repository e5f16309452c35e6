// Library-internal links between translation units of libvxba.so (not part of the C ABI, not exported to callers).
#pragma once
#include "../../include/vxba.h"

extern "C" {
// vxba_push_voxels with every array already in device memory (AoS, the formats of vxba_push_voxels).  Synchronous.
int vxba_internal_push_voxels_device(vxba_factor* f, int n, const double* d_clusters, const double* d_fix, const double* d_coe, const double* d_eigval,
                                     const double* d_eigvec, const double* d_merged);
// Device view of the (lambda, U, pcr_add) cache: plane k of voxel a at ptr[k * VS + a] (eig_vec planes column-major: k = 3 col + row).
// Waits for the factor's stream first.
int vxba_internal_cache_view(vxba_factor* f, const double** eigval, const double** eigvec, const double** merged, int* VS, int* V);
}
