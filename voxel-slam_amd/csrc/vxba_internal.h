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
// The scan resident in an odometry handle: body points + covariances as structure of arrays (rows 0..2 = body point, stride
// doubles apart), and what the last vxba_lio_pvec_update left on the device (world points n x 3, then world covariances n x 9).
int vxba_internal_lio_scan_view(vxba_lio* h, const double** d_pts_soa, long long* n, long long* stride, const double** d_world, int* world_valid);
// vxba_lio_map_update with device arrays.
int vxba_internal_lio_map_update_device(vxba_lio* h, long long n, const long long* d_loc, const int* d_layer, const int* d_path, const int* d_is_plane, const double* d_center,
                                        const double* d_normal, const double* d_plane_var, const double* d_radius);
int vxba_internal_lio_geometry(const vxba_lio* h, double* voxel_size, int* max_layer, int* device);
// Device ordinal a factor lives on (-1 for a null handle): handles that exchange raw device pointers must share it.
int vxba_internal_factor_device(const vxba_factor* f);
}
