"""ctypes binding of ``libvxba.so`` (include/vxba.h) with the reference's own names.

``LidarFactor`` mirrors ``class LidarFactor`` (VoxelSLAM/src/voxel_map.hpp:109-290) and
``Lidar_BA_Optimizer`` mirrors voxel_map.hpp:293-444, so parity tests read like calls into
the reference.  Everything numeric happens in the HIP kernels behind the C ABI; this module
holds no arithmetic and has NO CPU fallback -- a missing library or GPU raises ``VxbaError``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VXBA_LIB") or os.path.join(_HERE, "csrc", "libvxba.so")   # VXBA_LIB: A/B runs of two builds
MAX_WIN = 10
MAX_WIN_WIDE = 128
TRACE_COLS = 8

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_HESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))
_RESID_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))

# every symbol include/vxba.h declares (the CPU test suite checks the library exports all of them)
EXPORTS = [
    "vxba_create", "vxba_destroy", "vxba_clear", "vxba_set_win_size", "vxba_win_size", "vxba_size", "vxba_set_stream",
    "vxba_reserve", "vxba_last_error", "vxba_push_voxels", "vxba_push_points", "vxba_read_clusters", "vxba_acc_evaluate2",
    "vxba_evaluate_only_residual", "vxba_get_collective_time", "vxba_get_fused_time", "vxba_debug_partials", "vxba_acc_evaluate2_device", "vxba_evaluate_only_residual_device", "vxba_packed_len",
    "vxba_read_cache", "vxba_snapshot_cache", "vxba_restore_cache", "vxba_plane_fit", "vxba_plane_fit_judge", "vxba_build_clusters", "vxba_set_allreduce",
    "vxba_rccl_unique_id", "vxba_rccl_attach", "vxba_rccl_attach_bcast", "vxba_rccl_detach", "vxba_peer_export", "vxba_peer_attach", "vxba_peer_detach", "vxba_peer_status", "vxba_peer_selftest", "vxba_use_external_buffers", "vxba_damping_iter", "vxba_damping_iter_generic", "vxba_lm_steps", "vxba_set_profiling", "vxba_get_kernel_times", "vxba_algorithmic_bytes", "vxba_nnz", "vxba_device_bytes", "vxba_debug_mfma_probe", "vxba_debug_stamps", "vxba_debug_band_schur", "vxba_push_voxels_csr",
    "vxba_imu_init", "vxba_imu_add", "vxba_imu_evaluate", "vxba_imu_update_state", "vxba_hess_plus", "vxba_hess_plus_gravity", "vxba_li_evaluate", "vxba_li_evaluate_gravity",
    "vxba_li_only_residual", "vxba_li_damping_iter", "vxba_imu_evaluate_g", "vxba_li_damping_iter_gravity", "vxba_voxelize_push", "vxba_set_precision",
    "vxba_lio_create", "vxba_lio_destroy", "vxba_lio_last_error", "vxba_lio_map_update", "vxba_lio_map_clear", "vxba_lio_map_size", "vxba_lio_scan_raw",
    "vxba_lio_scan_set", "vxba_lio_scan_size", "vxba_lio_scan_read", "vxba_lio_sweep", "vxba_lio_state_estimation", "vxba_lio_pvec_update", "vxba_lio_leaf_stats", "vxba_cov_add_build", "vxba_plane_update", "vxba_down_sampling_voxel", "vxba_voxelize_push_device",
    "vxba_set_option", "vxba_get_option", "vxba_lio_set_option",
    "vxba_hba_create", "vxba_hba_destroy", "vxba_hba_last_error", "vxba_hba_add_keyframes", "vxba_hba_num_keyframes", "vxba_hba_threads_used", "vxba_hba_clear", "vxba_hba_pass", "vxba_hba_num_windows", "vxba_hba_window", "vxba_hba_bottom", "vxba_hba_export_submaps", "vxba_hba_import_submaps", "vxba_hba_top_factor", "vxba_hba_top", "vxba_voxelize_profile",
    "vxba_map_create", "vxba_map_destroy", "vxba_map_last_error", "vxba_map_cut_voxel", "vxba_map_cut_voxel_device", "vxba_map_recut", "vxba_map_margi",
    "vxba_map_slide", "vxba_map_counts", "vxba_map_fix_pool", "vxba_map_set_journey", "vxba_map_release", "vxba_map_device_bytes", "vxba_map_leaves", "vxba_map_cut_voxel_lio", "vxba_map_export_planes",
]

_ERRNAMES = {1: "VXBA_ERR_ARG", 2: "VXBA_ERR_HIP", 3: "VXBA_ERR_NODEV", 4: "VXBA_ERR_STATE", 5: "VXBA_ERR_UNSUPPORTED"}


class VxbaError(RuntimeError):
    pass


class VoxelizeParams(C.Structure):
    """vxba_voxelize_params: the knobs of OctreeGBA::recut (loop_refine.hpp:311-315, 358-378) and of the voxel grid."""
    _fields_ = [("voxel_size", C.c_double), ("max_layer", C.c_int), ("min_points", C.c_int), ("min_eigen_value", C.c_double),
                ("eigen_ratio", C.c_double * 4), ("factor_ratio_max", C.c_double), ("min_points_layer", C.c_int * 4), ("min_frames", C.c_int),
                ("shard_index", C.c_int), ("shard_count", C.c_int)]

    def __init__(self, voxel_size=1.0, max_layer=2, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 16, 1 / 16),
                 factor_ratio_max=0.12, min_points_layer=(0, 0, 0, 0), min_frames=2, shard_index=0, shard_count=0):
        """Defaults: OctreeGBA.  OctoTree's batch build (motion_init): min_points_layer=min_point[layer], min_frames=0.
        shard_count > 1: keep the root voxels that hash to shard_index (voxel-sharded windows, one rank per GPU)."""
        super().__init__(voxel_size, max_layer, min_points, min_eigen_value, (C.c_double * 4)(*eigen_ratio), factor_ratio_max,
                         (C.c_int * 4)(*min_points_layer), min_frames, shard_index, shard_count)

    def sharded(self, shard_index: int, shard_count: int) -> "VoxelizeParams":
        """The same parameters for one shard of a voxel-sharded window."""
        return VoxelizeParams(self.voxel_size, self.max_layer, self.min_points, self.min_eigen_value, tuple(self.eigen_ratio), self.factor_ratio_max,
                              tuple(self.min_points_layer), self.min_frames, shard_index, shard_count)

    def as_array(self):
        return np.array([self.voxel_size, self.max_layer, self.min_points, self.min_eigen_value, *self.eigen_ratio, self.factor_ratio_max,
                         *self.min_points_layer, self.min_frames], dtype=np.float64)


_lib = None


class MapParams(C.Structure):
    """vxba_map_params (include/vxba.h)."""
    _fields_ = [("voxel_size", C.c_double), ("max_layer", C.c_int), ("min_point", C.c_double * 4), ("min_eigen_value", C.c_double),
                ("plane_eigen_value_thre", C.c_double * 4), ("max_points", C.c_int), ("win_size", C.c_int), ("thread_num", C.c_int)]


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """Load libvxba.so; fails loudly if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise VxbaError(f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    L = C.CDLL(path)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    L.vxba_create.argtypes = [ci, ci, C.POINTER(vp)]
    L.vxba_destroy.argtypes = [vp]
    L.vxba_clear.argtypes = [vp]
    L.vxba_set_win_size.argtypes = [vp, ci]
    L.vxba_win_size.argtypes = [vp]
    L.vxba_size.argtypes = [vp]
    L.vxba_set_stream.argtypes = [vp, vp]
    L.vxba_reserve.argtypes = [vp, ci]
    L.vxba_last_error.argtypes = [vp]
    L.vxba_last_error.restype = C.c_char_p
    L.vxba_push_voxels.argtypes = [vp, ci, _f64p, _f64p, _f64p, vp, vp, vp]
    L.vxba_push_voxels_csr.argtypes = [vp, ci, vp, vp, vp, _f64p, _f64p, vp, vp, vp]
    L.vxba_push_points.argtypes = [vp, ci, C.c_int64, _f64p, _i64p, vp, vp]
    L.vxba_read_clusters.argtypes = [vp, ci, ci, _f64p]
    L.vxba_acc_evaluate2.argtypes = [vp, _f64p, ci, ci, _f64p, _f64p, C.POINTER(cd)]
    L.vxba_evaluate_only_residual.argtypes = [vp, _f64p, ci, ci, C.POINTER(cd)]
    L.vxba_acc_evaluate2_device.argtypes = [vp, _f64p, ci, ci, vp]
    L.vxba_evaluate_only_residual_device.argtypes = [vp, _f64p, ci, ci, vp]
    L.vxba_packed_len.argtypes = [vp]
    L.vxba_packed_len.restype = C.c_size_t
    L.vxba_read_cache.argtypes = [vp, ci, ci, _f64p, _f64p, _f64p]
    L.vxba_snapshot_cache.argtypes = [vp]
    L.vxba_restore_cache.argtypes = [vp]
    L.vxba_plane_fit.argtypes = [ci, C.c_int64, _f64p, _f64p, _f64p]
    L.vxba_plane_fit_judge.argtypes = [ci, C.c_int64, _f64p, ci, cd, cd, cd, _f64p, _f64p, np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")]
    L.vxba_build_clusters.argtypes = [ci, C.c_int64, C.c_int64, _f64p, _i64p, _f64p]
    L.vxba_set_allreduce.argtypes = [vp, _ALLREDUCE_FN, vp]
    L.vxba_use_external_buffers.argtypes = [vp, vp, vp]
    L.vxba_rccl_unique_id.argtypes = [C.c_char_p, vp]
    L.vxba_rccl_attach.argtypes = [vp, C.c_char_p, ci, ci, vp]
    L.vxba_rccl_detach.argtypes = [vp]
    L.vxba_damping_iter.argtypes = [vp, _f64p, ci, _f64p, _f64p, _f64p, C.POINTER(ci), C.POINTER(ci)]
    L.vxba_damping_iter_generic.argtypes = [ci, _f64p, ci, _HESS_FN, _RESID_FN, vp, _f64p, _f64p, _f64p, C.POINTER(ci), C.POINTER(ci)]
    L.vxba_lm_steps.argtypes = [vp, _f64p, ci, ci, _f64p, _f64p, _i64p]
    L.vxba_set_profiling.argtypes = [vp, ci]
    L.vxba_set_precision.argtypes = [vp, ci]
    L.vxba_get_kernel_times.argtypes = [vp, _f64p, _i64p, ci]
    L.vxba_get_collective_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), ci]
    L.vxba_get_fused_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), ci]
    L.vxba_algorithmic_bytes.argtypes = [vp, _f64p]
    L.vxba_nnz.argtypes = [vp, C.POINTER(C.c_int64)]
    L.vxba_device_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
    L.vxba_debug_mfma_probe.argtypes = [ci, _f64p, _f64p, _f64p]
    L.vxba_debug_stamps.argtypes = [ci, vp, C.c_size_t]
    L.vxba_imu_init.argtypes = [_f64p, vp, vp]
    L.vxba_imu_add.argtypes = [_f64p, _f64p, _f64p, cd, _f64p, _f64p]
    L.vxba_imu_evaluate.argtypes = [_f64p, _f64p, _f64p, ci, vp, vp, C.POINTER(cd)]
    L.vxba_imu_update_state.argtypes = [_f64p, _f64p]
    L.vxba_hess_plus.argtypes = [ci, _f64p, _f64p, _f64p, _f64p]
    L.vxba_li_evaluate.argtypes = [vp, _f64p, _f64p, cd, _f64p, _f64p, C.POINTER(cd)]
    L.vxba_hess_plus_gravity.argtypes = [ci, _f64p, _f64p, _f64p, _f64p]
    L.vxba_li_evaluate_gravity.argtypes = [vp, _f64p, _f64p, cd, _f64p, _f64p, C.POINTER(cd)]
    L.vxba_li_only_residual.argtypes = [vp, _f64p, _f64p, cd, C.POINTER(cd)]
    L.vxba_li_damping_iter.argtypes = [vp, _f64p, _f64p, cd, ci, vp, vp, C.POINTER(ci)]
    L.vxba_voxelize_push.argtypes = [vp, C.c_int64, _f64p, _i64p, _f64p, C.POINTER(VoxelizeParams), C.POINTER(C.c_int64), vp, C.c_int64]
    L.vxba_voxelize_push_device.argtypes = [vp, C.c_int64, vp, _i64p, _f64p, C.POINTER(VoxelizeParams), C.POINTER(C.c_int64), vp, C.c_int64]
    L.vxba_imu_evaluate_g.argtypes = [_f64p, _f64p, _f64p, ci, vp, vp, C.POINTER(cd)]
    L.vxba_li_damping_iter_gravity.argtypes = [vp, _f64p, _f64p, cd, ci, vp, _f64p, vp, C.POINTER(ci)]
    i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    L.vxba_lio_create.argtypes = [cd, ci, ci, C.POINTER(vp)]
    L.vxba_lio_destroy.argtypes = [vp]
    L.vxba_lio_last_error.argtypes = [vp]
    L.vxba_lio_last_error.restype = C.c_char_p
    L.vxba_lio_map_update.argtypes = [vp, C.c_int64, _i64p, i32p, i32p, vp, _f64p, _f64p, _f64p, _f64p]
    L.vxba_lio_map_clear.argtypes = [vp]
    L.vxba_lio_map_size.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.vxba_lio_scan_raw.argtypes = [vp, C.c_int64, np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS"), vp, cd, cd]
    L.vxba_lio_scan_set.argtypes = [vp, C.c_int64, _f64p, _f64p]
    L.vxba_lio_scan_size.argtypes = [vp]
    L.vxba_lio_scan_size.restype = C.c_int64
    L.vxba_lio_scan_read.argtypes = [vp, _f64p, _f64p]
    L.vxba_lio_sweep.argtypes = [vp, _f64p, _f64p, ci, _f64p, vp, vp]
    L.vxba_lio_state_estimation.argtypes = [vp, _f64p, _f64p, vp, vp]
    L.vxba_lio_pvec_update.argtypes = [vp, _f64p, _f64p, vp, vp]
    L.vxba_lio_leaf_stats.argtypes = [vp, C.c_int64, _i64p, vp, _f64p, _f64p]
    L.vxba_cov_add_build.argtypes = [ci, C.c_int64, C.c_int64, _f64p, _f64p, _i64p, _f64p]
    L.vxba_down_sampling_voxel.argtypes = [ci, C.c_int64, np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS"), cd,
                                           np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS"), C.POINTER(C.c_int64)]
    L.vxba_plane_update.argtypes = [ci, C.c_int64, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p]
    L.vxba_peer_export.argtypes = [vp, vp]
    L.vxba_peer_attach.argtypes = [vp, ci, ci, vp]
    L.vxba_peer_detach.argtypes = [vp]
    L.vxba_peer_status.argtypes = [vp, C.POINTER(ci)]
    L.vxba_peer_selftest.argtypes = [vp, C.POINTER(ci)]
    L.vxba_set_option.argtypes = [vp, ci, ci]
    L.vxba_get_option.argtypes = [vp, ci, C.POINTER(ci)]
    L.vxba_lio_set_option.argtypes = [vp, ci, ci]
    L.vxba_map_create.argtypes = [C.POINTER(MapParams), ci, C.POINTER(vp)]
    L.vxba_map_destroy.argtypes = [vp]
    L.vxba_map_last_error.argtypes = [vp]
    L.vxba_map_last_error.restype = C.c_char_p
    L.vxba_map_cut_voxel.argtypes = [vp, ci, C.c_int64, _f64p, _f64p, _f64p]
    L.vxba_map_cut_voxel_device.argtypes = [vp, ci, C.c_int64, vp, vp, vp]
    L.vxba_map_cut_voxel_lio.argtypes = [vp, ci, vp]
    L.vxba_map_export_planes.argtypes = [vp, vp, C.POINTER(C.c_int64)]
    L.vxba_map_recut.argtypes = [vp, ci, _f64p, vp, C.POINTER(C.c_int64)]
    L.vxba_map_margi.argtypes = [vp, ci, _f64p, vp]
    L.vxba_map_slide.argtypes = [vp, ci]
    L.vxba_map_counts.argtypes = [vp, _i64p]
    L.vxba_map_fix_pool.argtypes = [vp, _i64p]
    L.vxba_map_set_journey.argtypes = [vp, C.c_double]
    L.vxba_map_release.argtypes = [vp, C.c_double, ci, vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.vxba_map_device_bytes.argtypes = [vp, _i64p]
    L.vxba_map_leaves.argtypes = [vp, C.c_int64, vp, vp, vp, C.POINTER(C.c_int64)]
    _lib = L
    return L


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _opt(a):
    """Optional f64 array -> pointer (or NULL); returns (pointer, keepalive)."""
    if a is None:
        return None, None
    arr = _c(a)
    return arr.ctypes.data_as(C.c_void_p), arr


class LidarFactor:
    """The LiDAR BA factor on one MI355X (reference: voxel_map.hpp:109-290)."""

    def __init__(self, win_size: int, device: int = 0):
        self._L = load_library()
        self._h = C.c_void_p()
        rc = self._L.vxba_create(int(win_size), int(device), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise VxbaError(f"vxba_create(win_size={win_size}, device={device}) failed: {_ERRNAMES.get(rc, rc)} "
                            "(needs a gfx950 GPU; there is no CPU fallback)")
        self._cb = None
        self._owned = True

    @classmethod
    def from_handle(cls, handle):
        """A view of a factor somebody else owns (the top-level factor of an ``HbaSession``): every method works, ``close`` leaves it alone."""
        self = cls.__new__(cls)
        self._L = load_library()
        self._h = C.c_void_p(handle if isinstance(handle, int) else handle.value)
        self._cb = None
        self._owned = False
        return self

    # -- plumbing -------------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            msg = self._L.vxba_last_error(self._h)
            raise VxbaError(f"{_ERRNAMES.get(rc, rc)}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            if getattr(self, "_owned", True):
                self._L.vxba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def win_size(self) -> int:
        return self._L.vxba_win_size(self._h)

    @win_size.setter
    def win_size(self, w: int):
        self._chk(self._L.vxba_set_win_size(self._h, int(w)))

    def size(self) -> int:
        """``plvec_voxels.size()``"""
        return self._L.vxba_size(self._h)

    def __len__(self):
        return self.size()

    def set_stream(self, hip_stream: int | None):
        self._chk(self._L.vxba_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def reserve(self, n):
        self._chk(self._L.vxba_reserve(self._h, int(n)))

    # -- construction ---------------------------------------------------------------------------
    def push_voxel(self, vec_orig, fix, coe, eig_value, eig_vector, pcr_add):
        """Single-voxel form of voxel_map.hpp:122-130 (all arguments in packed format)."""
        self.push_voxels(np.asarray(vec_orig)[None], np.asarray(fix)[None], [coe], np.asarray(eig_value)[None],
                         np.asarray(eig_vector).reshape(1, 9), np.asarray(pcr_add)[None])

    def push_voxels(self, clusters, fix, coe, eig_val=None, eig_vec=None, merged=None):
        clusters = _c(clusters)
        n = clusters.shape[0]
        if clusters.shape != (n, self.win_size, 10):
            raise VxbaError(f"clusters must be (n, {self.win_size}, 10)")
        pv, kv = _opt(eig_val)
        pu, ku = _opt(eig_vec)
        pm, km = _opt(merged)
        self._chk(self._L.vxba_push_voxels(self._h, n, clusters, _c(fix), _c(coe), pv, pu, pm))

    def push_voxels_csr(self, row_ptr, frame_idx, clusters, fix, coe, eig_val=None, eig_vec=None, merged=None):
        """push_voxels for sparse incidence: voxel a's observed frames are frame_idx[row_ptr[a]:row_ptr[a + 1]] (strictly increasing), their
        clusters the matching rows of ``clusters`` (nnz, 10).  Nothing dense is built on the host or sent over PCIe."""
        rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
        fr = np.ascontiguousarray(frame_idx, dtype=np.int32)
        cl = _c(clusters).reshape(-1, 10)
        n = rp.size - 1
        if n < 0 or cl.shape[0] != fr.size or (n >= 0 and rp.size and rp[-1] != fr.size):
            raise VxbaError("push_voxels_csr: row_ptr[-1], frame_idx and clusters disagree on the number of entries")
        pv, kv = _opt(eig_val)
        pu, ku = _opt(eig_vec)
        pm, km = _opt(merged)
        self._chk(self._L.vxba_push_voxels_csr(self._h, n, rp.ctypes.data_as(C.c_void_p), fr.ctypes.data_as(C.c_void_p), cl.ctypes.data_as(C.c_void_p), _c(fix), _c(coe), pv, pu, pm))

    def push_points(self, n_voxels, xyz_body, cell_ptr, fix=None, coe=None):
        """K1: build the clusters of ``n_voxels`` voxels on the GPU from bucketed body-frame points."""
        xyz = _c(xyz_body).reshape(-1, 3)
        ptr = np.ascontiguousarray(cell_ptr, dtype=np.int64)
        pf, kf = _opt(fix)
        pc, kc = _opt(coe)
        self._chk(self._L.vxba_push_points(self._h, int(n_voxels), xyz.shape[0], xyz, ptr, pf, pc))

    def voxelize_push(self, xyz_local, frame_ptr, xs, params: "VoxelizeParams", want_ids=True):
        """OctreeGBA::cut_voxel + recut on the GPU (loop_refine.hpp:273-476): appends the factor voxels found in the window's
        points; returns their canonical node ids (uint64) in push order (or just the count)."""
        xyz = _c(xyz_local).reshape(-1, 3)
        fp = np.ascontiguousarray(frame_ptr, dtype=np.int64)
        n = C.c_int64(0)
        cap = self._ids_cap(xyz.shape[0], params) if want_ids else 0
        ids = np.zeros(max(cap, 1), dtype=np.uint64)
        self._chk(self._L.vxba_voxelize_push(self._h, xyz.shape[0], xyz, fp, _c(xs), C.byref(params), C.byref(n),
                                             ids.ctypes.data_as(C.c_void_p) if want_ids else None, cap))
        return ids[: n.value].copy() if want_ids else n.value

    @staticmethod
    def _ids_cap(n_points, params):
        floor_pts = min([params.min_points] + [v for v in params.min_points_layer[: params.max_layer + 1] if v > 0])
        return n_points // (max(floor_pts, 0) + 1) + 1

    def voxelize_push_device(self, d_xyz_ptr: int, n_points: int, frame_ptr, xs, params: "VoxelizeParams", want_ids=True):
        """:meth:`voxelize_push` on points that already live in device memory (``d_xyz_ptr``: n_points x 3 float64, e.g. a torch
        tensor's ``data_ptr()``)."""
        fp = np.ascontiguousarray(frame_ptr, dtype=np.int64)
        n = C.c_int64(0)
        cap = self._ids_cap(n_points, params) if want_ids else 0
        ids = np.zeros(max(cap, 1), dtype=np.uint64)
        self._chk(self._L.vxba_voxelize_push_device(self._h, int(n_points), C.c_void_p(d_xyz_ptr), fp, _c(xs), C.byref(params), C.byref(n),
                                                    ids.ctypes.data_as(C.c_void_p) if want_ids else None, cap))
        return ids[: n.value].copy() if want_ids else n.value

    def read_clusters(self, head=0, end=None):
        end = self.size() if end is None else end
        out = np.zeros((end - head, self.win_size, 10))
        self._chk(self._L.vxba_read_clusters(self._h, head, end, out))
        return out

    def clear(self):
        self._chk(self._L.vxba_clear(self._h))

    # -- sweeps ---------------------------------------------------------------------------------
    def acc_evaluate2(self, xs, head=0, end=None):
        """Returns (Hess[r, c], JacT, residual) for voxels [head, end) under packed poses ``xs`` (W, 12)."""
        end = self.size() if end is None else end
        n = 6 * self.win_size
        H = np.zeros((n, n)); J = np.zeros(n); r = C.c_double(0)
        self._chk(self._L.vxba_acc_evaluate2(self._h, _c(xs), head, end, H, J, C.byref(r)))
        return H.T.copy(), J, r.value   # column-major on the wire

    def evaluate_only_residual(self, xs, head=0, end=None):
        end = self.size() if end is None else end
        r = C.c_double(0)
        self._chk(self._L.vxba_evaluate_only_residual(self._h, _c(xs), head, end, C.byref(r)))
        return r.value

    def packed_len(self) -> int:
        return self._L.vxba_packed_len(self._h)

    def acc_evaluate2_device(self, xs, d_out_ptr: int, head=0, end=None):
        end = self.size() if end is None else end
        self._chk(self._L.vxba_acc_evaluate2_device(self._h, _c(xs), head, end, C.c_void_p(d_out_ptr)))

    def evaluate_only_residual_device(self, xs, d_out_ptr: int, head=0, end=None):
        end = self.size() if end is None else end
        self._chk(self._L.vxba_evaluate_only_residual_device(self._h, _c(xs), head, end, C.c_void_p(d_out_ptr)))

    def read_cache(self, head=0, end=None):
        """(eig_values (n,3), eig_vectors (n,9 col-major), pcr_adds (n,10)) of voxels [head, end)."""
        end = self.size() if end is None else end
        n = end - head
        ev = np.zeros((n, 3)); U = np.zeros((n, 9)); m = np.zeros((n, 10))
        self._chk(self._L.vxba_read_cache(self._h, head, end, ev, U, m))
        return ev, U, m

    def snapshot_cache(self):
        self._chk(self._L.vxba_snapshot_cache(self._h))

    def restore_cache(self):
        self._chk(self._L.vxba_restore_cache(self._h))

    # -- multi-GPU hook -------------------------------------------------------------------------
    def set_allreduce(self, fn):
        """``fn(device_ptr: int, count: int, hip_stream: int) -> None`` sums f64 across voxel shards in place."""
        if fn is None:
            self._cb = None
            self._chk(self._L.vxba_set_allreduce(self._h, C.cast(None, _ALLREDUCE_FN), None))
            return

        def tramp(ctx, ptr, count, stream):
            try:
                fn(int(ptr or 0), int(count), int(stream or 0))
                return 0
            except Exception:  # noqa: BLE001 - must not unwind through C
                import traceback
                traceback.print_exc()
                return 1

        self._cb = _ALLREDUCE_FN(tramp)
        self._chk(self._L.vxba_set_allreduce(self._h, self._cb, None))

    def rccl_attach(self, librccl_path: str, nranks: int, rank: int, unique_id: bytes):
        """Direct RCCL all-reduce of the exchange buffers inside the device-resident LM loop (collective call)."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._L.vxba_rccl_attach(self._h, librccl_path.encode(), int(nranks), int(rank), C.cast(buf, C.c_void_p)))

    def rccl_detach(self):
        self._chk(self._L.vxba_rccl_detach(self._h))

    def use_external_buffers(self, d_packed_ptr: int | None, d_scalar_ptr: int | None):
        self._chk(self._L.vxba_use_external_buffers(self._h, C.c_void_p(d_packed_ptr or 0), C.c_void_p(d_scalar_ptr or 0)))

    # -- measurement ----------------------------------------------------------------------------
    def peer_export(self) -> bytes:
        """64-byte IPC handle of this factor's all-reduce mailbox (vxba_peer_export)."""
        buf = C.create_string_buffer(64)
        self._chk(self._L.vxba_peer_export(self._h, buf))
        return buf.raw

    def peer_attach(self, nranks: int, rank: int, handles):
        """handles: the ranks' peer_export() results in rank order."""
        blob = b"".join(handles)
        assert len(blob) == 64 * nranks
        self._chk(self._L.vxba_peer_attach(self._h, int(nranks), int(rank), C.c_char_p(blob)))

    def peer_detach(self):
        self._chk(self._L.vxba_peer_detach(self._h))

    def peer_selftest(self) -> bool:
        ok = C.c_int(0)
        self._chk(self._L.vxba_peer_selftest(self._h, C.byref(ok)))
        return bool(ok.value)

    def peer_status(self) -> int:
        st = C.c_int(0)
        self._chk(self._L.vxba_peer_status(self._h, C.byref(st)))
        return st.value

    OPTIONS = {"fused_solve": 0, "spec_collective": 1, "wide_device_solve": 2, "k2_voxels_per_block": 4,
               "debug_solve_timeout": 5, "li_structured_solve": 6, "li_queued_sweeps": 7, "li_device_pose_solve": 8, "fused_sweeps": 9, "stat_fused_fallbacks": 100, "stat_li_last_call_us": 101, "stat_li_device_fallbacks": 102, "stat_reject_heavy": 103}

    def set_option(self, name: str, value: int):
        """Execution options of include/vxba.h (VXBA_OPT_*): fused_solve, spec_collective, wide_device_solve, k2_voxels_per_block."""
        self._chk(self._L.vxba_set_option(self._h, self.OPTIONS[name], int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        self._chk(self._L.vxba_get_option(self._h, self.OPTIONS[name], C.byref(v)))
        return v.value

    def set_precision(self, mode: str):
        """'f64' (default), 'mixed' (f32 products on the matrix cores, f64 accumulation; BASELINE configs[2]) or 'mixed_f32_clusters'
        (mixed, and the residual sweep reads f32 re-centred cluster records)."""
        self._chk(self._L.vxba_set_precision(self._h, {"f64": 0, "mixed": 1, "mixed_f32_clusters": 2}[mode]))

    def set_profiling(self, mask: int):
        """Bit mask of kernels to time with events: 1 K3, 2 K2, 4 K3 finalize, 8 K1, 16 all-reduce, 32 the fused launch (0 = off)."""
        self._chk(self._L.vxba_set_profiling(self._h, int(mask)))

    def kernel_times(self, reset=False):
        ms = np.zeros(4); calls = np.zeros(4, dtype=np.int64)
        self._chk(self._L.vxba_get_kernel_times(self._h, ms, calls, int(reset)))
        names = ("k3_hessian", "k2_residual", "k3_finalize", "k1_build")
        return {k: dict(ms_sum=float(ms[i]), calls=int(calls[i])) for i, k in enumerate(names)}

    def collective_time(self, reset=False):
        """All-reduces of a voxel-sharded factor bracketed while profiling bit 16 was set: dict(ms_sum, calls)."""
        ms = C.c_double(0); calls = C.c_int64(0)
        self._chk(self._L.vxba_get_collective_time(self._h, C.byref(ms), C.byref(calls), int(reset)))
        return dict(ms_sum=float(ms.value), calls=int(calls.value))

    def fused_time(self, reset=False):
        """Fused solve + residual + Hessian launches bracketed while profiling bit 32 was set: dict(ms_sum, calls)."""
        ms = C.c_double(0); calls = C.c_int64(0)
        self._chk(self._L.vxba_get_fused_time(self._h, C.byref(ms), C.byref(calls), int(reset)))
        return dict(ms_sum=float(ms.value), calls=int(calls.value))

    def algorithmic_bytes(self):
        b = np.zeros(2)
        self._chk(self._L.vxba_algorithmic_bytes(self._h, b))
        return dict(k3=float(b[0]), k2=float(b[1]))

    def nnz(self) -> int:
        v = C.c_int64(0)
        self._chk(self._L.vxba_nnz(self._h, C.byref(v)))
        return v.value

    def device_bytes(self):
        """Device memory held by the factor: dict(store, work, scratch, total) in bytes (vxba_device_bytes)."""
        b = (C.c_int64 * 4)()
        self._chk(self._L.vxba_device_bytes(self._h, b))
        return dict(store=int(b[0]), work=int(b[1]), scratch=int(b[2]), total=int(b[3]))

    def lm_steps(self, xs_init, n_steps, steps_per_solve=3):
        """Bench driver: returns (poses, [residual1, residual2] of the last step, dict(iters, accepted, rejected))."""
        out = np.zeros((self.win_size, 12)); resis = np.zeros(2); st = np.zeros(3, dtype=np.int64)
        self._chk(self._L.vxba_lm_steps(self._h, _c(xs_init), int(n_steps), int(steps_per_solve), out, resis, st))
        return out, resis, dict(iters=int(st[0]), accepted=int(st[1]), rejected=int(st[2]))


class Lidar_BA_Optimizer:
    """The LM shell that owns the loop (reference: voxel_map.hpp:293-444)."""

    def damping_iter(self, x_stats, voxhess: LidarFactor, max_iter: int = 3):
        """Returns dict(poses, hess, resis, trace, is_converge); ``x_stats`` (W,12) is not modified."""
        W = voxhess.win_size
        n = 6 * W
        Rp = _c(x_stats).copy()
        hess = np.zeros((n, n)); resis = np.zeros(2); trace = np.zeros((max(max_iter, 1), TRACE_COLS))
        nt = C.c_int(0); conv = C.c_int(0)
        voxhess._chk(voxhess._L.vxba_damping_iter(voxhess.handle, Rp, int(max_iter), hess, resis, trace, C.byref(nt), C.byref(conv)))
        return dict(poses=Rp, hess=hess.T.copy(), resis=resis, trace=trace[: nt.value].copy(), is_converge=bool(conv.value))


STATE_LEN, IMU_LEN, LI_DIM = 24, 304, 15


def pack_state(R, p, v=None, bg=None, ba=None, g=None):
    """One window state in the flat C-ABI format [R col-major | p | v | bg | ba | g] (tools.hpp:135-199)."""
    z = np.zeros(3)
    return np.concatenate([np.asarray(R, dtype=np.float64).T.reshape(9), p, z if v is None else v, z if bg is None else bg,
                           z if ba is None else ba, z if g is None else g]).astype(np.float64)


class IMU_PRE:
    """Preintegrated IMU factor between two consecutive frames (reference: preintegration.hpp:11-310), host-side.
    ``blob`` is the flat VXBA_IMU_LEN record; fields are views into it."""

    _FIELDS = dict(R_delta=(0, 9), p_delta=(9, 3), v_delta=(12, 3), bg=(15, 3), ba=(18, 3), R_bg=(21, 9), p_bg=(30, 9), p_ba=(39, 9),
                   v_bg=(48, 9), v_ba=(57, 9), dtime=(66, 1), dbg=(67, 3), dba=(70, 3), dbg_buf=(73, 3), dba_buf=(76, 3), cov=(79, 225))

    def __init__(self, bg=None, ba=None, blob=None):
        self._L = load_library()
        if blob is not None:
            self.blob = _c(blob).copy()
            assert self.blob.shape == (IMU_LEN,)
        else:
            self.blob = np.zeros(IMU_LEN)
            pb, kb = _opt(bg); pa, ka = _opt(ba)
            rc = self._L.vxba_imu_init(self.blob, pb, pa)
            if rc:
                raise VxbaError(f"vxba_imu_init failed ({rc})")

    def field(self, name):
        o, n = self._FIELDS[name]
        a = self.blob[o:o + n]
        if n == 9:
            return a.reshape(3, 3).T
        if n == 225:
            return a.reshape(15, 15).T
        return a[0] if n == 1 else a

    def add_imu(self, cur_gyr, cur_acc, dt, noise_meas, noise_walk):
        """One bias-corrected mid-point sample (preintegration.hpp:75-135); noise_* are the 6x6 noiseMeas / noiseWalk."""
        rc = self._L.vxba_imu_add(self.blob, _c(cur_gyr), _c(cur_acc), float(dt), _c(np.asarray(noise_meas).T), _c(np.asarray(noise_walk).T))
        if rc:
            raise VxbaError(f"vxba_imu_add failed ({rc})")

    def give_evaluate(self, st1, st2, jac_enable=True):
        """Returns (residual, jtj (30,30), gg (30,)) -- jtj/gg are None without jac_enable (preintegration.hpp:137-212)."""
        r = C.c_double(0)
        if jac_enable:
            jtj = np.zeros((30, 30)); gg = np.zeros(30)
            rc = self._L.vxba_imu_evaluate(self.blob, _c(st1), _c(st2), 1, jtj.ctypes.data_as(C.c_void_p), gg.ctypes.data_as(C.c_void_p), C.byref(r))
            if rc:
                raise VxbaError(f"vxba_imu_evaluate failed ({_ERRNAMES.get(rc, rc)})")
            return r.value, jtj.T.copy(), gg
        rc = self._L.vxba_imu_evaluate(self.blob, _c(st1), _c(st2), 0, None, None, C.byref(r))
        if rc:
            raise VxbaError(f"vxba_imu_evaluate failed ({_ERRNAMES.get(rc, rc)})")
        return r.value, None, None

    def give_evaluate_g(self, st1, st2):
        """give_evaluate with the gravity columns (preintegration.hpp:214-294): (residual, jtj (33,33), gg (33,))."""
        r = C.c_double(0)
        jtj = np.zeros((33, 33)); gg = np.zeros(33)
        rc = self._L.vxba_imu_evaluate_g(self.blob, _c(st1), _c(st2), 1, jtj.ctypes.data_as(C.c_void_p), gg.ctypes.data_as(C.c_void_p), C.byref(r))
        if rc:
            raise VxbaError(f"vxba_imu_evaluate_g failed ({_ERRNAMES.get(rc, rc)})")
        return r.value, jtj.T.copy(), gg

    def update_state(self, dxi15):
        self._L.vxba_imu_update_state(self.blob, _c(dxi15))


def hess_plus(win_size, Hess15, JacT15, Hess6, JacT6):
    """LI_BA_Optimizer::hess_plus (voxel_map.hpp:455-463) on (r, c)-indexed numpy arrays; returns the updated copies."""
    H = _c(np.asarray(Hess15).T).copy(); J = _c(JacT15).copy()
    rc = load_library().vxba_hess_plus(int(win_size), H, J, _c(np.asarray(Hess6).T), _c(JacT6))
    if rc:
        raise VxbaError(f"vxba_hess_plus failed ({rc})")
    return H.T.copy(), J


class LI_BA_Optimizer:
    """The LiDAR-inertial LM shell (reference: voxel_map.hpp:446-655): 15 unknowns per frame, IMU factors between
    consecutive frames, voxel sweeps on the GPU."""

    def __init__(self, imu_coef: float = 1e-4):
        self.imu_coef = float(imu_coef)

    @staticmethod
    def _blobs(imus_factor):
        return _c(np.stack([f.blob for f in imus_factor])) if len(imus_factor) else np.zeros((0, IMU_LEN))

    def divide_thread(self, x_stats, voxhess: LidarFactor, imus_factor):
        """Returns (Hess (15W,15W), JacT (15W,), residual)."""
        n = LI_DIM * voxhess.win_size
        H = np.zeros((n, n)); J = np.zeros(n); r = C.c_double(0)
        voxhess._chk(voxhess._L.vxba_li_evaluate(voxhess.handle, _c(x_stats), self._blobs(imus_factor), self.imu_coef, H, J, C.byref(r)))
        return H.T.copy(), J, r.value

    def only_residual(self, x_stats, voxhess: LidarFactor, imus_factor):
        r = C.c_double(0)
        voxhess._chk(voxhess._L.vxba_li_only_residual(voxhess.handle, _c(x_stats), self._blobs(imus_factor), self.imu_coef, C.byref(r)))
        return r.value

    def damping_iter(self, x_stats, voxhess: LidarFactor, imus_factor, max_iter: int = 3):
        """Returns dict(states, hess, trace); the IMU factors' dbg/dba are updated in place like upstream (:608-609)."""
        W = voxhess.win_size
        n = LI_DIM * W
        st = _c(x_stats).copy()
        blobs = self._blobs(imus_factor)
        hess = np.empty((n, n)) if max_iter > 0 else np.zeros((n, n))   # written in full by the library when an iteration runs
        trace = np.zeros((max(max_iter, 1), TRACE_COLS)); nt = C.c_int(0)
        voxhess._chk(voxhess._L.vxba_li_damping_iter(voxhess.handle, st, blobs, self.imu_coef, int(max_iter), hess.ctypes.data_as(C.c_void_p),
                                                     trace.ctypes.data_as(C.c_void_p), C.byref(nt)))
        for f, b in zip(imus_factor, blobs):
            f.blob[:] = b
        return dict(states=st, hess=hess.T, trace=trace[: nt.value])   # hess: (r, c)-indexed view of the column-major output, no copy


class LI_BA_OptimizerGravity(LI_BA_Optimizer):
    """LI_BA_Optimizer with the gravity vector as three more unknowns (reference: voxel_map.hpp:658-864)."""

    def divide_thread(self, x_stats, voxhess: LidarFactor, imus_factor):
        """The (15W+3)-dimensional joint system (voxel_map.hpp:673-736): returns (Hess[r, c], JacT, residual)."""
        W = voxhess.win_size
        n = LI_DIM * W + 3
        Hess = np.zeros((n, n)); JacT = np.zeros(n); r = C.c_double(0)
        voxhess._chk(voxhess._L.vxba_li_evaluate_gravity(voxhess.handle, _c(x_stats), self._blobs(imus_factor), self.imu_coef, Hess, JacT, C.byref(r)))
        return Hess.T.copy(), JacT, r.value

    # only_residual: inherited -- give_evaluate_g without Jacobian is give_evaluate without Jacobian (voxel_map.hpp:738-773)

    def damping_iter(self, x_stats, voxhess: LidarFactor, imus_factor, max_iter: int = 2):
        W = voxhess.win_size
        n = LI_DIM * W + 3
        st = _c(x_stats).copy()
        blobs = self._blobs(imus_factor)
        hess = np.empty((n, n)) if max_iter > 0 else np.zeros((n, n))
        resis = np.zeros(2); trace = np.zeros((max(max_iter, 1), TRACE_COLS)); nt = C.c_int(0)
        voxhess._chk(voxhess._L.vxba_li_damping_iter_gravity(voxhess.handle, st, blobs, self.imu_coef, int(max_iter), hess.ctypes.data_as(C.c_void_p),
                                                             resis, trace.ctypes.data_as(C.c_void_p), C.byref(nt)))
        for f, b in zip(imus_factor, blobs):
            f.blob[:] = b
        return dict(states=st, hess=hess.T, resis=resis, trace=trace[: nt.value])


def damping_iter_generic(win_size: int, x_stats, hess_fn, resid_fn, max_iter: int = 3):
    """``Lidar_BA_Optimizer::damping_iter`` over caller-supplied sweeps (host-only; no GPU needed).

    ``hess_fn(xs (W,12)) -> (Hess[r,c] (n,n), JacT (n,), residual)`` and ``resid_fn(xs) -> residual`` -- typically a local
    voxel shard's sweep followed by an all-reduce (see :mod:`voxel_slam_amd.dist`)."""
    L = load_library()
    W = int(win_size); n = 6 * W
    Rp = _c(x_stats).copy()

    def _h(ctx, rp, packed):
        try:
            xs = np.ctypeslib.as_array(rp, shape=(W, 12)).copy()
            H, J, r = hess_fn(xs)
            out = np.ctypeslib.as_array(packed, shape=(n * n + n + 1,))
            out[: n * n] = np.asarray(H, dtype=np.float64).T.reshape(-1)   # column-major on the wire
            out[n * n: n * n + n] = J
            out[n * n + n] = r
            return 0
        except Exception:  # noqa: BLE001 - must not unwind through C
            import traceback
            traceback.print_exc()
            return 1

    def _r(ctx, rp, res):
        try:
            res[0] = float(resid_fn(np.ctypeslib.as_array(rp, shape=(W, 12)).copy()))
            return 0
        except Exception:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    hess = np.zeros((n, n)); resis = np.zeros(2); trace = np.zeros((max(max_iter, 1), TRACE_COLS))
    nt = C.c_int(0); conv = C.c_int(0)
    rc = L.vxba_damping_iter_generic(W, Rp, int(max_iter), _HESS_FN(_h), _RESID_FN(_r), None, hess, resis, trace, C.byref(nt), C.byref(conv))
    if rc != 0:
        raise VxbaError(f"vxba_damping_iter_generic failed: {_ERRNAMES.get(rc, rc)}")
    return dict(poses=Rp, hess=hess.T.copy(), resis=resis, trace=trace[: nt.value].copy(), is_converge=bool(conv.value))


LIO_SWEEP_LEN = 52
LIO_MAX_ITER = 4


def unpack_sweep(sw):
    """One sweep record -> dict(HTH 6x6, HTz 6, nnt 3x3, match_num)."""
    sw = np.asarray(sw)
    return {"HTH": sw[:36].reshape(6, 6).T.copy(), "HTz": sw[36:42].copy(), "nnt": sw[42:51].reshape(3, 3).T.copy(), "match_num": int(sw[51])}


class LioEstimator:
    """The odometry's point-to-plane update on the GPU: the plane map (``surf_map`` flattened to its leaves), the current scan
    (``pptr``) and ``lio_state_estimation`` (voxelslam.cpp:855-958) with ``match`` (voxel_map.hpp:1335-1392, 1674-1698).
    States are the flat VXBA state vectors of :func:`pack_state`, covariances 15x15 (tangent order [dphi dp dv dbg dba])."""

    def __init__(self, voxel_size: float = 1.0, max_layer: int = 2, device: int = 0):
        self._L = load_library()
        self._h = C.c_void_p()
        rc = self._L.vxba_lio_create(float(voxel_size), int(max_layer), int(device), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise VxbaError(f"vxba_lio_create failed: {_ERRNAMES.get(rc, rc)} (no CPU fallback exists; an MI355X is required)")
        self.voxel_size, self.max_layer = float(voxel_size), int(max_layer)

    def set_option(self, name: str, value: int):
        """VXBA_LIO_OPT_*: 'device_ekf' (1 = EKF update as a kernel, default; 0 = host algebra between sweeps)."""
        self._chk(self._L.vxba_lio_set_option(self._h, {"device_ekf": 0}[name], int(value)))

    def _chk(self, rc):
        if rc != 0:
            msg = self._L.vxba_lio_last_error(self._h)
            raise VxbaError(f"{_ERRNAMES.get(rc, rc)}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None):
            self._L.vxba_lio_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- map ------------------------------------------------------------------------------------------------------------
    def map_update(self, loc, layer, path, center, normal, plane_var, radius, is_plane=None):
        """Upsert leaves.  plane_var: n x 6 x 6 (numpy order; symmetric up to round-off, sent column-major)."""
        loc = np.ascontiguousarray(loc, dtype=np.int64).reshape(-1, 3)
        n = loc.shape[0]
        layer = np.ascontiguousarray(layer, dtype=np.int32); path = np.ascontiguousarray(path, dtype=np.int32)
        pv = np.ascontiguousarray(np.transpose(np.asarray(plane_var, dtype=np.float64).reshape(n, 6, 6), (0, 2, 1)))
        isp = None if is_plane is None else np.ascontiguousarray(is_plane, dtype=np.int32)
        self._chk(self._L.vxba_lio_map_update(self._h, n, loc, layer, path, None if isp is None else isp.ctypes.data_as(C.c_void_p), _c(center), _c(normal), pv, _c(radius)))

    def map_clear(self):
        self._chk(self._L.vxba_lio_map_clear(self._h))

    def map_size(self):
        a, b = C.c_int64(), C.c_int64()
        self._chk(self._L.vxba_lio_map_size(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- scan -----------------------------------------------------------------------------------------------------------
    def var_init(self, xyz, ext_R=None, ext_p=None, dept_err=0.02, beam_err=0.05):
        """``var_init`` (voxelslam.hpp:187-201) on sensor-frame float32 points."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        ext = None
        if ext_R is not None:
            ext = np.concatenate([np.asarray(ext_R, dtype=np.float64).T.reshape(9), np.zeros(3) if ext_p is None else np.asarray(ext_p, dtype=np.float64)])
        self._chk(self._L.vxba_lio_scan_raw(self._h, xyz.shape[0], xyz, None if ext is None else ext.ctypes.data_as(C.c_void_p), float(dept_err), float(beam_err)))

    def set_points(self, pnt, var):
        """Ready ``pointVar`` arrays: pnt n x 3, var n x 3 x 3."""
        pnt = _c(pnt).reshape(-1, 3)
        var = np.ascontiguousarray(np.transpose(np.asarray(var, dtype=np.float64).reshape(-1, 3, 3), (0, 2, 1)))
        self._chk(self._L.vxba_lio_scan_set(self._h, pnt.shape[0], pnt, var))

    def scan_size(self) -> int:
        return int(self._L.vxba_lio_scan_size(self._h))

    def read_points(self):
        n = self.scan_size()
        pnt = np.zeros((n, 3)); var = np.zeros((n, 9))
        self._chk(self._L.vxba_lio_scan_read(self._h, pnt, var))
        return pnt, np.transpose(var.reshape(n, 3, 3), (0, 2, 1)).copy()

    # -- the path -------------------------------------------------------------------------------------------------------
    @staticmethod
    def _cov(cov):
        return np.ascontiguousarray(np.asarray(cov, dtype=np.float64).reshape(15, 15).T)

    def sweep(self, state, cov, reset_cache=True, want_points=False):
        out = np.zeros(LIO_SWEEP_LEN)
        n = self.scan_size()
        pop = np.zeros(n, dtype=np.int32) if want_points else None
        sig = np.zeros(n) if want_points else None
        self._chk(self._L.vxba_lio_sweep(self._h, _c(state), self._cov(cov), 1 if reset_cache else 0, out,
                                         None if pop is None else pop.ctypes.data_as(C.c_void_p), None if sig is None else sig.ctypes.data_as(C.c_void_p)))
        res = unpack_sweep(out)
        if want_points:
            res["plane_of_point"] = pop; res["sigma_of_point"] = sig
        return res

    def lio_state_estimation(self, state, cov):
        """Returns dict(ok, state, cov, iterations, match_num, min_eig, sweeps)."""
        st = _c(state).copy(); cv = self._cov(cov).copy()
        info = np.zeros(4); sweeps = np.zeros((LIO_MAX_ITER, LIO_SWEEP_LEN))
        self._chk(self._L.vxba_lio_state_estimation(self._h, st, cv, info.ctypes.data_as(C.c_void_p), sweeps.ctypes.data_as(C.c_void_p)))
        it = int(info[1])
        return {"ok": bool(info[0]), "state": st, "cov": cv.T.copy(), "iterations": it, "match_num": int(info[2]), "min_eig": float(info[3]),
                "sweeps": [unpack_sweep(sweeps[k]) for k in range(it)]}

    def pvec_update(self, state, cov, with_var: bool = True, resident: bool = False):
        """World points (and, unless ``with_var`` is False, world covariances n x 3 x 3) of the scan; both stay on the device for leaf_stats
        and LocalMap.cut_voxel_lio.  ``resident=True`` returns nothing: the result only stays on the device."""
        n = self.scan_size()
        if resident:
            self._chk(self._L.vxba_lio_pvec_update(self._h, _c(state), self._cov(cov), None, None))
            return None
        pw = np.zeros((n, 3))
        if not with_var:
            self._chk(self._L.vxba_lio_pvec_update(self._h, _c(state), self._cov(cov), pw.ctypes.data_as(C.c_void_p), None))
            return pw
        var = np.zeros((n, 9))
        self._chk(self._L.vxba_lio_pvec_update(self._h, _c(state), self._cov(cov), pw.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p)))
        return pw, np.transpose(var.reshape(n, 3, 3), (0, 2, 1)).copy()

    def leaf_stats(self, cell_ptr, order):
        """Per-leaf increments of ``cut_voxel`` (OctoTree::push, voxel_map.hpp:969-993) from the resident world points / covariances of the
        last pvec_update: leaf c owns scan points ``order[cell_ptr[c]:cell_ptr[c+1]]``.  Returns clusters (n_cells x 10), cov_add (n_cells x 9 x 9)."""
        cp = np.ascontiguousarray(cell_ptr, dtype=np.int64); n = cp.shape[0] - 1
        od = np.ascontiguousarray(order, dtype=np.int32)
        if n < 0 or (n >= 0 and od.shape[0] != (cp[-1] if cp.size else 0)):
            raise VxbaError("leaf_stats: order must hold cell_ptr[-1] indices")
        cl = np.zeros((max(n, 0), 10)); ca = np.zeros((max(n, 0), 81))
        self._chk(self._L.vxba_lio_leaf_stats(self._h, n, cp, od.ctypes.data_as(C.c_void_p), cl, ca))
        return cl, ca.reshape(-1, 9, 9)          # symmetric: column-major == row-major, no transpose needed


def down_sampling_voxel(xyz, voxel_size: float, device: int = 0):
    """``down_sampling_voxel`` (tools.hpp:201-238) on float32 points; one point per occupied voxel, ascending voxel index."""
    L = load_library()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.zeros_like(xyz)
    n_out = C.c_int64()
    rc = L.vxba_down_sampling_voxel(device, xyz.shape[0], xyz, float(voxel_size), out, C.byref(n_out))
    if rc != 0:
        raise VxbaError(f"vxba_down_sampling_voxel: {_ERRNAMES.get(rc, rc)}")
    return out[: n_out.value].copy()


def voxelize_profile(enable: bool):
    """``vxba_voxelize_profile``: start (True) or stop (False -> dict(ms_sum, launches, algorithmic_bytes)) the measurement of the
    cluster-build kernel inside the voxeliser."""
    L = load_library()
    ms, n, b = C.c_double(), C.c_longlong(), C.c_double()
    L.vxba_voxelize_profile(1 if enable else 0, C.byref(ms), C.byref(n), C.byref(b))
    return None if enable else dict(ms_sum=ms.value, launches=int(n.value), algorithmic_bytes=b.value)


class HbaSession:
    """``vxba_hba_*``: a session of keyframes resident on the device and the bottom-up pass of the hierarchical global BA over it
    (thd_globalmapping / HBA_add_edge, voxelslam.cpp:2485-2595, 2320-2482) below the C ABI -- see csrc/vxba_hba.hip.  The Python twin that
    orchestrates the same calls window by window (and runs on the CPU oracle in the tests) is ``voxel_slam_amd.hba.hierarchical_ba``."""

    def __init__(self, device: int = 0):
        L = load_library()
        L.vxba_hba_last_error.restype = C.c_char_p
        self._L = L
        self._h = C.c_void_p()
        rc = L.vxba_hba_create(int(device), C.byref(self._h))
        if rc != 0:
            raise VxbaError(f"vxba_hba_create: {_ERRNAMES.get(rc, rc)}")

    def close(self):
        if getattr(self, "_h", None):
            self._L.vxba_hba_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise VxbaError(f"{what}: {_ERRNAMES.get(rc, rc)}: {self._L.vxba_hba_last_error(self._h).decode()}")

    def add_keyframes(self, clouds):
        """clouds: list of (n_i, 3) arrays in keyframe coordinates (stored as float32, like PointType)."""
        ptr = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).astype(np.int64)
        xyz = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.float32).reshape(-1, 3) for c in clouds])) if len(clouds) else np.zeros((0, 3), np.float32)
        self._check(self._L.vxba_hba_add_keyframes(self._h, C.c_int64(len(clouds)), ptr.ctypes.data_as(C.c_void_p), xyz.ctypes.data_as(C.c_void_p)), "vxba_hba_add_keyframes")

    def num_keyframes(self) -> int:
        return int(self._L.vxba_hba_num_keyframes(self._h))

    def clear(self):
        self._check(self._L.vxba_hba_clear(self._h), "vxba_hba_clear")

    @staticmethod
    def windows(K: int, wdsize: int, mgsize: int, tail: bool = True):
        """``vxba_hba_num_windows`` / ``vxba_hba_window``: [(first keyframe, keyframe count)] of a pass over K keyframes -- the full windows at stride
        ``mgsize`` and, with ``tail``, the closing short window of upstream's last iteration (voxelslam.cpp:2519-2523)."""
        L = load_library()
        n = L.vxba_hba_num_windows(int(K), int(wdsize), int(mgsize), int(bool(tail)))
        out = []
        for w in range(n):
            f0, c = C.c_int(), C.c_int()
            L.vxba_hba_window(int(K), int(wdsize), int(mgsize), int(bool(tail)), w, C.byref(f0), C.byref(c))
            out.append((f0.value, c.value))
        return out

    @staticmethod
    def _edges(eij, edata, lo, hi, win=None):
        return [dict(i=int(eij[k, 0]), j=int(eij[k, 1]), rot=edata[k, :9].reshape(3, 3).copy(), tra=edata[k, 9:12].copy(), v6=edata[k, 12:18].copy(),
                     **({} if win is None else {"window": int(win[k])})) for k in range(lo, hi)]

    def _poses(self, poses):
        poses = _c(poses).reshape(-1, 12)
        if poses.shape[0] != self.num_keyframes():
            raise VxbaError(f"hba pass: {poses.shape[0]} poses for {self.num_keyframes()} keyframes")
        return poses

    def run_pass(self, poses, coarse: VoxelizeParams, fine: VoxelizeParams, wdsize: int = 10, mgsize: int = 5, top_max_iter: int = 1, n_threads: int = 0,
                 tail: bool = True):
        """One bottom-up pass on this device (``vxba_hba_pass``); returns what ``hba.hierarchical_ba`` returns (edges as dicts with keyframe indices)."""
        poses = self._poses(poses)
        K = poses.shape[0]
        wins = self.windows(K, wdsize, mgsize, tail)
        S = len(wins)
        if S < 1:
            raise VxbaError(f"hba pass: no window for {K} keyframes with wdsize {wdsize}, mgsize {mgsize}, tail {tail}")
        cap = sum(c * (c - 1) // 2 for _, c in wins) + S * (S - 1) // 2 + 1
        sub_poses = np.zeros((S, 12)); sizes = np.zeros(S, dtype=np.int64)
        eij = np.zeros((cap, 2), dtype=np.int32); edata = np.zeros((cap, 18))
        n1, n2, ntr = C.c_int64(), C.c_int64(), C.c_int()
        rounds = np.zeros((max(1, top_max_iter), 5))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self._L.vxba_hba_pass(self._h, vp(poses), C.byref(coarse), C.byref(fine), int(wdsize), int(mgsize), int(bool(tail)), int(top_max_iter), int(n_threads),
                                          vp(sub_poses), vp(sizes), C.c_int64(cap), vp(eij), vp(edata), C.byref(n1), C.byref(n2), vp(rounds), C.byref(ntr)), "vxba_hba_pass")
        return dict(edges1=self._edges(eij, edata, 0, n1.value), edges2=self._edges(eij, edata, n1.value, n1.value + n2.value), submap_ids=[f0 for f0, _ in wins],
                    submap_poses=sub_poses, submap_sizes=[int(x) for x in sizes], n_threads_used=int(self._L.vxba_hba_threads_used(self._h)),
                    top_rounds=self._rounds(rounds, ntr.value))

    @staticmethod
    def _rounds(rounds, n):
        return [dict(round=k, n_voxels=int(rounds[k, 0]), resis=(float(rounds[k, 1]), float(rounds[k, 2])), converged=bool(rounds[k, 3]), fine=bool(rounds[k, 4])) for k in range(n)]

    # ---- the pass in its two halves (one rank of N: voxel_slam_amd.dist.hba_pass) ----
    def bottom(self, poses, coarse: VoxelizeParams, fine: VoxelizeParams, wdsize: int = 10, mgsize: int = 5, tail: bool = True, w_first: int = 0, w_stride: int = 1,
               n_threads: int = 0):
        """``vxba_hba_bottom``: the windows w_first, w_first + w_stride, .. on this device.  Returns dict(sizes (S, -1 where not run here), edges (dicts with
        ``window``), windows)."""
        poses = self._poses(poses)
        wins = self.windows(poses.shape[0], wdsize, mgsize, tail)
        S = len(wins)
        cap = sum(c * (c - 1) // 2 for _, c in wins[w_first::w_stride]) + 1
        sub_poses = np.zeros((S, 12)); sizes = np.full(S, -1, dtype=np.int64)
        eij = np.zeros((cap, 2), dtype=np.int32); edata = np.zeros((cap, 18)); ewin = np.zeros(cap, dtype=np.int32)
        ne = C.c_int64()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self._last_geom = (poses.shape[0], int(wdsize), int(mgsize), int(bool(tail)))
        self._check(self._L.vxba_hba_bottom(self._h, vp(poses), C.byref(coarse), C.byref(fine), int(wdsize), int(mgsize), int(bool(tail)), int(w_first), int(w_stride),
                                            int(n_threads), vp(sub_poses), vp(sizes), C.c_int64(cap), vp(eij), vp(edata), vp(ewin), C.byref(ne)), "vxba_hba_bottom")
        return dict(sizes=sizes, edges=self._edges(eij, edata, 0, ne.value, ewin), windows=wins, n_threads_used=int(self._L.vxba_hba_threads_used(self._h)))

    def export_submaps(self, w_first: int, w_stride: int, d_ptr: int, capacity_points: int) -> int:
        """``vxba_hba_export_submaps``: this selection's submaps packed into the DEVICE buffer at ``d_ptr`` (float xyz); returns the point count."""
        n = C.c_int64()
        self._check(self._L.vxba_hba_export_submaps(self._h, int(w_first), int(w_stride), C.c_void_p(d_ptr), C.c_int64(capacity_points), C.byref(n)), "vxba_hba_export_submaps")
        return int(n.value)

    def import_submaps(self, w_first: int, w_stride: int, sizes, d_ptr: int):
        """``vxba_hba_import_submaps``: a peer's packed submaps (DEVICE pointer) into place; ``sizes``: S entries."""
        sizes = np.ascontiguousarray(sizes, dtype=np.int64)
        self._check(self._L.vxba_hba_import_submaps(self._h, int(w_first), int(w_stride), sizes.ctypes.data_as(C.c_void_p), C.c_void_p(d_ptr)), "vxba_hba_import_submaps")

    def top_factor(self) -> "LidarFactor":
        """``vxba_hba_top_factor``: the (session-owned) factor the top level runs on, to attach a collective to."""
        h = C.c_void_p()
        self._check(self._L.vxba_hba_top_factor(self._h, C.byref(h)), "vxba_hba_top_factor")
        return LidarFactor.from_handle(h)

    def top(self, poses, coarse: VoxelizeParams, fine: VoxelizeParams, top_max_iter: int = 1):
        """``vxba_hba_top``: the top level over all submaps of the pass in progress.  Returns dict(submap_poses, edges2, top_rounds)."""
        poses = self._poses(poses)
        S = self._L.vxba_hba_num_windows(*self._geom(poses.shape[0]))
        cap = S * (S - 1) // 2 + 1
        sub_poses = np.zeros((S, 12))
        eij = np.zeros((cap, 2), dtype=np.int32); edata = np.zeros((cap, 18))
        ne, ntr = C.c_int64(), C.c_int()
        rounds = np.zeros((max(1, top_max_iter), 5))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self._L.vxba_hba_top(self._h, vp(poses), C.byref(coarse), C.byref(fine), int(top_max_iter), vp(sub_poses), C.c_int64(cap), vp(eij), vp(edata), C.byref(ne),
                                         vp(rounds), C.byref(ntr)), "vxba_hba_top")
        return dict(submap_poses=sub_poses, edges2=self._edges(eij, edata, 0, ne.value), top_rounds=self._rounds(rounds, ntr.value))

    def _geom(self, K):
        g = getattr(self, "_last_geom", None)
        if not g or g[0] != K:
            raise VxbaError("hba top: no pass in progress (HbaSession.bottom first)")
        return g


def cov_add_build(xyz_world, var, cell_ptr, device: int = 0):
    """cov_add of OctoTree::push (voxel_map.hpp:91-106, 990-992) per cell: n_cells x 9 x 9."""
    L = load_library()
    cp = np.ascontiguousarray(cell_ptr, dtype=np.int64); n = cp.shape[0] - 1
    xyz = _c(xyz_world).reshape(-1, 3)
    var9 = np.ascontiguousarray(np.transpose(np.asarray(var, dtype=np.float64).reshape(-1, 3, 3), (0, 2, 1)))
    out = np.zeros((n, 81))
    rc = L.vxba_cov_add_build(device, n, xyz.shape[0], xyz, var9, cp, out)
    if rc != 0:
        raise VxbaError(f"vxba_cov_add_build: {_ERRNAMES.get(rc, rc)}")
    return np.transpose(out.reshape(n, 9, 9), (0, 2, 1)).copy()


def plane_update(clusters, eig_val, eig_vec, cov_add, device: int = 0):
    """OctoTree::plane_update (voxel_map.hpp:1118-1146) batched: dict(center, normal, plane_var n x 6 x 6, radius)."""
    L = load_library()
    cl = _c(clusters).reshape(-1, 10); n = cl.shape[0]
    ca = np.ascontiguousarray(np.transpose(np.asarray(cov_add, dtype=np.float64).reshape(n, 9, 9), (0, 2, 1))).reshape(n, 81)
    center = np.zeros((n, 3)); normal = np.zeros((n, 3)); pv = np.zeros((n, 36)); rad = np.zeros(n)
    rc = L.vxba_plane_update(device, n, cl, _c(eig_val).reshape(n, 3), _c(eig_vec).reshape(n, 9), ca, center, normal, pv, rad)
    if rc != 0:
        raise VxbaError(f"vxba_plane_update: {_ERRNAMES.get(rc, rc)}")
    return dict(center=center, normal=normal, plane_var=np.transpose(pv.reshape(n, 6, 6), (0, 2, 1)).copy(), radius=rad)


def rccl_unique_id(librccl_path: str) -> bytes:
    """128-byte ncclUniqueId from the given librccl.so (call on one rank, broadcast to the others)."""
    L = load_library()
    buf = C.create_string_buffer(128)
    rc = L.vxba_rccl_unique_id(librccl_path.encode(), C.cast(buf, C.c_void_p))
    if rc != 0:
        raise VxbaError(f"vxba_rccl_unique_id failed: {_ERRNAMES.get(rc, rc)}")
    return buf.raw


def plane_fit(clusters, device: int = 0):
    """K4: batched eig(cluster.cov()) -> (eig_val (n,3), eig_vec (n,9 col-major))."""
    L = load_library()
    cl = _c(clusters).reshape(-1, 10)
    n = cl.shape[0]
    ev = np.zeros((n, 3)); U = np.zeros((n, 9))
    rc = L.vxba_plane_fit(int(device), n, cl, ev, U)
    if rc != 0:
        raise VxbaError(f"vxba_plane_fit failed: {_ERRNAMES.get(rc, rc)}")
    return ev, U


def plane_fit_judge(clusters, min_point=5, min_eigen_value=0.0025, eigen_ratio_thre=1.0, factor_ratio_max=0.12, device: int = 0):
    """K4 + the reference's plane criteria: (eig_val, eig_vec, flags) with flags bits 1 = N > min_point,
    2 = plane_judge (voxel_map.hpp:1015-1019), 4 = lambda0/lambda1 <= factor_ratio_max (voxel_map.hpp:1314)."""
    L = load_library()
    cl = _c(clusters).reshape(-1, 10)
    n = cl.shape[0]
    ev = np.zeros((n, 3)); U = np.zeros((n, 9)); fl = np.zeros(n, dtype=np.uint8)
    rc = L.vxba_plane_fit_judge(int(device), n, cl, int(min_point), float(min_eigen_value), float(eigen_ratio_thre), float(factor_ratio_max), ev, U, fl)
    if rc != 0:
        raise VxbaError(f"vxba_plane_fit_judge failed: {_ERRNAMES.get(rc, rc)}")
    return ev, U, fl


def build_clusters(xyz, cell_ptr, device: int = 0):
    """Stand-alone K1: packed clusters (n_cells, 10) of bucketed points."""
    L = load_library()
    xyz = _c(xyz).reshape(-1, 3)
    ptr = np.ascontiguousarray(cell_ptr, dtype=np.int64)
    out = np.zeros((ptr.shape[0] - 1, 10))
    rc = L.vxba_build_clusters(int(device), ptr.shape[0] - 1, xyz.shape[0], xyz, ptr, out)
    if rc != 0:
        raise VxbaError(f"vxba_build_clusters failed: {_ERRNAMES.get(rc, rc)}")
    return out


def debug_mfma_probe(A, B, device: int = 0):
    """D = A (16x4) @ B (4x16) through one f64 MFMA with the lane maps K3 assumes."""
    L = load_library()
    D = np.zeros((16, 16))
    rc = L.vxba_debug_mfma_probe(int(device), _c(A).reshape(16, 4), _c(B).reshape(4, 16), D)
    if rc != 0:
        raise VxbaError(f"vxba_debug_mfma_probe failed: {_ERRNAMES.get(rc, rc)}")
    return D


def debug_stamps(n_waves: int, clear: bool = False):
    """Per-wave s_memtime stamps of the instrumented kernels: array (n_waves, 32) of uint64."""
    L = load_library()
    out = np.zeros((n_waves, 32), dtype=np.uint64)
    L.vxba_debug_stamps(int(clear), out.ctypes.data_as(C.c_void_p), out.size)
    return out


class LocalMap:
    """The incremental local map on the GPU (include/vxba.h ``vxba_map_*``; reference: ``surf_map`` / ``surf_map_slide`` of OctoTree nodes,
    voxel_map.hpp:896-1639, driven as voxelslam.cpp:1592-1700 does)."""

    def __init__(self, voxel_size=1.0, max_layer=2, min_point=(5, 5, 5, 5), min_eigen_value=0.0025, plane_eigen_value_thre=(0.25, 0.25, 0.25, 0.25),
                 max_points=100, win_size=10, thread_num=5, device=0):
        self._L = load_library()
        self.win_size = int(win_size)
        p = MapParams(float(voxel_size), int(max_layer), (C.c_double * 4)(*min_point), float(min_eigen_value), (C.c_double * 4)(*plane_eigen_value_thre),
                      int(max_points), int(win_size), int(thread_num))
        self._h = C.c_void_p()
        rc = self._L.vxba_map_create(C.byref(p), int(device), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise VxbaError(f"vxba_map_create failed: {_ERRNAMES.get(rc, rc)} (no CPU fallback exists; an MI355X is required)")

    def _chk(self, rc):
        if rc != 0:
            msg = self._L.vxba_map_last_error(self._h)
            raise VxbaError(f"{_ERRNAMES.get(rc, rc)}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None):
            self._L.vxba_map_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cut_voxel(self, ord_, pnt_body, var_world, pwld):
        pnt = np.ascontiguousarray(pnt_body, dtype=np.float64).reshape(-1, 3)
        var = np.ascontiguousarray(np.transpose(np.asarray(var_world, dtype=np.float64).reshape(-1, 3, 3), (0, 2, 1)))
        self._chk(self._L.vxba_map_cut_voxel(self._h, int(ord_), pnt.shape[0], pnt, var.reshape(-1, 9), np.ascontiguousarray(pwld, dtype=np.float64).reshape(-1, 3)))

    def cut_voxel_lio(self, ord_, est: "LioEstimator"):
        """cut_voxel_multi on the scan resident in the odometry handle after ``est.pvec_update(..., resident=True)``."""
        self._chk(self._L.vxba_map_cut_voxel_lio(self._h, int(ord_), est._h))

    def export_planes(self, est: "LioEstimator") -> int:
        """Plane records of the changed part of the tree into the odometry's plane map, on the device."""
        n = C.c_int64(0)
        self._chk(self._L.vxba_map_export_planes(self._h, est._h, C.byref(n)))
        return int(n.value)

    def recut(self, win_count, poses, factor: "LidarFactor"):
        n = C.c_int64(0)
        self._chk(self._L.vxba_map_recut(self._h, int(win_count), np.ascontiguousarray(poses, dtype=np.float64)[:win_count], factor._h, C.byref(n)))
        return int(n.value)

    def margi(self, win_count, poses, factor: "LidarFactor"):
        self._chk(self._L.vxba_map_margi(self._h, int(win_count), np.ascontiguousarray(poses, dtype=np.float64)[:win_count], factor._h))

    def slide(self, mgsize=1):
        self._chk(self._L.vxba_map_slide(self._h, int(mgsize)))

    def counts(self):
        out = np.zeros(4, dtype=np.int64)
        self._chk(self._L.vxba_map_counts(self._h, out))
        return dict(roots=int(out[0]), slide=int(out[1]), leaves=int(out[2]), mp0=int(out[3]))

    def fix_pool(self):
        """Pool of the marginalised points: cursor / capacity in points, compactions so far (include/vxba.h ``vxba_map_fix_pool``)."""
        out = np.zeros(3, dtype=np.int64)
        self._chk(self._L.vxba_map_fix_pool(self._h, out))
        return dict(cursor=int(out[0]), capacity=int(out[1]), compactions=int(out[2]))

    def set_journey(self, jour: float):
        """The journey odometer the next margi stamps on the slide map's roots (voxelslam.cpp:1349, 1677)."""
        self._chk(self._L.vxba_map_set_journey(self._h, float(jour)))

    def release(self, jour_now: float, min_age: int = 700, est: "LioEstimator | None" = None):
        """The release branch of the local-mapping loop (voxelslam.cpp:1503-1523): roots last seen ``min_age`` journeys ago leave the map,
        the node pool is compacted.  Returns dict(roots, nodes) released."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self._L.vxba_map_release(self._h, float(jour_now), int(min_age), est._h if est is not None else None, C.byref(a), C.byref(b)))
        return dict(roots=int(a.value), nodes=int(b.value))

    def device_bytes(self):
        out = np.zeros(5, dtype=np.int64)
        self._chk(self._L.vxba_map_device_bytes(self._h, out))
        return dict(nodes=int(out[0]), fix_pool=int(out[1]), scans=int(out[2]), other=int(out[3]), total=int(out[4]))

    def leaves(self):
        """Every leaf, sorted by node id, as a dictionary of arrays (fields of include/vxba.h ``vxba_map_leaves``)."""
        W = self.win_size
        n = C.c_int64(0)
        self._chk(self._L.vxba_map_leaves(self._h, 0, None, None, None, C.byref(n)))
        n = int(n.value)
        ids = np.zeros(n, dtype=np.uint64); ints = np.zeros((n, 8), dtype=np.int32); d = np.zeros((n, 156 + 11 * W))
        got = C.c_int64(0)
        if n:
            self._chk(self._L.vxba_map_leaves(self._h, n, ids.ctypes.data_as(C.c_void_p), ints.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), C.byref(got)))
            assert got.value == n
        o = np.argsort(ids, kind="stable")
        ids, ints, d = ids[o], ints[o], d[o]
        return dict(node_id=ids, layer=ints[:, 0], isexist=ints[:, 1].astype(bool), is_plane=ints[:, 2].astype(bool), has_sw=ints[:, 3].astype(bool),
                    opt_state=ints[:, 4], last_num=ints[:, 5], n_point_fix=ints[:, 6], in_slide=ints[:, 7].astype(bool),
                    pcr_add=d[:, 0:10], pcr_fix=d[:, 10:20], eig_val=d[:, 20:23], eig_vec=d[:, 23:32], center=d[:, 32:35], normal=d[:, 35:38],
                    radius=d[:, 38], plane_var=np.transpose(d[:, 39:75].reshape(n, 6, 6), (0, 2, 1)), cov_add=np.transpose(d[:, 75:156].reshape(n, 9, 9), (0, 2, 1)),
                    pcrs_local=d[:, 156:156 + 10 * W].reshape(n, W, 10), n_points=d[:, 156 + 10 * W:].astype(np.int64))
