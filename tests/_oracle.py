"""ctypes driver for the CPU oracle (oracle/liboracle.so).  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None
# tests/_ref.py loads a second instance of this module with these three overridden: the same wrapper then drives
# oracle/_ref/libref.so (the reference's own headers compiled in place), which exports the same vxo_* names for the part it covers.
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
MAKE_TARGET = "all"
PARTIAL = False

f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), MAKE_TARGET])


class _Partial:
    """CDLL proxy for a library that exports only part of the surface: prototypes of missing symbols are dropped silently,
    calling one raises AttributeError."""

    class _Sink:
        pass

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)

    def __getattr__(self, name):
        try:
            return getattr(self._cdll, name)
        except AttributeError:
            if name.startswith("vxo_"):
                return _Partial._Sink()
            raise


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = LIB_PATH
    if not os.path.exists(path):
        build_oracle()
    else:
        try:                                   # a stale checker (sources newer than the .so) is rebuilt; a box without make keeps what it has
            build_oracle()
        except (OSError, subprocess.CalledProcessError):
            pass
    L = C.CDLL(path)
    if PARTIAL:
        L = _Partial(L)
    L.vxo_create.restype = C.c_void_p
    L.vxo_create.argtypes = [C.c_int]
    L.vxo_destroy.argtypes = [C.c_void_p]
    L.vxo_clear.argtypes = [C.c_void_p]
    L.vxo_size.argtypes = [C.c_void_p]
    L.vxo_push_voxels.argtypes = [C.c_void_p, C.c_int, f64p, f64p, f64p, f64p, f64p, f64p]
    L.vxo_acc_evaluate2.argtypes = [C.c_void_p, f64p, C.c_int, C.c_int, f64p, f64p, C.POINTER(C.c_double)]
    L.vxo_evaluate_only_residual.argtypes = [C.c_void_p, f64p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.vxo_read_cache.argtypes = [C.c_void_p, C.c_int, C.c_int, f64p, f64p, f64p]
    L.vxo_divide_thread.restype = C.c_double
    L.vxo_divide_thread.argtypes = [C.c_void_p, f64p, C.c_int, f64p, f64p]
    L.vxo_only_residual.restype = C.c_double
    L.vxo_only_residual.argtypes = [C.c_void_p, f64p, C.c_int]
    L.vxo_damping_iter.argtypes = [C.c_void_p, f64p, C.c_int, C.c_int, f64p, f64p, f64p, C.POINTER(C.c_int)]
    L.vxo_time_ba_iteration.restype = C.c_double
    L.vxo_time_ba_iteration.argtypes = [C.c_void_p, f64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.vxo_eig_sym3.argtypes = [f64p, f64p, f64p]
    L.vxo_exp.argtypes = [f64p, f64p]
    L.vxo_log.argtypes = [f64p, f64p]
    L.vxo_ldlt_solve.argtypes = [C.c_int, f64p, f64p, f64p]
    L.vxo_cluster_transform.argtypes = [f64p, f64p, f64p]
    L.vxo_build_clusters.argtypes = [C.c_int64, i64p, f64p, f64p]
    L.vxo_plane_fit.argtypes = [C.c_int64, f64p, f64p, f64p]
    L.vxo_jr.argtypes = [f64p, f64p]
    L.vxo_jr_inv.argtypes = [f64p, f64p]
    L.vxo_mat_inverse.argtypes = [C.c_int, f64p, f64p]
    L.vxo_imu_init.argtypes = [f64p, f64p, f64p]
    L.vxo_imu_add.argtypes = [f64p, f64p, f64p, C.c_double, f64p, f64p]
    L.vxo_imu_evaluate.restype = C.c_double
    L.vxo_imu_evaluate.argtypes = [f64p, f64p, f64p, f64p, f64p, C.c_int]
    L.vxo_li_divide_thread.restype = C.c_double
    L.vxo_li_divide_thread.argtypes = [C.c_void_p, f64p, f64p, C.c_int, C.c_double, f64p, f64p]
    if hasattr(L, "vxo_li_divide_thread_gravity"):      # (a libref.so built before round 4 lacks the two gravity members)
        L.vxo_li_divide_thread_gravity.restype = C.c_double
        L.vxo_li_divide_thread_gravity.argtypes = [C.c_void_p, f64p, f64p, C.c_int, C.c_double, f64p, f64p]
        L.vxo_li_only_residual_gravity.restype = C.c_double
        L.vxo_li_only_residual_gravity.argtypes = [C.c_void_p, f64p, f64p, C.c_int, C.c_double]
    L.vxo_li_only_residual.restype = C.c_double
    L.vxo_li_only_residual.argtypes = [C.c_void_p, f64p, f64p, C.c_int, C.c_double]
    L.vxo_li_damping_iter.argtypes = [C.c_void_p, f64p, f64p, C.c_int, C.c_double, C.c_int, f64p, f64p, C.POINTER(C.c_int)]
    L.vxo_imu_evaluate_g.restype = C.c_double
    L.vxo_imu_evaluate_g.argtypes = [f64p, f64p, f64p, f64p, f64p, C.c_int]
    L.vxo_li_damping_iter_gravity.argtypes = [C.c_void_p, f64p, f64p, C.c_int, C.c_double, C.c_int, f64p, f64p, f64p, C.POINTER(C.c_int)]
    L.vxo_voxelize.restype = C.c_int64
    L.vxo_voxelize.argtypes = [C.c_int, C.c_int64, f64p, i64p, f64p, f64p, C.c_int64, np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS"),
                               f64p, f64p, f64p, f64p]
    i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    vp = C.c_void_p
    L.vxo_lio_create.restype = vp
    L.vxo_lio_create.argtypes = [C.c_double, C.c_int]
    L.vxo_lio_destroy.argtypes = [vp]
    L.vxo_lio_map_add.argtypes = [vp, C.c_int64, i64p, i32p, i32p, vp, f64p, f64p, f64p, f64p]
    L.vxo_lio_scan_set.argtypes = [vp, C.c_int64, f64p, f64p]
    L.vxo_lio_scan_raw.argtypes = [vp, C.c_int64, np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS"), f64p, C.c_double, C.c_double]
    L.vxo_lio_scan_size.restype = C.c_int64
    L.vxo_lio_scan_size.argtypes = [vp]
    L.vxo_lio_scan_read.argtypes = [vp, f64p, f64p]
    L.vxo_lio_sweep.argtypes = [vp, f64p, f64p, C.c_int, f64p, i32p, f64p]
    L.vxo_lio_state_estimation.argtypes = [vp, f64p, f64p, f64p, f64p, i32p, f64p]
    L.vxo_time_lio_state_estimation.restype = C.c_double
    L.vxo_time_lio_state_estimation.argtypes = [vp, f64p, f64p, C.c_int]
    L.vxo_lio_pvec_update.argtypes = [vp, f64p, f64p, f64p, f64p]
    L.vxo_cov_add_build.argtypes = [C.c_int64, i64p, f64p, f64p, f64p]
    f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
    L.vxo_down_sampling_voxel.restype = C.c_int64
    L.vxo_down_sampling_voxel.argtypes = [C.c_int64, f32p, C.c_double, f32p]
    L.vxo_plane_update.argtypes = [C.c_int64, f64p, f64p, f64p, f64p, f64p, f64p, f64p, f64p]
    u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
    L.vxo_localmap_create.restype = vp
    L.vxo_localmap_create.argtypes = [f64p]
    L.vxo_localmap_destroy.argtypes = [vp]
    L.vxo_localmap_cut_voxel.argtypes = [vp, C.c_int, C.c_int64, f64p, f64p, f64p]
    L.vxo_localmap_recut.argtypes = [vp, C.c_int, f64p, vp]
    L.vxo_localmap_margi.argtypes = [vp, C.c_int, f64p, vp]
    L.vxo_localmap_slide.argtypes = [vp, C.c_int]
    L.vxo_localmap_counts.argtypes = [vp, i64p]
    L.vxo_localmap_leaves.restype = C.c_int64
    L.vxo_localmap_leaves.argtypes = [vp, C.c_int64, u64p, i32p, f64p]
    L.vxo_localmap_leaf_points.restype = C.c_int64
    L.vxo_localmap_leaf_points.argtypes = [vp, C.c_uint64, C.c_int, C.c_int64, f64p]
    _LIB = L
    return L


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """Mirror of the reference LidarFactor + Lidar_BA_Optimizer interface on the CPU oracle."""

    def __init__(self, win_size):
        self.win_size = int(win_size)
        self._h = lib().vxo_create(self.win_size)

    def __del__(self):
        try:                                   # at interpreter shutdown the module globals may already be gone
            if getattr(self, "_h", None):
                lib().vxo_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def clear(self):
        lib().vxo_clear(self._h)

    def size(self):
        return lib().vxo_size(self._h)

    def push_voxels(self, clusters, fix, coe, eig_val=None, eig_vec=None, merged=None):
        clusters = _c(clusters)
        n = clusters.shape[0]
        assert clusters.shape == (n, self.win_size, 10)
        eig_val = np.zeros((n, 3)) if eig_val is None else _c(eig_val)
        eig_vec = np.tile(np.eye(3).reshape(1, 9), (n, 1)) if eig_vec is None else _c(eig_vec)
        merged = np.zeros((n, 10)) if merged is None else _c(merged)
        lib().vxo_push_voxels(self._h, n, clusters, _c(fix), _c(coe), eig_val, eig_vec, merged)

    def acc_evaluate2(self, Rp, head=0, end=None):
        end = self.size() if end is None else end
        n = 6 * self.win_size
        H = np.zeros((n, n)); J = np.zeros(n); r = C.c_double(0)
        lib().vxo_acc_evaluate2(self._h, _c(Rp), head, end, H, J, C.byref(r))
        return H.T.copy(), J, r.value  # H returned col-major -> transpose view gives H[r,c]

    def evaluate_only_residual(self, Rp, head=0, end=None):
        end = self.size() if end is None else end
        r = C.c_double(0)
        lib().vxo_evaluate_only_residual(self._h, _c(Rp), head, end, C.byref(r))
        return r.value

    def read_cache(self, head=0, end=None):
        end = self.size() if end is None else end
        n = end - head
        ev = np.zeros((n, 3)); U = np.zeros((n, 9)); m = np.zeros((n, 10))
        lib().vxo_read_cache(self._h, head, end, ev, U, m)
        return ev, U, m

    def divide_thread(self, Rp, thd_num=2):
        n = 6 * self.win_size
        H = np.zeros((n, n)); J = np.zeros(n)
        r = lib().vxo_divide_thread(self._h, _c(Rp), thd_num, H, J)
        return H.T.copy(), J, r

    def only_residual(self, Rp, thd_num=2):
        return lib().vxo_only_residual(self._h, _c(Rp), thd_num)

    def damping_iter(self, Rp, max_iter=3, thd_num=2):
        Rp = _c(Rp).copy()
        n = 6 * self.win_size
        hess = np.zeros((n, n)); resis = np.zeros(2); trace = np.zeros((max_iter, 8)); nt = C.c_int(0)
        conv = lib().vxo_damping_iter(self._h, Rp, thd_num, max_iter, hess, resis, trace, C.byref(nt))
        return dict(poses=Rp, hess=hess.T.copy(), resis=resis, trace=trace[: nt.value].copy(), is_converge=bool(conv))

    def time_ba_iteration(self, Rp, thd_num, warmup=1, iters=5):
        th = C.c_double(0); tr = C.c_double(0)
        t = lib().vxo_time_ba_iteration(self._h, _c(Rp), thd_num, warmup, iters, C.byref(th), C.byref(tr))
        return t, th.value, tr.value


def eig_sym3(Cm):
    val = np.zeros(3); vec = np.zeros(9)
    lib().vxo_eig_sym3(_c(np.asarray(Cm).T.reshape(9)), val, vec)
    return val, vec.reshape(3, 3).T.copy()


def exp_so3(a):
    R = np.zeros(9)
    lib().vxo_exp(_c(a), R)
    return R.reshape(3, 3).T.copy()


def ldlt_solve(A, b):
    n = len(b); x = np.zeros(n)
    lib().vxo_ldlt_solve(n, _c(np.asarray(A).T), _c(b), x)
    return x


def cluster_transform(cluster, Rp):
    out = np.zeros(10)
    lib().vxo_cluster_transform(_c(cluster), _c(Rp), out)
    return out


def build_clusters(xyz, cell_ptr):
    cell_ptr = np.ascontiguousarray(cell_ptr, dtype=np.int64)
    out = np.zeros((cell_ptr.shape[0] - 1, 10))
    lib().vxo_build_clusters(cell_ptr.shape[0] - 1, cell_ptr, _c(xyz), out)
    return out


def plane_fit(clusters):
    clusters = _c(clusters)
    n = clusters.shape[0]
    ev = np.zeros((n, 3)); U = np.zeros((n, 9))
    lib().vxo_plane_fit(n, clusters, ev, U)
    return ev, U


# ---- inertial half ------------------------------------------------------------------------------------------
IMU_LEN, STATE_LEN, LI_DIM = 304, 24, 15


def jr(v):
    out = np.zeros(9); lib().vxo_jr(_c(v), out)
    return out.reshape(3, 3).T.copy()


def jr_inv(R):
    out = np.zeros(9); lib().vxo_jr_inv(_c(np.asarray(R).T.reshape(9)), out)
    return out.reshape(3, 3).T.copy()


def mat_inverse(A):
    A = np.asarray(A, dtype=np.float64); n = A.shape[0]
    out = np.zeros(n * n); lib().vxo_mat_inverse(n, _c(A.T.reshape(-1)), out)
    return out.reshape(n, n).T.copy()


def imu_init(bg=None, ba=None):
    blob = np.zeros(IMU_LEN)
    lib().vxo_imu_init(blob, _c(np.zeros(3) if bg is None else bg), _c(np.zeros(3) if ba is None else ba))
    return blob


def imu_add(blob, gyr, acc, dt, noise_meas, noise_walk):
    lib().vxo_imu_add(blob, _c(gyr), _c(acc), float(dt), _c(np.asarray(noise_meas).T), _c(np.asarray(noise_walk).T))


def imu_preintegrate(samples, noise_meas, noise_walk, bg=None, ba=None):
    """(W-1, 304) blobs from synth.make_imu(...).samples through the oracle's add_imu."""
    blobs = []
    for gyr, acc, dts in samples:
        b = imu_init(bg, ba)
        for g, a, dt in zip(gyr, acc, dts):
            imu_add(b, g, a, dt, noise_meas, noise_walk)
        blobs.append(b)
    return np.stack(blobs) if blobs else np.zeros((0, IMU_LEN))


def imu_evaluate(blob, st1, st2, jac_enable=True):
    jtj = np.zeros((30, 30)); gg = np.zeros(30)
    r = lib().vxo_imu_evaluate(_c(blob), _c(st1), _c(st2), jtj, gg, 1 if jac_enable else 0)
    return r, jtj.T.copy(), gg


def li_divide_thread(o, states, blobs, thd_num=5, imu_coef=1e-4):
    n = LI_DIM * o.win_size
    H = np.zeros((n, n)); J = np.zeros(n)
    r = lib().vxo_li_divide_thread(o._h, _c(states), _c(blobs), thd_num, imu_coef, H, J)
    return H.T.copy(), J, r


def li_divide_thread_gravity(o, states, blobs, thd_num=5, imu_coef=1e-4):
    """LI_BA_OptimizerGravity::divide_thread (voxel_map.hpp:673-736): (15W+3)^2 system, gravity unknowns at the tail."""
    n = LI_DIM * o.win_size + 3
    H = np.zeros((n, n)); J = np.zeros(n)
    r = lib().vxo_li_divide_thread_gravity(o._h, _c(states), _c(blobs), thd_num, imu_coef, H, J)
    return H.T.copy(), J, r


def li_only_residual_gravity(o, states, blobs, thd_num=5, imu_coef=1e-4):
    return lib().vxo_li_only_residual_gravity(o._h, _c(states), _c(blobs), thd_num, imu_coef)


def li_only_residual(o, states, blobs, thd_num=5, imu_coef=1e-4):
    return lib().vxo_li_only_residual(o._h, _c(states), _c(blobs), thd_num, imu_coef)


def li_damping_iter(o, states, blobs, max_iter=3, thd_num=5, imu_coef=1e-4):
    st = _c(states).copy(); bl = _c(blobs).copy()
    n = LI_DIM * o.win_size
    hess = np.zeros((n, n)); trace = np.zeros((max(max_iter, 1), 8)); nt = C.c_int(0)
    lib().vxo_li_damping_iter(o._h, st, bl, thd_num, imu_coef, max_iter, hess, trace, C.byref(nt))
    return dict(states=st, imus=bl, hess=hess.T.copy(), trace=trace[: nt.value].copy())


def imu_evaluate_g(blob, st1, st2):
    jtj = np.zeros((33, 33)); gg = np.zeros(33)
    r = lib().vxo_imu_evaluate_g(_c(blob), _c(st1), _c(st2), jtj, gg, 1)
    return r, jtj.T.copy(), gg


def li_damping_iter_gravity(o, states, blobs, max_iter=2, thd_num=5, imu_coef=1e-4):
    st = _c(states).copy(); bl = _c(blobs).copy()
    n = LI_DIM * o.win_size + 3
    hess = np.zeros((n, n)); resis = np.zeros(2); trace = np.zeros((max(max_iter, 1), 8)); nt = C.c_int(0)
    lib().vxo_li_damping_iter_gravity(o._h, st, bl, thd_num, imu_coef, max_iter, hess, resis, trace, C.byref(nt))
    return dict(states=st, imus=bl, hess=hess.T.copy(), resis=resis, trace=trace[: nt.value].copy())


def voxelize(W, xyz_local, frame_ptr, Rp, params9):
    """OctreeGBA::cut_voxel + recut on the CPU: dict(node_id, clusters (n,W,10), eig_val, eig_vec, merged), canonical order."""
    xyz = _c(xyz_local).reshape(-1, 3)
    fp = np.ascontiguousarray(frame_ptr, dtype=np.int64)
    pr = _c(params9)
    floor_pts = int(min([pr[2]] + [v for v in pr[9:13] if v > 0])) if pr.shape[0] >= 13 else int(pr[2])
    cap = xyz.shape[0] // (max(floor_pts, 0) + 1) + 16
    ids = np.zeros(cap, dtype=np.uint64); cl = np.zeros((cap, W, 10)); ev = np.zeros((cap, 3)); U = np.zeros((cap, 9)); m = np.zeros((cap, 10))
    n = lib().vxo_voxelize(W, xyz.shape[0], xyz, fp, _c(Rp), _c(params9), cap, ids, cl, ev, U, m)
    if n < 0:
        raise ValueError("voxel coordinates out of range")
    assert n <= cap
    return dict(node_id=ids[:n].copy(), clusters=cl[:n].copy(), eig_val=ev[:n].copy(), eig_vec=U[:n].copy(), merged=m[:n].copy())


def _unpack_sweep(sw):
    return {"HTH": sw[:36].reshape(6, 6).T.copy(), "HTz": sw[36:42].copy(), "nnt": sw[42:51].reshape(3, 3).T.copy(), "match_num": int(sw[51])}


class LioOracle:
    """lio_state_estimation + match on the CPU (oracle/vxo_lio.hpp), same call shapes as vxba.LioEstimator."""

    def __init__(self, voxel_size=1.0, max_layer=2):
        self._h = C.c_void_p(lib().vxo_lio_create(float(voxel_size), int(max_layer)))

    def __del__(self):
        try:
            if self._h:
                lib().vxo_lio_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def map_update(self, loc, layer, path, center, normal, plane_var, radius, is_plane=None):
        loc = np.ascontiguousarray(loc, dtype=np.int64).reshape(-1, 3)
        n = loc.shape[0]
        pv = np.ascontiguousarray(np.transpose(np.asarray(plane_var, dtype=np.float64).reshape(n, 6, 6), (0, 2, 1)))
        isp = None if is_plane is None else np.ascontiguousarray(is_plane, dtype=np.int32)
        lib().vxo_lio_map_add(self._h, n, loc, np.ascontiguousarray(layer, dtype=np.int32), np.ascontiguousarray(path, dtype=np.int32),
                              None if isp is None else isp.ctypes.data_as(C.c_void_p), _c(center), _c(normal), pv, _c(radius))

    def var_init(self, xyz, ext_R=None, ext_p=None, dept_err=0.02, beam_err=0.05):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        R = np.eye(3) if ext_R is None else np.asarray(ext_R, dtype=np.float64)
        ext = np.concatenate([R.T.reshape(9), np.zeros(3) if ext_p is None else np.asarray(ext_p, dtype=np.float64)])
        lib().vxo_lio_scan_raw(self._h, xyz.shape[0], xyz, ext, float(dept_err), float(beam_err))

    def set_points(self, pnt, var):
        pnt = _c(pnt).reshape(-1, 3)
        var = np.ascontiguousarray(np.transpose(np.asarray(var, dtype=np.float64).reshape(-1, 3, 3), (0, 2, 1)))
        lib().vxo_lio_scan_set(self._h, pnt.shape[0], pnt, var)

    def scan_size(self):
        return int(lib().vxo_lio_scan_size(self._h))

    def read_points(self):
        n = self.scan_size()
        pnt = np.zeros((n, 3)); var = np.zeros((n, 9))
        lib().vxo_lio_scan_read(self._h, pnt, var)
        return pnt, np.transpose(var.reshape(n, 3, 3), (0, 2, 1)).copy()

    @staticmethod
    def _cov(cov):
        return np.ascontiguousarray(np.asarray(cov, dtype=np.float64).reshape(15, 15).T)

    def sweep(self, state, cov, reset_cache=True, want_points=False):
        out = np.zeros(52); n = self.scan_size()
        pop = np.zeros(n, dtype=np.int32); sig = np.zeros(n)
        lib().vxo_lio_sweep(self._h, _c(state), self._cov(cov), 1 if reset_cache else 0, out, pop, sig)
        res = _unpack_sweep(out)
        if want_points:
            res["plane_of_point"] = pop; res["sigma_of_point"] = sig
        return res

    def lio_state_estimation(self, state, cov):
        st = _c(state).copy(); cv = self._cov(cov).copy()
        n = self.scan_size()
        info = np.zeros(4); sweeps = np.zeros((4, 52)); pop = np.zeros(n, dtype=np.int32); sig = np.zeros(n)
        lib().vxo_lio_state_estimation(self._h, st, cv, info, sweeps, pop, sig)
        it = int(info[1])
        return {"ok": bool(info[0]), "state": st, "cov": cv.T.copy(), "iterations": it, "match_num": int(info[2]), "min_eig": float(info[3]),
                "sweeps": [_unpack_sweep(sweeps[k]) for k in range(it)], "plane_of_point": pop, "sigma_of_point": sig}

    def time_state_estimation(self, state, cov, reps=3):
        return float(lib().vxo_time_lio_state_estimation(self._h, _c(state), self._cov(cov), int(reps)))

    def pvec_update(self, state, cov):
        n = self.scan_size()
        pw = np.zeros((n, 3)); var = np.zeros((n, 9))
        lib().vxo_lio_pvec_update(self._h, _c(state), self._cov(cov), pw, var)
        return pw, np.transpose(var.reshape(n, 3, 3), (0, 2, 1)).copy()


def cov_add_build(xyz_world, var, cell_ptr):
    """Sum of Bf_var over the points of each cell (voxel_map.hpp:91-106, 990-992): n_cells x 9 x 9."""
    cp = np.ascontiguousarray(cell_ptr, dtype=np.int64)
    n = cp.shape[0] - 1
    var9 = np.ascontiguousarray(np.transpose(np.asarray(var, dtype=np.float64).reshape(-1, 3, 3), (0, 2, 1)))
    out = np.zeros((n, 81))
    lib().vxo_cov_add_build(n, cp, _c(xyz_world).reshape(-1, 3), var9, out)
    return np.transpose(out.reshape(n, 9, 9), (0, 2, 1)).copy()


def plane_update(clusters, eig_val, eig_vec, cov_add):
    """OctoTree::plane_update (voxel_map.hpp:1118-1146) batched: dict(center, normal, plane_var n x 6 x 6, radius)."""
    cl = _c(clusters).reshape(-1, 10); n = cl.shape[0]
    ca = np.ascontiguousarray(np.transpose(np.asarray(cov_add, dtype=np.float64).reshape(n, 9, 9), (0, 2, 1)))
    center = np.zeros((n, 3)); normal = np.zeros((n, 3)); pv = np.zeros((n, 36)); rad = np.zeros(n)
    lib().vxo_plane_update(n, cl, _c(eig_val).reshape(n, 3), _c(eig_vec).reshape(n, 9), ca, center, normal, pv, rad)
    return dict(center=center, normal=normal, plane_var=np.transpose(pv.reshape(n, 6, 6), (0, 2, 1)).copy(), radius=rad)


def down_sampling_voxel(xyz, voxel_size):
    """down_sampling_voxel (tools.hpp:201-238) on the CPU; output in ascending voxel index."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.zeros_like(xyz)
    n = lib().vxo_down_sampling_voxel(xyz.shape[0], xyz, float(voxel_size), out)
    return out[:n].copy()


class LocalMapOracle:
    """The incremental local map on the CPU (oracle/vxo_octree.hpp): `surf_map` / `surf_map_slide` driven scan by scan the way the
    local-mapping thread drives them (cut_voxel_multi -> multi_recut (+ tras_opt) -> BA -> multi_margi -> ring shift)."""

    def __init__(self, voxel_size=1.0, max_layer=2, min_point=(5, 5, 5, 5), min_eigen_value=0.0025, plane_eigen_value_thre=(0.25, 0.25, 0.25, 0.25),
                 max_points=100, win_size=10, thread_num=5):
        self.win_size = int(win_size)
        prm = np.array([voxel_size, max_layer, *min_point, min_eigen_value, *plane_eigen_value_thre, max_points, win_size, thread_num], dtype=np.float64)
        self._h = C.c_void_p(lib().vxo_localmap_create(prm))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().vxo_localmap_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def cut_voxel(self, ord_, pnt_body, var_world, pwld):
        pnt = _c(pnt_body).reshape(-1, 3)
        var = np.ascontiguousarray(np.transpose(np.asarray(var_world, dtype=np.float64).reshape(-1, 3, 3), (0, 2, 1)))
        lib().vxo_localmap_cut_voxel(self._h, int(ord_), pnt.shape[0], pnt, var, _c(pwld).reshape(-1, 3))

    def recut(self, win_count, poses, factor: "Oracle"):
        """multi_recut: recut + tras_opt into ``factor`` (an ``Oracle``, cleared by the caller like voxhess.clear())."""
        lib().vxo_localmap_recut(self._h, int(win_count), _c(poses)[:win_count], factor._h)

    def margi(self, win_count, poses, factor: "Oracle"):
        if lib().vxo_localmap_margi(self._h, int(win_count), _c(poses)[:win_count], factor._h) != 0:
            raise RuntimeError("margi: opt_state beyond the factor")

    def slide(self, mgsize=1):
        lib().vxo_localmap_slide(self._h, int(mgsize))

    def leaf_points(self, node_id, which, cap=4096):
        """Points a leaf keeps for a later subdivision: which = -1 the fix points (world), which = i the window's i-th scan (body): (n,3), (n,3,3)."""
        buf = np.zeros((cap, 12))
        n = lib().vxo_localmap_leaf_points(self._h, int(node_id), int(which), cap, buf)
        if n < 0:
            raise KeyError(node_id)
        if n > cap:
            return self.leaf_points(node_id, which, cap=int(n))
        return buf[:n, :3].copy(), np.transpose(buf[:n, 3:].reshape(n, 3, 3), (0, 2, 1)).copy()

    def counts(self):
        out = np.zeros(4, dtype=np.int64)
        lib().vxo_localmap_counts(self._h, out)
        return dict(roots=int(out[0]), slide=int(out[1]), leaves=int(out[2]), mp0=int(out[3]))

    def leaves(self):
        W = self.win_size
        n = self.counts()["leaves"]
        ids = np.zeros(n, dtype=np.uint64); ints = np.zeros((n, 8), dtype=np.int32); d = np.zeros((n, 156 + 11 * W))
        got = lib().vxo_localmap_leaves(self._h, n, ids, ints, d)
        if got < 0:
            raise ValueError("root voxel outside the id range")
        assert got == n
        return dict(node_id=ids, layer=ints[:, 0], isexist=ints[:, 1].astype(bool), is_plane=ints[:, 2].astype(bool), has_sw=ints[:, 3].astype(bool),
                    opt_state=ints[:, 4], last_num=ints[:, 5], n_point_fix=ints[:, 6], in_slide=ints[:, 7].astype(bool),
                    pcr_add=d[:, 0:10], pcr_fix=d[:, 10:20], eig_val=d[:, 20:23], eig_vec=d[:, 23:32], center=d[:, 32:35], normal=d[:, 35:38],
                    radius=d[:, 38], plane_var=np.transpose(d[:, 39:75].reshape(n, 6, 6), (0, 2, 1)), cov_add=np.transpose(d[:, 75:156].reshape(n, 9, 9), (0, 2, 1)),
                    pcrs_local=d[:, 156:156 + 10 * W].reshape(n, W, 10), n_points=d[:, 156 + 10 * W:].astype(np.int64))
