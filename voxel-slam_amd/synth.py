"""Deterministic synthetic sliding-window scenes for the local-BA hot path.

Input synthesis only (no BA arithmetic lives here): a voxel lattice of planar
patches observed from W poses, emitted in the formats the boundary consumes --
body-frame points bucketed per (frame, voxel) cell for K1, or pre-built
per-(voxel, frame) clusters for ``push_voxels``.  Follows the recipe of
SURVEY.md section 8(d): voxel_size 1 m (reference voxelslam.cpp:795), range noise
sigma 0.02 m along the normal (``dept_err``, voxelslam.cpp:793), three near-
orthogonal normal families so the window is fully constrained, trajectory
p_i = (0.5 i, 0.1 sin i, 0), R_i = Exp(0.02 i a), frame 0 exact (gauge).

Packed formats (same as include/vxba.h):
  cluster : 10 f64 [Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz N]
  pose    : 12 f64 [R col-major (9) | p (3)]
"""
from __future__ import annotations

import dataclasses

import numpy as np

MASTER_SEED = 20250410

# name -> (W, pts/scan, voxels); BASELINE.json configs[0..3]
CONFIGS = {
    "cfg1": dict(win_size=5, pts_per_scan=20_000, n_voxels=5_000),
    "cfg2": dict(win_size=10, pts_per_scan=100_000, n_voxels=50_000),
    "cfg3": dict(win_size=10, pts_per_scan=200_000, n_voxels=100_000),
    "cfg4": dict(win_size=10, pts_per_scan=1_000_000, n_voxels=400_000),
    # SURVEY 8d's secondary runs of cfg2: half the (voxel, frame) incidences missing; 30 % of the voxels with a world-frame fix cluster
    "cfg2_sparse": dict(win_size=10, pts_per_scan=100_000, n_voxels=50_000, p_obs=0.5),
    "cfg2_fix": dict(win_size=10, pts_per_scan=100_000, n_voxels=50_000, fix_frac=0.3),
    # development: cfg2 with a voxel count that fills whole steps of the Hessian sweep on 256 CUs (256 x 4 x 48 voxels: no ragged step) -- what the ragged step costs
    "cfg2_even": dict(win_size=10, pts_per_scan=100_000, n_voxels=49_152),
}


def rodrigues(ang: np.ndarray) -> np.ndarray:
    """Exp map, same 1e-11 cut-off as the reference (tools.hpp:51-66)."""
    ang = np.asarray(ang, dtype=np.float64)
    th = np.linalg.norm(ang)
    if th < 1e-11:
        return np.eye(3)
    k = ang / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def pack_poses(Rs: np.ndarray, ps: np.ndarray) -> np.ndarray:
    W = Rs.shape[0]
    out = np.empty((W, 12))
    out[:, :9] = np.transpose(Rs, (0, 2, 1)).reshape(W, 9)  # column-major
    out[:, 9:] = ps
    return out


def unpack_poses(Rp: np.ndarray):
    Rp = np.asarray(Rp, dtype=np.float64).reshape(-1, 12)
    Rs = np.transpose(Rp[:, :9].reshape(-1, 3, 3), (0, 2, 1))
    return Rs.copy(), Rp[:, 9:].copy()


def clusters_from_points(xyz: np.ndarray, cell_ptr: np.ndarray) -> np.ndarray:
    """numpy restatement of PointCluster::push over bucketed points (independent of the oracle)."""
    n_cells = cell_ptr.shape[0] - 1
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    feats = np.stack([x * x, x * y, x * z, y * y, y * z, z * z, x, y, z, np.ones_like(x)], axis=1)
    cs = np.zeros((xyz.shape[0] + 1, 10))
    np.cumsum(feats, axis=0, out=cs[1:])
    # cumulative sums lose a few ulps; fine for an input generator (the K1 parity tests use their own sums)
    return cs[cell_ptr[1:]] - cs[cell_ptr[:-1]] if n_cells else np.zeros((0, 10))


@dataclasses.dataclass
class Scene:
    win_size: int
    n_voxels: int
    points_body: np.ndarray   # (Npts, 3) f64, sorted by cell = frame * V + voxel
    cell_ptr: np.ndarray      # (W*V + 1,) int64
    clusters: np.ndarray      # (V, W, 10) f64 body-frame clusters (N = 0 -> unobserved)
    fix: np.ndarray           # (V, 10) world-frame fix clusters (N = 0 -> none)
    coe: np.ndarray           # (V,)
    poses_gt: np.ndarray      # (W, 12)
    poses_init: np.ndarray    # (W, 12) perturbed initial guess, frame 0 exact
    normals: np.ndarray       # (V, 3)

    @property
    def nnz(self) -> int:
        return int(np.count_nonzero(self.clusters[:, :, 9]))


def make_scene(win_size=5, pts_per_scan=20_000, n_voxels=5_000, p_obs=1.0, fix_frac=0.0, noise=0.02,
               rot_sigma_deg=0.05, trans_sigma=0.02, seed=MASTER_SEED, exact_clusters=True, pose_seed=None) -> Scene:
    """Build one window.  ``exact_clusters`` sums clusters per cell with np.add.reduceat
    (sequential per-cell order) instead of the cumulative-sum difference.  ``pose_seed`` draws the initial-guess
    perturbation from its own stream so voxel shards generated with different ``seed`` share one window."""
    rng = np.random.Generator(np.random.PCG64(seed))
    prng = np.random.Generator(np.random.PCG64([seed if pose_seed is None else pose_seed, 7919]))
    W, V = win_size, n_voxels

    # occupied cells in a slab around the trajectory (surfaces: ground/walls within a few metres of height)
    L = max(4, int(np.ceil(np.sqrt(V / (8 * 0.3)))))
    n_lattice = L * L * 8
    flat = rng.choice(n_lattice, size=V, replace=False)
    flat.sort()
    ix, iy, iz = flat // (L * 8), (flat // 8) % L, flat % 8
    centres = np.stack([ix - L / 2 + 0.5, iy - L / 2 + 0.5, iz - 2 + 0.5], axis=1).astype(np.float64)

    # normals: +-x / +-y / +-z families tilted by <= 10 degrees
    fam = rng.integers(0, 3, size=V)
    sign = rng.choice([-1.0, 1.0], size=V)
    base = np.zeros((V, 3))
    base[np.arange(V), fam] = sign
    tilt = np.deg2rad(10.0) * rng.uniform(0, 1, size=V)
    az = rng.uniform(0, 2 * np.pi, size=V)
    t1 = np.zeros((V, 3)); t1[np.arange(V), (fam + 1) % 3] = 1.0
    t2 = np.zeros((V, 3)); t2[np.arange(V), (fam + 2) % 3] = 1.0
    normals = np.cos(tilt)[:, None] * base + np.sin(tilt)[:, None] * (np.cos(az)[:, None] * t1 + np.sin(az)[:, None] * t2)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    b1 = np.cross(normals, t1); b1 /= np.linalg.norm(b1, axis=1, keepdims=True)
    b2 = np.cross(normals, b1)
    origin = centres + rng.uniform(-0.3, 0.3, size=V)[:, None] * normals

    # trajectory
    axis = np.array([0.2, 0.1, 1.0]); axis /= np.linalg.norm(axis)
    Rs = np.stack([rodrigues(0.02 * i * axis) for i in range(W)])
    ps = np.stack([np.array([0.5 * i, 0.1 * np.sin(i), 0.0]) for i in range(W)])
    Rs_init, ps_init = Rs.copy(), ps.copy()
    for i in range(1, W):
        Rs_init[i] = Rs[i] @ rodrigues(prng.normal(0, np.deg2rad(rot_sigma_deg), size=3))
        ps_init[i] = ps[i] + prng.normal(0, trans_sigma, size=3)

    # per-frame observation pattern and point counts: every observed cell gets >= 1 point, and every voxel is
    # seen from >= min(2, W) frames (the reference drops voxels with too few points / observers:
    # voxel_map.hpp:1155, loop_refine.hpp:371-376)
    if p_obs < 1.0:
        seen_mask = rng.uniform(size=(W, V)) < p_obs
        need = min(2, W)
        for a in np.nonzero(seen_mask.sum(axis=0) < need)[0]:
            missing = np.nonzero(~seen_mask[:, a])[0]
            seen_mask[rng.choice(missing, size=need - int(seen_mask[:, a].sum()), replace=False), a] = True
    else:
        seen_mask = np.ones((W, V), dtype=bool)
    counts = np.zeros((W, V), dtype=np.int64)
    for i in range(W):
        seen = np.nonzero(seen_mask[i])[0]
        if seen.size == 0:
            continue
        extra = max(0, pts_per_scan - seen.size)
        counts[i, seen] = 1 + rng.multinomial(extra, np.full(seen.size, 1.0 / seen.size))
    cell_ptr = np.zeros(W * V + 1, dtype=np.int64)
    np.cumsum(counts.reshape(-1), out=cell_ptr[1:])
    npts = int(cell_ptr[-1])
    cell_of_pt = np.repeat(np.arange(W * V, dtype=np.int64), counts.reshape(-1))
    vox_of_pt = cell_of_pt % V
    frm_of_pt = cell_of_pt // V

    s1 = rng.uniform(-0.45, 0.45, size=npts)
    s2 = rng.uniform(-0.45, 0.45, size=npts)
    nz = rng.normal(0, noise, size=npts)
    world = origin[vox_of_pt] + s1[:, None] * b1[vox_of_pt] + s2[:, None] * b2[vox_of_pt] + nz[:, None] * normals[vox_of_pt]
    # body frame x = R_gt^T (w - p_gt)
    d = world - ps[frm_of_pt]
    points_body = np.einsum("nji,nj->ni", Rs[frm_of_pt], d)
    points_body = np.ascontiguousarray(points_body)

    if exact_clusters and npts:
        x, y, z = points_body[:, 0], points_body[:, 1], points_body[:, 2]
        feats = np.stack([x * x, x * y, x * z, y * y, y * z, z * z, x, y, z, np.ones_like(x)], axis=1)
        nonempty = counts.reshape(-1) > 0
        cl = np.zeros((W * V, 10))
        cl[nonempty] = np.add.reduceat(feats, cell_ptr[:-1][nonempty], axis=0)
    else:
        cl = clusters_from_points(points_body, cell_ptr)
    clusters = np.ascontiguousarray(cl.reshape(W, V, 10).transpose(1, 0, 2))

    # optional world-frame fix clusters (marginalised scans, voxel_map.hpp:1256-1268)
    fix = np.zeros((V, 10))
    if fix_frac > 0:
        has = np.nonzero(rng.uniform(size=V) < fix_frac)[0]
        nfix = rng.integers(20, 101, size=has.size)
        for a, k in zip(has, nfix):
            q1 = rng.uniform(-0.45, 0.45, size=k); q2 = rng.uniform(-0.45, 0.45, size=k); qn = rng.normal(0, noise, size=k)
            w = origin[a] + q1[:, None] * b1[a] + q2[:, None] * b2[a] + qn[:, None] * normals[a]
            fix[a] = [np.sum(w[:, 0] ** 2), np.sum(w[:, 0] * w[:, 1]), np.sum(w[:, 0] * w[:, 2]), np.sum(w[:, 1] ** 2),
                      np.sum(w[:, 1] * w[:, 2]), np.sum(w[:, 2] ** 2), w[:, 0].sum(), w[:, 1].sum(), w[:, 2].sum(), k]

    return Scene(win_size=W, n_voxels=V, points_body=points_body, cell_ptr=cell_ptr, clusters=clusters, fix=fix,
                 coe=np.ones(V), poses_gt=pack_poses(Rs, ps), poses_init=pack_poses(Rs_init, ps_init), normals=normals)


def make_config(name: str, **overrides) -> Scene:
    idx = list(CONFIGS).index(name) + 1
    kw = dict(CONFIGS[name]); kw.setdefault("seed", MASTER_SEED + idx); kw.update(overrides)
    return make_scene(**kw)


def pose_errors(Rp_a: np.ndarray, Rp_b: np.ndarray):
    """(translation RMSE [m], rotation RMSE [rad]) over the window between two packed pose sets."""
    Ra, pa = unpack_poses(Rp_a)
    Rb, pb = unpack_poses(Rp_b)
    dt = np.linalg.norm(pa - pb, axis=1)
    dr = []
    for A, B in zip(Ra, Rb):
        c = np.clip((np.trace(A.T @ B) - 1) / 2, -1, 1)
        # for tiny angles use the skew part (acos loses precision near 1)
        S = A.T @ B
        s = 0.5 * np.linalg.norm([S[2, 1] - S[1, 2], S[0, 2] - S[2, 0], S[1, 0] - S[0, 1]])
        dr.append(np.arctan2(s, c))
    return float(np.sqrt(np.mean(dt ** 2))), float(np.sqrt(np.mean(np.square(dr))))


# ------------------------------------------------------------------------------------------------------------
# Inertial side of a window: a smooth trajectory through the scene's ground-truth poses, an IMU sample stream
# measured along it, and the 15-dimensional window states (R, p, v, bg, ba | g) the LiDAR-inertial BA optimises.
# ------------------------------------------------------------------------------------------------------------
GRAVITY = np.array([0.0, 0.0, -9.8])          # G_m_s2 of the reference (tools.hpp), world frame
STATE_LEN = 24


def so3_log(R: np.ndarray) -> np.ndarray:
    """Rotation vector of R (angle < pi); atan2 form, accurate for tiny angles too."""
    K = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    sn = np.linalg.norm(K)
    th = np.arctan2(sn, 0.5 * (np.trace(R) - 1))
    return K if sn < 1e-12 else (th / sn) * K


def pack_states(Rs, ps, vs, bg, ba, g=GRAVITY) -> np.ndarray:
    W = len(Rs)
    out = np.zeros((W, STATE_LEN))
    for i in range(W):
        out[i, :9] = np.asarray(Rs[i]).T.reshape(9)
        out[i, 9:12] = ps[i]; out[i, 12:15] = vs[i]; out[i, 15:18] = bg; out[i, 18:21] = ba; out[i, 21:24] = g
    return out


@dataclasses.dataclass
class ImuWindow:
    states_gt: np.ndarray      # (W, 24)
    states_init: np.ndarray    # (W, 24): scene.poses_init + perturbed velocities + the bias estimate
    samples: list              # W-1 entries of (gyr (K,3), acc (K,3), dt (K,)): mid-point, bias-estimate-corrected samples
    noise_meas: np.ndarray     # (6,6)
    noise_walk: np.ndarray     # (6,6)
    dt_frame: float


def make_imu(scene: Scene, rate_hz=200.0, frame_dt=0.1, gyr_sigma=1e-3, acc_sigma=1e-2, bias_g=(0.002, -0.001, 0.0015),
             bias_a=(0.02, 0.01, -0.015), bias_est_err=0.3, vel_sigma=0.02, cov_gyr=0.01, cov_acc=1.0, rdw_gyr=1e-4, rdw_acc=1e-4,
             seed=MASTER_SEED + 77) -> ImuWindow:
    """IMU stream consistent with ``scene.poses_gt``: between frames the body turns with a constant body rate and moves
    with a constant world acceleration (so frame poses are hit exactly); the accelerometer reads R^T (a - g) + ba + noise,
    the gyro w + bg + noise.  The bias *estimate* used for preintegration is off by ``bias_est_err`` (relative), which is
    what the BA's bias states have to absorb.  Noise densities default to the reference's LocalBA settings
    (config/avia.yaml:39-42)."""
    rng = np.random.Generator(np.random.PCG64([seed, 4242]))
    W = scene.win_size
    Rs, ps = unpack_poses(scene.poses_gt)
    K = int(round(rate_hz * frame_dt))
    h = frame_dt / K
    bg_true = np.asarray(bias_g, dtype=np.float64); ba_true = np.asarray(bias_a, dtype=np.float64)
    bg_est = bg_true * (1 - bias_est_err); ba_est = ba_true * (1 - bias_est_err)
    # velocities at the frames: v_{i+1} = 2 (p_{i+1} - p_i) / T - v_i  (constant acceleration per interval)
    vs = np.zeros((W, 3))
    vs[0] = (ps[1] - ps[0]) / frame_dt if W > 1 else 0.0
    samples = []
    for i in range(W - 1):
        a_w = 2 * (ps[i + 1] - ps[i] - vs[i] * frame_dt) / frame_dt ** 2
        vs[i + 1] = vs[i] + a_w * frame_dt
        w_b = so3_log(Rs[i].T @ Rs[i + 1]) / frame_dt
        t = np.arange(K + 1) * h
        gyr_raw = np.zeros((K + 1, 3)); acc_raw = np.zeros((K + 1, 3))
        for k in range(K + 1):
            Rt = Rs[i] @ rodrigues(w_b * t[k])
            gyr_raw[k] = w_b + bg_true + rng.normal(0, gyr_sigma, 3)
            acc_raw[k] = Rt.T @ (a_w - GRAVITY) + ba_true + rng.normal(0, acc_sigma, 3)
        gyr = 0.5 * (gyr_raw[:-1] + gyr_raw[1:]) - bg_est        # push_imu: mid-point, minus the factor's bias (preintegration.hpp:58-69)
        acc = 0.5 * (acc_raw[:-1] + acc_raw[1:]) - ba_est
        samples.append((gyr, acc, np.full(K, h)))
    Ri, pi = unpack_poses(scene.poses_init)
    vi = vs + rng.normal(0, vel_sigma, size=vs.shape)
    vi[0] = vs[0]
    noise_meas = np.diag([cov_gyr] * 3 + [cov_acc] * 3).astype(np.float64)
    noise_walk = np.diag([rdw_gyr] * 3 + [rdw_acc] * 3).astype(np.float64)
    return ImuWindow(states_gt=pack_states(Rs, ps, vs, bg_true, ba_true), states_init=pack_states(Ri, pi, vi, bg_est, ba_est),
                     samples=samples, noise_meas=noise_meas, noise_walk=noise_walk, dt_frame=frame_dt)


# ------------------------------------------------------------------------------------------------------------
# Raw scans for the batch factor construction (voxel hash -> octree -> plane test): a few large planar walls / floors
# plus clutter, sampled by every frame, NOT aligned with the voxel grid -- so root voxels contain one plane, two planes
# (edges: subdivided), or clutter (rejected), like a real scene.
# ------------------------------------------------------------------------------------------------------------
def make_scans(win_size=5, pts_per_scan=20_000, extent=20.0, noise=0.01, clutter_frac=0.1, seed=MASTER_SEED + 555, rot_sigma_deg=0.0,
               trans_sigma=0.0):
    """Returns (xyz_local (N,3) frame-major, frame_ptr (W+1,), poses (W,12)).  Points are expressed in their frame's body
    coordinates with the returned poses (optionally perturbed away from the poses the scans were taken at)."""
    rng = np.random.Generator(np.random.PCG64([seed, 99]))
    W = win_size
    axis = np.array([0.2, 0.1, 1.0]); axis /= np.linalg.norm(axis)
    Rs = np.stack([rodrigues(0.03 * i * axis) for i in range(W)])
    ps = np.stack([np.array([0.4 * i, 0.1 * np.sin(i), 0.02 * i]) for i in range(W)])
    # planes: point + two in-plane unit vectors + size
    planes = []
    for k in range(8):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        if k < 3:
            n = np.eye(3)[k] + 0.05 * rng.normal(size=3); n /= np.linalg.norm(n)
        a = np.cross(n, [0.3, 0.5, 0.8]); a /= np.linalg.norm(a)
        b = np.cross(n, a)
        c = rng.uniform(-extent / 2, extent / 2, size=3)
        planes.append((c, a, b, n, rng.uniform(extent / 4, extent / 2)))
    clouds = []
    for i in range(W):
        n_cl = int(pts_per_scan * clutter_frac)
        n_pl = pts_per_scan - n_cl
        which = rng.integers(0, len(planes), size=n_pl)
        w = np.zeros((n_pl, 3))
        for k, (c, a, b, n, sz) in enumerate(planes):
            sel = np.nonzero(which == k)[0]
            # a fixed lattice of surface samples per plane, so every frame revisits the same surface patches
            u = rng.uniform(-sz, sz, size=sel.size); v = rng.uniform(-sz, sz, size=sel.size)
            w[sel] = c + u[:, None] * a + v[:, None] * b + rng.normal(0, noise, size=sel.size)[:, None] * n
        cl = rng.uniform(-extent / 2, extent / 2, size=(n_cl, 3))
        world = np.concatenate([w, cl])[rng.permutation(pts_per_scan)]
        clouds.append(np.einsum("ji,nj->ni", Rs[i], world - ps[i]))
    xyz = np.ascontiguousarray(np.concatenate(clouds))
    frame_ptr = np.arange(W + 1, dtype=np.int64) * pts_per_scan
    Rs2, ps2 = Rs.copy(), ps.copy()
    for i in range(1, W):
        if rot_sigma_deg > 0:
            Rs2[i] = Rs[i] @ rodrigues(rng.normal(0, np.deg2rad(rot_sigma_deg), size=3))
        if trans_sigma > 0:
            ps2[i] = ps[i] + rng.normal(0, trans_sigma, size=3)
    return xyz, frame_ptr, pack_poses(Rs2, ps2), pack_poses(Rs, ps)


# ------------------------------------------------------------------------------------------------------------
# Odometry: a voxel plane map as the reference's `surf_map` would hold it after some scans -- root voxels that
# are one plane, or are subdivided once / twice with planes, plane-less leaves and missing children below -- flattened to its
# leaves, and one scan that sees those planes from a pose near the true one.
# ------------------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class PlaneMapData:
    voxel_size: float
    max_layer: int
    loc: np.ndarray        # (n,3) int64  root voxel
    layer: np.ndarray      # (n,)  int32
    path: np.ndarray       # (n,)  int32  leafnum per level, 3 bits each, first level lowest
    is_plane: np.ndarray   # (n,)  int32
    center: np.ndarray     # (n,3)
    normal: np.ndarray     # (n,3)
    plane_var: np.ndarray  # (n,6,6)
    radius: np.ndarray     # (n,)  float32-representable
    box_center: np.ndarray  # (n,3) centre of the leaf's cell
    box_half: np.ndarray   # (n,)

    def args(self):
        return self.loc, self.layer, self.path, self.center, self.normal, self.plane_var, self.radius, self.is_plane


def make_plane_map(n_roots=2000, extent=12, voxel_size=1.0, max_layer=2, seed=MASTER_SEED + 900) -> PlaneMapData:
    rng = np.random.Generator(np.random.PCG64([seed, 31]))
    side = 2 * extent
    flat = rng.choice(side ** 3, size=n_roots, replace=False)
    roots = np.stack([flat // (side * side), (flat // side) % side, flat % side], axis=1).astype(np.int64) - extent
    rows = []

    def leaf(loc, layer, path, c_box, half, plane):
        if plane:
            fam = np.eye(3)[rng.integers(0, 3)] * rng.choice([-1.0, 1.0])
            n = fam + 0.15 * rng.normal(size=3); n /= np.linalg.norm(n)
            c = c_box + rng.uniform(-0.3, 0.3, size=3) * half
            B = rng.normal(size=(6, 6)) * 3e-3
            rows.append((loc, layer, path, 1, c, n, B @ B.T, np.float32((2 * half) ** 2 / 12 * rng.uniform(0.6, 1.2)), c_box, half))
        else:
            rows.append((loc, layer, path, 0, c_box, np.array([0.0, 0.0, 1.0]), np.zeros((6, 6)), np.float32(0.0), c_box, half))

    def grow(loc, layer, path, c_box, half):
        # a node at `layer`: leaf (plane or not) or subdivided
        p_split = 0.0 if layer >= max_layer else (0.4 if layer == 0 else 0.3)
        if rng.random() < p_split:
            for leafnum in range(8):
                if rng.random() < 0.65:
                    sgn = np.array([(leafnum >> 2) & 1, (leafnum >> 1) & 1, leafnum & 1]) * 2 - 1
                    grow(loc, layer + 1, path | (leafnum << (3 * layer)), c_box + sgn * half / 2, half / 2)
        else:
            leaf(loc, layer, path, c_box, half, rng.random() < 0.9)

    for loc in roots:
        grow(loc, 0, 0, (loc + 0.5) * voxel_size, voxel_size / 2)
    cols = list(zip(*rows))
    return PlaneMapData(voxel_size, max_layer, np.array(cols[0], dtype=np.int64), np.array(cols[1], dtype=np.int32), np.array(cols[2], dtype=np.int32),
                        np.array(cols[3], dtype=np.int32), np.array(cols[4]), np.array(cols[5]), np.array(cols[6]), np.array(cols[7], dtype=np.float64),
                        np.array(cols[8]), np.array(cols[9]))


@dataclasses.dataclass
class LioScan:
    xyz: np.ndarray        # (n,3) float32, sensor (= IMU) frame
    state_gt: np.ndarray   # VXBA state vector of the pose the scan was taken at
    state_init: np.ndarray  # the propagated guess lio_state_estimation starts from
    cov: np.ndarray        # 15x15


def make_lio_scan(pm: PlaneMapData, n_points=20_000, noise=0.02, clutter_frac=0.05, rot_sigma_deg=0.3, trans_sigma=0.03, seed=MASTER_SEED + 901,
                  coherent=False, planes_hit=None) -> LioScan:
    """coherent: points ordered along the voxel grid (a spinning LiDAR's neighbouring returns fall on neighbouring surfaces) instead of
    shuffled; planes_hit: restrict the scan to that many of the map's planes (a real scan sees a small part of the map, many points each)."""
    rng = np.random.Generator(np.random.PCG64([seed, 32]))
    R_gt = rodrigues(np.array([0.02, -0.03, 0.4])); p_gt = np.array([0.3, -0.2, 0.1])
    planes = np.nonzero(pm.is_plane == 1)[0]
    if planes_hit is not None and planes_hit < planes.size:
        planes = planes[rng.choice(planes.size, size=planes_hit, replace=False)]
    n_cl = int(n_points * clutter_frac); n_pl = n_points - n_cl
    k = planes[rng.integers(0, planes.size, size=n_pl)]
    n = pm.normal[k]
    a = np.cross(n, np.array([0.31, 0.52, 0.79])); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.cross(n, a)
    h = pm.box_half[k][:, None]
    w = pm.center[k] + rng.uniform(-1, 1, size=(n_pl, 1)) * h * a + rng.uniform(-1, 1, size=(n_pl, 1)) * h * b + rng.normal(0, noise, size=(n_pl, 1)) * n
    lo = pm.loc.min(axis=0) * pm.voxel_size; hi = (pm.loc.max(axis=0) + 1) * pm.voxel_size
    cl = rng.uniform(lo, hi, size=(n_cl, 3))
    world = np.concatenate([w, cl])[rng.permutation(n_points)]
    if coherent:
        cell = np.floor(world / pm.voxel_size).astype(np.int64)
        world = world[np.lexsort((cell[:, 2], cell[:, 1], cell[:, 0]))]
    xyz = np.ascontiguousarray(((world - p_gt) @ R_gt).astype(np.float32))
    R0 = R_gt @ rodrigues(rng.normal(0, np.deg2rad(rot_sigma_deg), size=3)); p0 = p_gt + rng.normal(0, trans_sigma, size=3)
    v = np.array([0.5, 0.1, 0.0]); g = np.array([0.0, 0.0, -9.8])
    st = lambda R, p: np.concatenate([R.T.reshape(9), p, v, np.zeros(3), np.zeros(3), g])
    cov = np.eye(15) * 1e-4
    cov[9:, 9:] = np.eye(6) * 1e-5
    M = rng.normal(size=(15, 15)) * 1e-3       # a propagated covariance is not diagonal
    cov = cov + 0.02 * (M @ M.T)
    return LioScan(xyz, st(R_gt, p_gt), st(R0, p0), cov)


# BASELINE configs[4]: a long session for the hierarchical global BA (scripts/run_cfg5.py, bench.py --config cfg5)
def corridor_session(K, pts, seed):
    """A long hall (floor, ceiling, two side walls, all slightly tilted against the voxel grid) with partial cross walls every 10 m,
    seen by a sensor moving along it with a 30 m range: every keyframe samples only the surfaces around it."""
    rng = np.random.Generator(np.random.PCG64([seed, 7]))
    axis = np.array([0.2, 0.1, 1.0]); axis /= np.linalg.norm(axis)
    Rs = np.stack([rodrigues(0.004 * i * axis) for i in range(K)])
    ps = np.stack([np.array([0.4 * i, 0.6 * np.sin(0.05 * i), 0.1 * np.sin(0.03 * i)]) for i in range(K)])
    tilt = rodrigues(np.array([0.013, -0.021, 0.017]))
    clouds = []
    for i in range(K):
        x0 = ps[i, 0]
        n_each = pts // 5
        u = rng.uniform(x0 - 30, x0 + 30, size=(5, n_each)); v = rng.uniform(0, 1, size=(5, n_each))
        floor = np.stack([u[0], -8 + 16 * v[0], np.full(n_each, -2.0)], 1)
        ceil_ = np.stack([u[1], -8 + 16 * v[1], np.full(n_each, 4.0)], 1)
        wall1 = np.stack([u[2], np.full(n_each, -8.0), -2 + 6 * v[2]], 1)
        wall2 = np.stack([u[3], np.full(n_each, 8.0), -2 + 6 * v[3]], 1)
        kx = np.round(u[4] / 10.0) * 10.0                                   # cross walls at x = 10 k, alternating sides, 4 m wide
        side = np.where((kx / 10.0) % 2 == 0, 1.0, -1.0)
        cross = np.stack([kx, side * (4 + 4 * v[4]), -2 + 6 * rng.uniform(0, 1, n_each)], 1)
        w = np.concatenate([floor, ceil_, wall1, wall2, cross]) @ tilt.T
        w += rng.normal(0, 0.01, size=w.shape)
        clouds.append(((w - ps[i]) @ Rs[i]).astype(np.float32))
    gt = pack_poses(Rs, ps)
    Ri, pi = Rs.copy(), ps.copy()
    for i in range(1, K):
        Ri[i] = Rs[i] @ rodrigues(rng.normal(0, np.deg2rad(0.05), size=3)); pi[i] = ps[i] + rng.normal(0, 0.02, size=3)
    return clouds, pack_poses(Ri, pi), gt
