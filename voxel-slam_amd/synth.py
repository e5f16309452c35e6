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
