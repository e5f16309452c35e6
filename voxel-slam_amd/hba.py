"""One window of the hierarchical / global BA (reference: ``HBA_add_edge``, voxelslam.cpp:2320-2430) on top of the C ABI:
repeat { re-voxelise the window's scans at the current poses (OctreeGBA::cut_voxel + recut -> vxba_voxelize_push), run
Lidar_BA_Optimizer::damping_iter with up to 4 iterations } with the reference's convergence schedule (coarse voxel
parameters first, the odometry's finer ones for the last round), then turn the off-diagonal 6x6 blocks of the final Hessian
into pose-graph edge weights.  Host orchestration only -- the per-round work (hashing, octree, plane tests, sweeps, solve)
runs on the GPU; this file is the harness-level mirror used by the tests and as an integration example (config 5's bottom
level: windows of 10 keyframes on the MFMA path; the top level's ~100 submap poses go through the same calls on the wide-window
path, VXBA_MAX_WIN < W <= VXBA_MAX_WIN_WIDE)."""
from __future__ import annotations

import numpy as np

from . import vxba


def window_refine(xyz_local, frame_ptr, poses, coarse: "vxba.VoxelizeParams", fine: "vxba.VoxelizeParams", max_iter: int = 10, up: int = 4,
                  device: int = 0, factor_cls=None, optimizer=None, voxelize=None, factor=None):
    """Returns dict(poses, hess, rounds, n_voxels, resis).  ``factor_cls`` / ``optimizer`` / ``voxelize`` let the tests run the
    same schedule on the CPU oracle (defaults: the GPU path).  ``factor``: a LidarFactor of the right win_size to reuse (cleared
    before every round) instead of creating one per round -- what a long-running mapper does (the reference constructs a
    LidarFactor per round, voxelslam.cpp:2378; here that would be a dozen device allocations each time)."""
    W = poses.shape[0]
    xs = np.ascontiguousarray(poses, dtype=np.float64).copy()
    converge_flag, converge_thre = 0, 0.05
    hess = None
    log = []
    for it in range(max_iter):
        params = fine if (converge_flag == 1 or it == max_iter - 1) else coarse          # voxelslam.cpp:2362-2372
        if voxelize is None:
            if factor is not None:
                f = factor
                f.clear()
            else:
                f = vxba.LidarFactor(W, device=device) if factor_cls is None else factor_cls(W)
            n_vox = f.voxelize_push(xyz_local, frame_ptr, xs, params, want_ids=False)
        else:
            f, n_vox = voxelize(xyz_local, frame_ptr, xs, params)
        opt = vxba.Lidar_BA_Optimizer() if optimizer is None else optimizer
        if optimizer is None and int(n_vox) == 0:
            # no factor voxel (too few points for any plane): upstream's damping_iter then runs on an all-zero system -- zero step, poses unchanged,
            # *hess zero, residuals 0 and 0 / 0 -- and the schedule goes on; the GPU entry point refuses an empty factor, so that outcome is written here
            out = dict(poses=xs, hess=np.zeros((6 * W, 6 * W)), resis=(0.0, float("nan")), is_converge=False)
        else:
            out = opt.damping_iter(xs, f, max_iter=up)
        xs = out["poses"]
        hess = out["hess"]
        r0, r1 = out["resis"]
        log.append(dict(round=it, n_voxels=int(n_vox), resis=(float(r0), float(r1)), converged=bool(out["is_converge"]), fine=params is fine))
        if factor is None and hasattr(f, "close"):
            f.close()
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.float64(abs(r0 - r1)) / np.float64(r0)          # 0 / 0 for a window without factor voxels: NaN, the test below is false (as upstream)
        if (rel < converge_thre and out["is_converge"]) or (it == max_iter - 2 and converge_flag == 0):   # :2387-2398
            converge_thre = 0.01
            if converge_flag == 0:
                converge_flag = 1
            elif converge_flag == 1:
                break
    return dict(poses=xs, hess=hess, rounds=log)


def edges_from_hessian(poses, hess, min_abs: float = 1e-6):
    """Pose-graph edges of a refined window (voxelslam.cpp:2405-2427): for every frame pair whose six Hessian entries
    hess(6i+k, 6j+k) all exceed ``min_abs`` in magnitude, the relative pose and the weights v6 = 1 / |hess(6i+k, 6j+k)|."""
    W = poses.shape[0]
    R = poses[:, :9].reshape(W, 3, 3).transpose(0, 2, 1)
    p = poses[:, 9:12]
    out = []
    for i in range(W - 1):
        for j in range(i + 1, W):
            hc = np.abs(np.array([hess[6 * i + k, 6 * j + k] for k in range(6)]))
            if np.any(hc < min_abs):
                continue
            out.append(dict(i=i, j=j, rot=R[i].T @ R[j], tra=R[i].T @ (p[j] - p[i]), v6=1.0 / hc))
    return out


def merge_submap(clouds, poses, voxel_size: float, downsample=None, device: int = 0):
    """Tail of ``HBA_add_edge`` (voxelslam.cpp:2430-2450): the window's scans expressed in the FIRST frame's coordinates
    (dR = R0^T Ri, dp = R0^T (pi - p0)), stored as float like ``PointType``, then ``down_sampling_voxel(pl, voxel_size / 8)``."""
    W = poses.shape[0]
    R = poses[:, :9].reshape(W, 3, 3).transpose(0, 2, 1)
    p = poses[:, 9:12]
    parts = []
    for i in range(W):
        dR = R[0].T @ R[i]; dp = R[0].T @ (p[i] - p[0])
        parts.append((np.asarray(clouds[i], dtype=np.float64) @ dR.T + dp).astype(np.float32))
    pl = np.ascontiguousarray(np.concatenate(parts))
    ds = (lambda x, s: vxba.down_sampling_voxel(x, s, device=device)) if downsample is None else downsample
    return ds(pl, voxel_size / 8)


def windows(K: int, wdsize: int, mgsize: int, tail: bool = True):
    """[(first keyframe, keyframe count)] of one bottom-up pass: the numpy-free twin of ``vxba_hba_num_windows`` / ``vxba_hba_window`` (include/vxba.h) --
    full windows of ``wdsize`` keyframes at stride ``mgsize`` (thd_globalmapping runs one whenever localID holds wdsize keyframes and pops mgsize,
    voxelslam.cpp:2536-2541, 2571-2574) and, with ``tail``, the CLOSING window of upstream's last iteration (total_ba == 1, :2519-2523: no size test) over
    the keyframes left behind the last pop, [S mgsize, K)."""
    S = (K - wdsize) // mgsize + 1 if K >= wdsize else 0
    out = [(w * mgsize, wdsize) for w in range(S)]
    if tail and S * mgsize < K:
        out.append((S * mgsize, K - S * mgsize))
    return out


def hierarchical_ba(clouds, poses, coarse: "vxba.VoxelizeParams", fine: "vxba.VoxelizeParams", wdsize: int = 10, mgsize: int = 5, top_max_iter: int = 1,
                    device: int = 0, optimizer=None, voxelize=None, downsample=None, tail: bool = True):
    """Bottom-up pass of the global mapping thread (``thd_globalmapping``, voxelslam.cpp:2485-2595) over one session: the windows of ``windows(K, wdsize,
    mgsize, tail)``, each refined by one round of ``HBA_add_edge`` (max_iter = 1: the odometry's voxel parameters straight away, :2362-2372) and merged
    into a submap anchored at its first keyframe; then one ``HBA_add_edge`` over all submap poses (the top level, up to VXBA_MAX_WIN_WIDE of them) with
    ``top_max_iter`` rounds.  A closing window of ONE keyframe is not refined (its only pose is the gauge; its cloud is the submap).  Returns the
    pose-graph edges of both levels (the GTSAM optimisation that consumes them is outside this library) and the refined submap poses.
    ``clouds``: list of (n_i, 3) arrays in keyframe coordinates; ``poses``: (K, 12).  The keyword hooks run the same schedule on the
    CPU oracle in the tests."""
    K = poses.shape[0]
    sub_clouds, sub_ids, edges1 = [], [], []
    factors = {}                                                  # one factor per window size (the closing window has its own)
    for base, cnt in windows(K, wdsize, mgsize, tail):
        ids = list(range(base, base + cnt))
        if cnt >= 2:
            xyz = np.ascontiguousarray(np.concatenate([np.asarray(clouds[i], dtype=np.float64) for i in ids]))
            fp = np.concatenate([[0], np.cumsum([len(clouds[i]) for i in ids])]).astype(np.int64)
            if voxelize is None and cnt not in factors:
                factors[cnt] = vxba.LidarFactor(cnt, device=device)
            r = window_refine(xyz, fp, poses[ids], coarse, fine, max_iter=1, device=device, optimizer=optimizer, voxelize=None if voxelize is None else voxelize(cnt),
                              factor=factors.get(cnt))
            for e in edges_from_hessian(r["poses"], r["hess"]):
                edges1.append(dict(e, i=ids[e["i"]], j=ids[e["j"]]))
            refined = r["poses"]
        else:
            refined = poses[ids]
        sub_clouds.append(merge_submap([clouds[i] for i in ids], refined, fine.voxel_size, downsample=downsample, device=device))
        sub_ids.append(base)
    for f in factors.values():
        f.close()
    S = len(sub_ids)
    top_xyz = np.ascontiguousarray(np.concatenate(sub_clouds).astype(np.float64))
    top_fp = np.concatenate([[0], np.cumsum([len(c) for c in sub_clouds])]).astype(np.int64)
    if S >= 2:
        top = window_refine(top_xyz, top_fp, poses[sub_ids], coarse, fine, max_iter=top_max_iter, device=device, optimizer=optimizer,
                            voxelize=None if voxelize is None else voxelize(S))
        edges2 = [dict(e, i=sub_ids[e["i"]], j=sub_ids[e["j"]]) for e in edges_from_hessian(top["poses"], top["hess"])]
    else:
        top, edges2 = dict(poses=np.array(poses[sub_ids], dtype=np.float64), rounds=[]), []
    return dict(edges1=edges1, edges2=edges2, submap_ids=sub_ids, submap_poses=top["poses"], submap_sizes=[len(c) for c in sub_clouds], top_rounds=top["rounds"])
