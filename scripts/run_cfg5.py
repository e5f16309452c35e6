"""BASELINE configs[4] on one GPU: hierarchical global BA over a session of K keyframes (default 500) -- bottom level = windows of 10
keyframes with stride 5 (99 of them), top level = one HBA_add_edge over the ~99 submap poses (wide-window path).  Prints wall times per
level and the factor sizes; with --cpu also times the CPU oracle on ONE bottom window and on the top level's two sweeps."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from voxel_slam_amd import hba, synth, vxba

ap = argparse.ArgumentParser()
ap.add_argument("--keyframes", type=int, default=500)
ap.add_argument("--pts", type=int, default=20_000)
ap.add_argument("--cpu", action="store_true")
a = ap.parse_args()
K = a.keyframes
t0 = time.perf_counter()




clouds, poses, gt = synth.corridor_session(K, a.pts, synth.MASTER_SEED + 5000)
print("synthetic session: %d keyframes x %d points (%.1f s to generate)" % (K, a.pts, time.perf_counter() - t0), flush=True)
coarse = vxba.VoxelizeParams(voxel_size=2.0, max_layer=2, min_points=10, min_eigen_value=0.02, eigen_ratio=(1 / 9, 1 / 9, 1 / 9, 1 / 9))
fine = vxba.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))

# instrumented copy of hba.hierarchical_ba's two levels
wd, mg = 10, 5
t_bottom = time.perf_counter()
sub, ids, nvox, e1 = [], [], [], 0
bottom = vxba.LidarFactor(wd)
t_parts = np.zeros(3)
for base in range(0, K - wd + 1, mg):
    w = list(range(base, base + wd))
    x = np.ascontiguousarray(np.concatenate([clouds[i].astype(np.float64) for i in w]))
    f = np.concatenate([[0], np.cumsum([len(clouds[i]) for i in w])]).astype(np.int64)
    ta = time.perf_counter()
    r = hba.window_refine(x, f, poses[w], coarse, fine, max_iter=1, factor=bottom)
    tb = time.perf_counter()
    e1 += len(hba.edges_from_hessian(r["poses"], r["hess"]))
    nvox.append(r["rounds"][-1]["n_voxels"])
    tc = time.perf_counter()
    sub.append(hba.merge_submap([clouds[i] for i in w], r["poses"], fine.voxel_size)); ids.append(base)
    t_parts += [tb - ta, tc - tb, time.perf_counter() - tc]
t_bottom = time.perf_counter() - t_bottom
bottom.close()
S = len(ids)
top_xyz = np.ascontiguousarray(np.concatenate(sub).astype(np.float64)); top_fp = np.concatenate([[0], np.cumsum([len(c) for c in sub])]).astype(np.int64)
t_top = time.perf_counter()
top = hba.window_refine(top_xyz, top_fp, poses[ids], coarse, fine, max_iter=2)
t_top = time.perf_counter() - t_top
e2 = len(hba.edges_from_hessian(top["poses"], top["hess"]))
idx = np.array(ids)
err0 = synth.pose_errors(poses[idx], gt[idx]); err1 = synth.pose_errors(top["poses"], gt[idx])
out = dict(keyframes=K, points_per_keyframe=a.pts, bottom_windows=S, bottom_s=t_bottom, bottom_split_s=dict(refine=t_parts[0], edges=t_parts[1], merge_downsample=t_parts[2]), bottom_factor_voxels_mean=float(np.mean(nvox)), bottom_factor_voxels_total=int(np.sum(nvox)),
           submap_points_mean=float(np.mean([len(c) for c in sub])), top_W=S, top_s=t_top, top_points=int(top_xyz.shape[0]),
           top_rounds=[dict(n_voxels=r["n_voxels"], resis=r["resis"]) for r in top["rounds"]], top_packed_bytes=8 * (36 * S * S + 6 * S + 1), edges=[e1, e2],
           anchor_error_before_m_rad=list(err0), anchor_error_after_m_rad=list(err1))
if a.cpu:
    from tests import _oracle as O
    w = list(range(0, wd))
    x = np.ascontiguousarray(np.concatenate([clouds[i].astype(np.float64) for i in w])); f = np.concatenate([[0], np.cumsum([len(clouds[i]) for i in w])]).astype(np.int64)
    t = time.perf_counter(); r = O.voxelize(wd, x, f, poses[w], fine.as_array()); fo = O.Oracle(wd); n = r["node_id"].size
    fo.push_voxels(r["clusters"], np.zeros((n, 10)), np.ones(n), r["eig_val"], r["eig_vec"], r["merged"]); fo.damping_iter(poses[w], max_iter=4, thd_num=5)
    out["cpu_oracle_one_bottom_window_s"] = time.perf_counter() - t
    t = time.perf_counter(); r = O.voxelize(S, top_xyz, top_fp, poses[ids], fine.as_array()); out["cpu_oracle_top_voxelize_s"] = time.perf_counter() - t
    fo = O.Oracle(S); n = r["node_id"].size
    fo.push_voxels(r["clusters"], np.zeros((n, 10)), np.ones(n), r["eig_val"], r["eig_vec"], r["merged"])
    t = time.perf_counter(); fo.acc_evaluate2(poses[ids]); out["cpu_oracle_top_hessian_sweep_s"] = time.perf_counter() - t
print(json.dumps(out), flush=True)
