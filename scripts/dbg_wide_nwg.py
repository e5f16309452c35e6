import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/voxel-slam_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from voxel_slam_amd import synth, vxba
W, V = 99, 100_000
sc = synth.make_scene(win_size=W, pts_per_scan=60_000, n_voxels=V, p_obs=0.05, seed=5)
obs = sc.clusters[:, :, 9] != 0
row_ptr = np.concatenate([[0], np.cumsum(obs.sum(axis=1))]).astype(np.int64)
vv, fr = np.nonzero(obs)
ecl = np.ascontiguousarray(sc.clusters[vv, fr])
f = vxba.LidarFactor(W)
f.push_voxels_csr(row_ptr, fr.astype(np.int32), ecl, sc.fix, sc.coe)
f.evaluate_only_residual(sc.poses_init)
ts = []
for rep in range(6):
    f.evaluate_only_residual(sc.poses_init)
    t0 = time.perf_counter(); out = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=4); ts.append(time.perf_counter() - t0)
print("VXBA_WIDE_NWG=%s: damping_iter(4) median %.2f ms = %.3f ms per iteration" % (os.environ.get("VXBA_WIDE_NWG", "default"), 1e3 * np.median(ts[1:]), 1e3 * np.median(ts[1:]) / out["trace"].shape[0]))
