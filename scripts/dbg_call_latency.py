"""Development: wall time of ONE Lidar_BA_Optimizer::damping_iter call (3 iterations, cfg2) -- what a caller that runs the BA once per scan pays,
launch latencies, the read-back and the completion wait included (bench.py's headline loop amortises those over 300 steps)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from voxel_slam_amd import synth, vxba
sc = synth.make_config("cfg2")
f = vxba.LidarFactor(sc.win_size)
f.push_points(sc.n_voxels, sc.points_body, sc.cell_ptr)
f.evaluate_only_residual(sc.poses_init); f.snapshot_cache()
opt = vxba.Lidar_BA_Optimizer()
for it in (3, 1):
    ts = []
    for k in range(80):
        f.restore_cache()
        t0 = time.perf_counter(); out = opt.damping_iter(sc.poses_init, f, max_iter=it); ts.append(1e6 * (time.perf_counter() - t0))
    print("damping_iter(max_iter=%d): median %.1f us per call, min %.1f (%d iterations ran)" % (it, np.median(ts[10:]), np.min(ts[10:]), out["trace"].shape[0]))
