"""PCIe-inclusive rate of the headline workload: points handed over as host arrays, K1 on the GPU, a
3-iteration Lidar_BA_Optimizer::damping_iter, poses back -- everything the boundary does for one window, per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from voxel_slam_amd import synth, vxba
sc = synth.make_config("cfg2")
f = vxba.LidarFactor(sc.win_size)
ts = {"push_points": [], "seed_cache": [], "damping_iter": [], "total": []}
for k in range(12):
    f.clear()
    t0 = time.perf_counter(); f.push_points(sc.n_voxels, sc.points_body, sc.cell_ptr)
    t1 = time.perf_counter(); f.evaluate_only_residual(sc.poses_init)
    t2 = time.perf_counter(); out = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=3)
    t3 = time.perf_counter()
    if k >= 2:
        ts["push_points"].append(t1 - t0); ts["seed_cache"].append(t2 - t1); ts["damping_iter"].append(t3 - t2); ts["total"].append(t3 - t0)
it = out["trace"].shape[0]
med = {k: 1e3 * float(np.median(v)) for k, v in ts.items()}
print({k: round(v, 3) for k, v in med.items()}, "iterations", it, "PCIe-inclusive it/s", round(it / (med["total"] * 1e-3), 1))
