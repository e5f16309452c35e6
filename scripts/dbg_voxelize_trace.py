"""Kernel-level view of the batch factor construction (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from voxel_slam_amd import synth, vxba
W = 10
xyz, fp, poses, _ = synth.make_scans(win_size=W, pts_per_scan=100_000, extent=60.0)
P = vxba.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
f = vxba.LidarFactor(W)
for k in range(6):
    f.clear()
    t0 = time.perf_counter(); n = f.voxelize_push(xyz, fp, poses, P, want_ids=False); dt = time.perf_counter() - t0
    print("voxelize_push: %.2f ms, %d factor voxels" % (1e3 * dt, n))
