"""Development: only the device-resident half of scripts/demo_scan_cycle.py (the scan cycle with the tree on the GPU), for rocprofv3 traces
(scripts/gpu_profile_map.sh)."""

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from voxel_slam_amd import synth, vxba

# ---- the same cycle with the tree resident on the device (vxba_map_*): only the raw scan goes up and the poses come down --------------
print()
from tests.test_oracle_octree import PRM
S, win, pts = 14, 10, 100_000
xyz, fp, poses_gt, _ = synth.make_scans(win_size=S, pts_per_scan=pts, extent=60.0, seed=synth.MASTER_SEED + 950)
m = vxba.LocalMap(win_size=win, **PRM)
fac = vxba.LidarFactor(win)
est2 = vxba.LioEstimator(PRM["voxel_size"], PRM["max_layer"])
stage = {k: [] for k in ("var_init", "lio_state_estimation", "pvec_update (resident)", "cut_voxel (device scan)", "recut + tras_opt into the factor", "damping_iter (3 iterations)", "margi (device cache) + slide", "plane export to the odometry map")}
xb, win_count = [], 0
cov = np.eye(15) * 1e-4
def lap(key, fn):
    t0 = time.perf_counter(); r = fn(); stage[key].append(1e3 * (time.perf_counter() - t0)); return r
for k in range(S):
    scan32 = xyz[fp[k]:fp[k + 1]].astype(np.float32)
    prior = np.concatenate([poses_gt[k], np.zeros(9), [0, 0, -9.8]])
    lap("var_init", lambda: est2.var_init(scan32))
    state, cv = prior, cov
    if k >= 3:
        r = lap("lio_state_estimation", lambda: est2.lio_state_estimation(prior, cov)); state, cv = r["state"], r["cov"]
    lap("pvec_update (resident)", lambda: est2.pvec_update(state, cv, resident=True))
    win_count += 1; xb.append(state[:12].copy()); fac.clear()
    lap("cut_voxel (device scan)", lambda: m.cut_voxel_lio(win_count - 1, est2))
    nf = lap("recut + tras_opt into the factor", lambda: m.recut(win_count, np.stack(xb), fac))
    if win_count >= win:
        out = lap("damping_iter (3 iterations)", lambda: vxba.Lidar_BA_Optimizer().damping_iter(np.stack(xb), fac, max_iter=3))
        lap("margi (device cache) + slide", lambda: (m.margi(win_count, out["poses"], fac), m.slide(1)))
        xb = [p for p in out["poses"][1:]]; win_count -= 1
    lap("plane export to the odometry map", lambda: m.export_planes(est2))
print("device-resident map, %d-point scans, window %d, %s, %d factor voxels in the last window:" % (pts, win, m.counts(), nf))
for key, v in stage.items():
    if v:
        print("%-86s %8.3f ms" % ("  " + key, float(np.median(v[-4:]))))
