"""Development: per-wave timelines of K2 / K3 from the instrumented kernels (run on the GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1]
if which == "k2":
    os.environ["VXBA_DBG"] = "1"
else:
    os.environ["VXBA_DBG"] = "1"
from voxel_slam_amd import synth, vxba
sc = synth.make_config("cfg2")
f = vxba.LidarFactor(sc.win_size)
f.push_points(sc.n_voxels, sc.points_body, sc.cell_ptr)
f.evaluate_only_residual(sc.poses_init)
for _ in range(3):
    f.acc_evaluate2(sc.poses_init); f.evaluate_only_residual(sc.poses_init)
vxba.debug_stamps(1, clear=True)
if which == "solve":
    from voxel_slam_amd.vxba import Lidar_BA_Optimizer
    Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=1)
    st = vxba.debug_stamps(4001).astype(np.int64)[4000, :6]
    print("solve kernel stamps (cycles since start):", st - st[0])
    sys.exit(0)
if which == "fused":
    from voxel_slam_amd.vxba import Lidar_BA_Optimizer
    Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=1)
    full = vxba.debug_stamps(4001).astype(np.int64)
    sol = full[4000, :6]
    n = (sc.n_voxels + 63) // 64
    k2 = full[:n, :6]
    t0 = min(sol[0], k2[:, 0].min())
    print("solver stamps (cycles since kernel start): start %d, done-checked %d, loaded %d, eliminated %d, back-substituted %d, end %d" % tuple(sol - t0))
    names = ["start", "loads landed", "cov done", "eig done", "end", "flag seen"]
    for k in (0, 1, 5, 2, 3, 4):
        print("voxel waves %-13s min %7d  median %7d  max %7d" % (names[k], k2[:, k].min() - t0, np.median(k2[:, k]) - t0, k2[:, k].max() - t0))
    sys.exit(0)
if which == "k2":
    f.evaluate_only_residual(sc.poses_init); n = (sc.n_voxels + 63) // 64; ns = 5
else:
    f.acc_evaluate2(sc.poses_init); n = 1024; ns = 7
full = vxba.debug_stamps(n).astype(np.int64)
st = full[:, :ns]
if which != "k2":
    it = full[:, 8:28]
    ok = it[:, 13] > 0          # waves with >= 7 loop iterations
    top = it[ok][:, 0:14:2]; ready = it[ok][:, 1:14:2]
    print("loop iterations: wait-for-loads cycles (median per iteration):", np.median(ready - top, axis=0))
    print("iteration period cycles (median):", np.median(np.diff(top, axis=1), axis=0))
    ph = full[ok][:, 28:32]
    okp = ph[:, 3] > 0
    print("one loop iteration split (cycles, median): MFMA tile %.0f | phase A %.0f | LDS stores %.0f" % tuple(np.median(np.diff(ph[okp], axis=1), axis=0)))
t0 = st[:, 0].min()
rel = (st - t0) / 100.0     # s_memtime ticks at 100 MHz -> us
print(which, "waves", n, "kernel span us:", (st[:, -1].max() - t0) / 100.0)
print("per-slot (us since first wave start): min / median / max")
for k in range(ns):
    print(k, "%.2f %.2f %.2f" % (rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max()))
d = np.diff(st, axis=1) / 100.0
print("per-phase durations (us): median", np.median(d, axis=0), "max", d.max(axis=0))

if which == "solve":
    pass
