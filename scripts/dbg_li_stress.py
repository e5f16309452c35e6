"""Development: many back-to-back LiDAR-inertial solves in the queued-sweeps mode (a smaller window than cfg2, so that thousands fit in seconds);
stops at the first failure and prints the library's diagnostics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from voxel_slam_amd import synth, vxba
n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sc = synth.make_scene(win_size=10, pts_per_scan=40000, n_voxels=int(sys.argv[2]) if len(sys.argv) > 2 else 20000, seed=11)
f = vxba.LidarFactor(sc.win_size)
f.push_voxels(sc.clusters, sc.fix, sc.coe)
f.evaluate_only_residual(sc.poses_init); f.snapshot_cache()
iw = synth.make_imu(sc)
facs = []
for gyr, acc, dts in iw.samples:
    fac = vxba.IMU_PRE(iw.states_init[0, 15:18], iw.states_init[0, 18:21])
    for g, a, dt in zip(gyr, acc, dts):
        fac.add_imu(g, a, dt, iw.noise_meas, iw.noise_walk)
    facs.append(fac)
blobs0 = [x.blob.copy() for x in facs]
opts = (vxba.LI_BA_Optimizer(), vxba.LI_BA_OptimizerGravity())
ref = {}
t0 = time.time()
for k in range(n_calls):
    for x, b in zip(facs, blobs0): x.blob[:] = b
    f.restore_cache()
    o = opts[k & 1]
    try:
        out = o.damping_iter(iw.states_init, f, facs, max_iter=2 + (k // 2) % 4)
    except vxba.VxbaError as e:
        print("call %d FAILED after %.1f s: %s" % (k, time.time() - t0, e)); sys.exit(1)
    key = (k & 1, 2 + (k // 2) % 4)
    if key in ref:
        if not np.array_equal(ref[key], out["states"]):
            print("call %d: result differs from the first call of its kind by %.3e" % (k, np.abs(ref[key] - out["states"]).max())); sys.exit(1)
    else:
        ref[key] = out["states"].copy()
print("%d calls, all identical to their first of a kind, %.1f s" % (n_calls, time.time() - t0))
