import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from voxel_slam_amd import synth, vxba
from tests import _oracle as O
sc = synth.make_config("cfg2")
f = vxba.LidarFactor(sc.win_size)
f.push_points(sc.n_voxels, sc.points_body, sc.cell_ptr)
f.evaluate_only_residual(sc.poses_init); f.snapshot_cache()
iw = synth.make_imu(sc)
facs = []
for gyr, acc, dts in iw.samples:
    fac = vxba.IMU_PRE(iw.states_init[0, 15:18], iw.states_init[0, 18:21])
    for g, a, dt in zip(gyr, acc, dts):
        fac.add_imu(g, a, dt, iw.noise_meas, iw.noise_walk)
    facs.append(fac)
blobs0 = [x.blob.copy() for x in facs]
opt = vxba.LI_BA_Optimizer()
res = {}
for mode in ("0", "1", "0", "1"):
    f.set_option("li_device_loop", int(mode))
    ts = []
    for k in range(12):
        for x, b in zip(facs, blobs0): x.blob[:] = b
        f.restore_cache()
        t2 = time.perf_counter(); out = opt.damping_iter(iw.states_init, f, facs, max_iter=3); ts.append(1e6 * (time.perf_counter() - t2))
    res[mode] = out
    print("VXBA_LI_DEVICE=%s: median %.0f us per damping_iter(3) = %.1f us/iteration, %d iterations, trace accept %s" % (mode, np.median(ts[2:]), np.median(ts[2:]) / out["trace"].shape[0], out["trace"].shape[0], out["trace"][:, 6]))
a, b = res["0"], res["1"]
print("pose diff host vs device:", synth.pose_errors(a["states"][:, :12], b["states"][:, :12]), "v/bias diff", np.abs(a["states"][:, 12:21] - b["states"][:, 12:21]).max())
print("trace r1/r2 rel diff", np.abs(a["trace"][:, :2] / b["trace"][:, :2] - 1).max(), "hess rel", np.abs(a["hess"] - b["hess"]).max() / np.abs(a["hess"]).max())
