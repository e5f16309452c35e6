import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from voxel_slam_amd import synth, vxba
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
res = {}
for mode in (1, 0, 1, 0):
    f.set_option("li_queued_sweeps", mode)
    for name, opt in (("LI_BA_Optimizer", vxba.LI_BA_Optimizer()), ("LI_BA_OptimizerGravity", vxba.LI_BA_OptimizerGravity())):
        ts, inside = [], []
        for k in range(40):
            for x, b in zip(facs, blobs0): x.blob[:] = b
            f.restore_cache()
            t2 = time.perf_counter(); out = opt.damping_iter(iw.states_init, f, facs, max_iter=3); ts.append(1e6 * (time.perf_counter() - t2)); inside.append(f.get_option("stat_li_last_call_us"))
        nit = out["trace"].shape[0]
        res[(mode, name)] = out
        print("queued=%d %s: median %.0f us per call (%.1f us per iteration; inside the call %.1f), %d iterations, residual %.9e" % (mode, name, np.median(ts[5:]), np.median(ts[5:]) / nit, np.median(inside[5:]) / nit, nit, out["trace"][-1, 1]))
for name in ("LI_BA_Optimizer", "LI_BA_OptimizerGravity"):
    a, b = res[(1, name)], res[(0, name)]
    print(name, "queued vs plain: max |state diff| %.3e, trace equal %s" % (np.abs(a["states"] - b["states"]).max(), np.array_equal(a["trace"][:, 6:], b["trace"][:, 6:])))
