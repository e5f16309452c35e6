"""Development: LI_BA_Optimizer::damping_iter at cfg2 with the reduced pose system solved inside the residual-sweep launch (default) against the
host pose solve with the trial poses fed to the waiting sweep, and against the plain shell -- same box, alternating."""
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
opt = vxba.LI_BA_Optimizer()
for rnd in range(2):
    for name, queued, dev in (("device pose solve", 1, 1), ("host pose solve + feed", 1, 0), ("plain shell", 0, 0)):
        f.set_option("li_queued_sweeps", queued); f.set_option("li_device_pose_solve", dev)
        ts, inside = [], []
        for k in range(60):
            for x, b in zip(facs, blobs0): x.blob[:] = b
            f.restore_cache()
            t2 = time.perf_counter(); out = opt.damping_iter(iw.states_init, f, facs, max_iter=3); ts.append(1e6 * (time.perf_counter() - t2)); inside.append(f.get_option("stat_li_last_call_us"))
        nit = out["trace"].shape[0]
        res[name] = out
        print("%-24s median %.0f us per call (%.1f us per iteration; inside the call %.1f), %d iterations, residual %.12e" % (name, np.median(ts[5:]), np.median(ts[5:]) / nit, np.median(inside[5:]) / nit, nit, out["trace"][-1, 1]))
a, b = res["device pose solve"], res["host pose solve + feed"]
print("device vs host pose solve: max |state diff| %.3e, trace flags equal %s, max rel trace diff %.3e" % (np.abs(a["states"] - b["states"]).max(), np.array_equal(a["trace"][:, 6:], b["trace"][:, 6:]), np.abs(a["trace"][:, :6] / b["trace"][:, :6] - 1).max()))
