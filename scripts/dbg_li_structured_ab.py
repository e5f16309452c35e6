"""Development: LI_BA_Optimizer::damping_iter with the structured host solve on / off (run on the GPU box; VXBA_LI_TIMING=1 prints the host split)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
opt = vxba.LI_BA_Optimizer()
res = {}
for mode in (1, 0, 1, 0):
    f.set_option("li_structured_solve", mode)
    ts = []
    for k in range(12):
        for x, b in zip(facs, blobs0): x.blob[:] = b
        f.restore_cache()
        t2 = time.perf_counter(); out = opt.damping_iter(iw.states_init, f, facs, max_iter=3); ts.append(1e6 * (time.perf_counter() - t2))
    res[mode] = out["states"].copy()
    print("structured=%d: damping_iter(3) median %.0f us (%.1f us per iteration), iterations %d" % (mode, np.median(ts), np.median(ts) / out["trace"].shape[0], out["trace"].shape[0]))
print("max |state difference| structured vs dense:", np.abs(res[1] - res[0]).max())
