"""Development: median wall time of LI_BA_Optimizer::damping_iter (cfg2, 3 iterations) with the library VXBA_LIB names; VXBA_LI_TIMING=1
adds the host split.  Used by scripts/gpu_li_ab.sh to compare two builds on the same box."""
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
for name, opt in (("LI_BA_Optimizer", vxba.LI_BA_Optimizer()), ("LI_BA_OptimizerGravity", vxba.LI_BA_OptimizerGravity())):
    ts = []
    for k in range(40):
        for x, b in zip(facs, blobs0): x.blob[:] = b
        f.restore_cache()
        t2 = time.perf_counter(); out = opt.damping_iter(iw.states_init, f, facs, max_iter=3); ts.append(1e6 * (time.perf_counter() - t2))
    nit = out["trace"].shape[0]
    print("%s: damping_iter(3) median %.0f us, min %.0f us (%.1f us per iteration, %d iterations), final residual %.9e" % (name, np.median(ts[5:]), np.min(ts), np.median(ts[5:]) / nit, nit, out["trace"][-1, 1]))
