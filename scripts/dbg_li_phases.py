"""Development: VXBA_LI_TIMING=1 phase breakdown of LI_BA_Optimizer::damping_iter calls at cfg2 (prints the library's own line per call)."""
import os, sys
os.environ["VXBA_LI_TIMING"] = "1"
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
opt = vxba.LI_BA_Optimizer()
for k in range(12):
    for x, b in zip(facs, blobs0): x.blob[:] = b
    f.restore_cache()
    opt.damping_iter(iw.states_init, f, facs, max_iter=3)
