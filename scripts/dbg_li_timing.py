"""Development: where does one LI-BA solve spend its wall time (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
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
for k in range(6):
    for x, b in zip(facs, blobs0): x.blob[:] = b
    t0 = time.perf_counter(); f.restore_cache(); f.evaluate_only_residual(sc.poses_init); t1 = time.perf_counter()
    f.restore_cache()
    t2 = time.perf_counter(); out = opt.damping_iter(iw.states_init, f, facs, max_iter=3); t3 = time.perf_counter()
    t4 = time.perf_counter(); r = f.evaluate_only_residual(sc.poses_init); t5 = time.perf_counter()
    t6 = time.perf_counter(); H = f.acc_evaluate2(sc.poses_init); t7 = time.perf_counter()
    print("restore+K2 %.0f us | damping_iter(3) %.0f us | K2 host call %.0f us | K3 host call %.0f us | iters %d" % (1e6*(t1-t0), 1e6*(t3-t2), 1e6*(t5-t4), 1e6*(t7-t6), out["trace"].shape[0]))

def solve_time(tag):
    ts = []
    for k in range(5):
        for x, b in zip(facs, blobs0): x.blob[:] = b
        f.restore_cache()
        t2 = time.perf_counter(); opt.damping_iter(iw.states_init, f, facs, max_iter=3); ts.append(1e6 * (time.perf_counter() - t2))
    print(tag, ["%.0f" % t for t in ts])

solve_time("baseline")
f.set_stream(torch.cuda.current_stream().cuda_stream); solve_time("after set_stream(torch current)")
f.lm_steps(sc.poses_init, 30, 3); solve_time("after lm_steps")
f.set_profiling(1); f.lm_steps(sc.poses_init, 30, 3); f.set_profiling(0); f.kernel_times(reset=True); solve_time("after profiled lm_steps")
f.set_profiling(15); f.lm_steps(sc.poses_init, 30, 3); f.set_profiling(0); f.kernel_times(reset=True); solve_time("after mask-15 lm_steps")
