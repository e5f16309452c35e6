"""Round 6: the pieces a --gpus N prediction is built from: per-rank sweep times of the cfg2 window split N ways (strong) at W = 10,
the three-launch step the sharded loop runs (no fused launch with a collective attached), and the same for the weak leg's per-rank size (= cfg2)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from voxel_slam_amd import synth, vxba
us = lambda d: 1e3 * d["ms_sum"] / max(1, d["calls"])
for n in (1, 2, 4, 8):
    V = 50000 // n
    sc = synth.make_scene(win_size=10, pts_per_scan=100000 // n, n_voxels=V, seed=11)
    f = vxba.LidarFactor(sc.win_size, device=0)
    f.push_voxels(sc.clusters, sc.fix, sc.coe)
    f.evaluate_only_residual(sc.poses_init)
    f.snapshot_cache()
    for fused in (0, 1):
        f.set_option("fused_sweeps", fused)
        for _ in range(5):
            f.lm_steps(sc.poses_init, 150, 3)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); f.lm_steps(sc.poses_init, 150, 3); ts.append(time.perf_counter() - t0)
        f.set_profiling(1 | 2 | 4 | 32)
        f.lm_steps(sc.poses_init, 150, 3)
        kt = f.kernel_times(reset=True); ft = f.fused_time(reset=True)
        f.set_profiling(0)
        print(f"N={n} voxels/rank {V:6d} fused={fused}: {1e6 * np.median(ts) / 150:6.2f} us/step | K3 {us(kt['k3_hessian']):6.2f}  solve+K2 {us(kt['k2_residual']):6.2f}  reduction {us(kt['k3_finalize']):5.2f}  fused launch {us(ft):6.2f} x{ft['calls']}", flush=True)
    f.close()
