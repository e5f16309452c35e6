"""Round 6: what the hipEvent brackets of set_profiling cost the LM step (cfg2, vxba_lm_steps) -- steps per call x profiling mask."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from voxel_slam_amd import synth, vxba
sc = synth.make_config("cfg2")
f = vxba.LidarFactor(sc.win_size, device=0)
f.push_voxels(sc.clusters, sc.fix, sc.coe)
f.evaluate_only_residual(sc.poses_init)
f.snapshot_cache()
for _ in range(20):
    f.lm_steps(sc.poses_init, 300, 3)
for rnd in range(3):
    for steps in (20, 150):
        for mask in (0, 1 | 32, 1, 32):
            f.set_profiling(mask)
            ts = []
            for _ in range(15):
                t0 = time.perf_counter()
                f.lm_steps(sc.poses_init, steps, 3)
                ts.append(time.perf_counter() - t0)
            f.kernel_times(reset=True); f.fused_time(reset=True)
            print(f"steps {steps:4d} profiling mask {mask:2d}: median {1e6 * np.median(ts) / steps:6.2f} us/step  min {1e6 * min(ts) / steps:6.2f}", flush=True)
f.set_profiling(0)
