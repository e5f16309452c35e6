"""Development: timings of the wide-window (W ~ 100) path at a top-level-HBA-like size (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from voxel_slam_amd import synth, vxba
from tests import _oracle as O
W, V = 99, 100_000
sc = synth.make_scene(win_size=W, pts_per_scan=60_000, n_voxels=V, p_obs=0.05, seed=5)
print("nnz", sc.nnz, "mean observers", sc.nnz / V)
f = vxba.LidarFactor(W)
obs = sc.clusters[:, :, 9] != 0
row_ptr = np.concatenate([[0], np.cumsum(obs.sum(axis=1))]).astype(np.int64)
vv, fr = np.nonzero(obs)
ecl = np.ascontiguousarray(sc.clusters[vv, fr])
t0 = time.perf_counter(); f.push_voxels_csr(row_ptr, fr.astype(np.int32), ecl, sc.fix, sc.coe); print("push_voxels_csr %.1f ms" % (1e3 * (time.perf_counter() - t0)))
f.evaluate_only_residual(sc.poses_init)
f.acc_evaluate2(sc.poses_init)
print("device bytes after one residual + one Hessian sweep [MB]:", {k: round(v / 1e6, 1) for k, v in f.device_bytes().items()}, " (dense planes would be %.0f MB)" % (V * W * 80 / 1e6))
for name, fn in (("residual sweep", lambda: f.evaluate_only_residual(sc.poses_init)), ("Hessian sweep", lambda: f.acc_evaluate2(sc.poses_init))):
    fn(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    print("%s: %.3f ms (host call incl. D2H)" % (name, 1e3 * np.median(ts)))
f.set_profiling(3)
for _ in range(5):
    f.acc_evaluate2(sc.poses_init); f.evaluate_only_residual(sc.poses_init)
print(f.kernel_times(reset=True))
for mode in (1, 0):
    f.set_option("wide_device_solve", mode)
    for rep in range(3):
        f.evaluate_only_residual(sc.poses_init)
        t0 = time.perf_counter(); out = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=4); dt = time.perf_counter() - t0
        print("damping_iter(4), %s solve, run %d: %.2f ms (%.2f ms per iteration), %d iterations" % ("device" if mode else "host", rep, 1e3 * dt, 1e3 * dt / out["trace"].shape[0], out["trace"].shape[0]))
print("device bytes at the end [MB]:", {k: round(v / 1e6, 1) for k, v in f.device_bytes().items()})
if os.environ.get("WIDE_ORACLE", "0") != "1": sys.exit(0)
fo = O.Oracle(W); fo.push_voxels(sc.clusters, sc.fix, sc.coe); fo.evaluate_only_residual(sc.poses_init)
t0 = time.perf_counter(); ref = fo.damping_iter(sc.poses_init, max_iter=4, thd_num=5); dto = time.perf_counter() - t0
print("oracle damping_iter(4), 5 threads: %.1f ms" % (1e3 * dto), "pose diff", synth.pose_errors(out["poses"], ref["poses"]))
