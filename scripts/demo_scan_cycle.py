"""One scan's worth of work through the library, stage by stage (GPU box): voxel-grid filter -> var_init -> lio_state_estimation ->
pvec_update -> per-leaf clusters / plane fits / plane covariances for the leaves the scan touched -> plane-map update -> the
LiDAR-inertial BA of the window.  The host octree's bookkeeping (which leaf a point belongs to, which leaves exist) is taken as
given -- it is the part of the local map this library does not own.  Prints the median time of each stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from voxel_slam_amd import synth, vxba

def timed(fn, reps=12):
    ts = []
    out = None
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts[2:])), out

rows = []
pm = synth.make_plane_map(n_roots=20_000, extent=20, seed=synth.MASTER_SEED + 910)
raw = synth.make_lio_scan(pm, n_points=240_000, seed=synth.MASTER_SEED + 912, coherent=True, planes_hit=12_000)
ms, ds = timed(lambda: vxba.down_sampling_voxel(raw.xyz, 0.1)); rows.append(("down_sampling_voxel (240k raw points -> %d)" % ds.shape[0], ms))
est = vxba.LioEstimator(pm.voxel_size, pm.max_layer)
ms, _ = timed(lambda: est.map_update(*pm.args()), reps=4); rows.append(("plane map upload, whole map (%d leaves; once)" % len(pm.layer), ms))
ms, _ = timed(lambda: est.var_init(ds)); rows.append(("var_init (%d points)" % ds.shape[0], ms))
ms, res = timed(lambda: est.lio_state_estimation(raw.state_init, raw.cov)); rows.append(("lio_state_estimation (%d iterations, %d matches)" % (res["iterations"], res["match_num"]), ms))
ms, (pw, vw) = timed(lambda: est.pvec_update(res["state"], res["cov"])); rows.append(("pvec_update", ms))
# the leaves this scan touched, as the host tree would bucket them (here: by root voxel)
cell = np.floor(pw / pm.voxel_size).astype(np.int64)
order = np.lexsort((cell[:, 2], cell[:, 1], cell[:, 0]))
pw_s, vw_s = np.ascontiguousarray(pw[order]), np.ascontiguousarray(vw[order])
_, first = np.unique(cell[order], axis=0, return_index=True)
cell_ptr = np.concatenate([np.sort(first), [pw.shape[0]]]).astype(np.int64)
nleaf = cell_ptr.shape[0] - 1
ms, cl = timed(lambda: vxba.build_clusters(pw_s, cell_ptr)); rows.append(("cluster build for %d touched leaves" % nleaf, ms))
ms, (ev, U, flags) = timed(lambda: vxba.plane_fit_judge(cl, min_point=5, min_eigen_value=0.0025, eigen_ratio_thre=0.05)); rows.append(("plane fit + plane_judge", ms))
ms, ca = timed(lambda: vxba.cov_add_build(pw_s, vw_s, cell_ptr)); rows.append(("cov_add (sum of Bf_var)", ms))
# the same three stages with the covariances left on the device: world points only -> the host's bucketing as indices -> per-leaf increments
ms, pw2 = timed(lambda: est.pvec_update(res["state"], res["cov"], with_var=False)); rows.append(("  resident variant: pvec_update, world points only", ms))
ms, (cl2, ca2) = timed(lambda: est.leaf_stats(cell_ptr, order)); rows.append(("  resident variant: leaf_stats (clusters + cov_add of the touched leaves)", ms))
assert np.array_equal(pw2, pw) and np.array_equal(cl2, cl) and np.all(np.abs(ca2 - ca) <= 1e-11 * np.abs(ca).max(axis=(1, 2), keepdims=True))
good = (flags & 3) == 3
ms, pl = timed(lambda: vxba.plane_update(cl[good], ev[good], U[good], ca[good])); rows.append(("plane_update (%d planes)" % int(good.sum()), ms))
loc = cell[order][np.sort(first)][good]
ms, _ = timed(lambda: est.map_update(loc, np.zeros(len(loc), dtype=np.int32), np.zeros(len(loc), dtype=np.int32), pl["center"], pl["normal"], pl["plane_var"], pl["radius"]))
rows.append(("plane map update (%d leaves)" % len(loc), ms))
# the window's LiDAR-inertial BA (cfg2-sized window)
sc = synth.make_config("cfg2")
f = vxba.LidarFactor(sc.win_size); f.push_points(sc.n_voxels, sc.points_body, sc.cell_ptr); f.evaluate_only_residual(sc.poses_init); f.snapshot_cache()
iw = synth.make_imu(sc)
facs = []
for gyr, acc, dts in iw.samples:
    fac = vxba.IMU_PRE(iw.states_init[0, 15:18], iw.states_init[0, 18:21])
    for g, a, dt in zip(gyr, acc, dts):
        fac.add_imu(g, a, dt, iw.noise_meas, iw.noise_walk)
    facs.append(fac)
blobs0 = [x.blob.copy() for x in facs]
def li():
    for x, b in zip(facs, blobs0): x.blob[:] = b
    f.restore_cache()
    return vxba.LI_BA_Optimizer().damping_iter(iw.states_init, f, facs, max_iter=3)
ms, out = timed(li); rows.append(("LI_BA_Optimizer::damping_iter (W=10, 50k voxels, %d iterations)" % out["trace"].shape[0], ms))
ms, (evc, Uc, mc) = timed(lambda: f.read_cache()); rows.append(("read back pcr_adds / eig_values / eig_vectors for margi (50k voxels)", ms))
for name, ms in rows:
    print("%-86s %8.3f ms" % (name, ms))
