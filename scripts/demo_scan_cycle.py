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

# ---- the same cycle with the tree resident on the device (vxba_map_*): only the raw scan goes up and the poses come down --------------
print()
from tests.test_oracle_octree import PRM
S, win, pts = 14, 10, 100_000
xyz, fp, poses_gt, _ = synth.make_scans(win_size=S, pts_per_scan=pts, extent=60.0, seed=synth.MASTER_SEED + 950)
m = vxba.LocalMap(win_size=win, **PRM)
fac = vxba.LidarFactor(win)
est2 = vxba.LioEstimator(PRM["voxel_size"], PRM["max_layer"])
stage = {k: [] for k in ("var_init", "lio_state_estimation", "pvec_update (resident)", "cut_voxel (device scan)", "recut + tras_opt into the factor", "damping_iter (3 iterations)", "margi (device cache) + slide", "plane export to the odometry map")}
xb, win_count = [], 0
cov = np.eye(15) * 1e-4
def lap(key, fn):
    t0 = time.perf_counter(); r = fn(); stage[key].append(1e3 * (time.perf_counter() - t0)); return r
for k in range(S):
    scan32 = xyz[fp[k]:fp[k + 1]].astype(np.float32)
    prior = np.concatenate([poses_gt[k], np.zeros(9), [0, 0, -9.8]])
    lap("var_init", lambda: est2.var_init(scan32))
    state, cv = prior, cov
    if k >= 3:
        r = lap("lio_state_estimation", lambda: est2.lio_state_estimation(prior, cov)); state, cv = r["state"], r["cov"]
    lap("pvec_update (resident)", lambda: est2.pvec_update(state, cv, resident=True))
    win_count += 1; xb.append(state[:12].copy()); fac.clear()
    lap("cut_voxel (device scan)", lambda: m.cut_voxel_lio(win_count - 1, est2))
    nf = lap("recut + tras_opt into the factor", lambda: m.recut(win_count, np.stack(xb), fac))
    if win_count >= win:
        out = lap("damping_iter (3 iterations)", lambda: vxba.Lidar_BA_Optimizer().damping_iter(np.stack(xb), fac, max_iter=3))
        lap("margi (device cache) + slide", lambda: (m.margi(win_count, out["poses"], fac), m.slide(1)))
        xb = [p for p in out["poses"][1:]]; win_count -= 1
    lap("plane export to the odometry map", lambda: m.export_planes(est2))
print("device-resident map, %d-point scans, window %d, %s, %d factor voxels in the last window:" % (pts, win, m.counts(), nf))
for key, v in stage.items():
    if v:
        print("%-86s %8.3f ms" % ("  " + key, float(np.median(v[-4:]))))
