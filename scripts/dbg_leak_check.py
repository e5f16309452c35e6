"""Create / use / destroy the library's handles many times and watch the device's free memory (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from voxel_slam_amd import synth, vxba

def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20

sc = synth.make_scene(win_size=10, pts_per_scan=20000, n_voxels=3000, seed=3)
xyz, fp, poses, _ = synth.make_scans(win_size=6, pts_per_scan=20000)
P = vxba.VoxelizeParams()
pm = synth.make_plane_map(n_roots=1500, extent=6, seed=4); ls = synth.make_lio_scan(pm, n_points=8000, seed=5)
iw = synth.make_imu(sc)
marks = []
scw = None
for rep in range(60):
    f = vxba.LidarFactor(10); f.push_points(sc.n_voxels, sc.points_body, sc.cell_ptr); f.evaluate_only_residual(sc.poses_init)
    vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=3)
    if rep % 3 == 0:   # f32 re-centred cluster rows: their copy is allocated on the first residual sweep and freed with the handle
        f.set_precision("mixed_f32_clusters"); f.evaluate_only_residual(sc.poses_init); vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=2)
        if rep % 6 == 0: f.set_precision("f64")
    facs = []
    for gyr, acc, dts in iw.samples:
        fac = vxba.IMU_PRE(iw.states_init[0, 15:18], iw.states_init[0, 18:21])
        for g, a, dt in zip(gyr[:5], acc[:5], dts[:5]):
            fac.add_imu(g, a, dt, iw.noise_meas, iw.noise_walk)
        facs.append(fac)
    f.set_option("li_queued_sweeps", rep % 2)
    vxba.LI_BA_Optimizer().damping_iter(iw.states_init, f, facs, max_iter=2)
    f.close()
    f2 = vxba.LidarFactor(6); f2.voxelize_push(xyz, fp, poses, P, want_ids=False); f2.close()
    w = vxba.LidarFactor(24)
    if scw is None:
        scw = synth.make_scene(win_size=24, pts_per_scan=6000, n_voxels=1500, p_obs=0.2, seed=6)
        obs = scw.clusters[:, :, 9] != 0
        w_rp = np.concatenate([[0], np.cumsum(obs.sum(axis=1))]).astype(np.int64); w_v, w_fr = np.nonzero(obs); w_cl = np.ascontiguousarray(scw.clusters[w_v, w_fr])
    w.push_voxels_csr(w_rp, w_fr.astype(np.int32), w_cl, scw.fix, scw.coe); w.evaluate_only_residual(scw.poses_init)
    vxba.Lidar_BA_Optimizer().damping_iter(scw.poses_init, w, max_iter=2)          # compressed-row store, pair index, single-launch Cholesky
    g = vxba.LioEstimator(pm.voxel_size, pm.max_layer); g.map_update(*pm.args()); g.var_init(ls.xyz); g.lio_state_estimation(ls.state_init, ls.cov); g.close(); w.close()
    vxba.down_sampling_voxel(ls.xyz, 0.2)
    if rep % 10 == 9:
        marks.append(free_mb())
        print("after %3d rounds: free %.1f MB" % (rep + 1, marks[-1]), flush=True)
print("drift over the last 40 rounds: %.1f MB" % (marks[1] - marks[-1]))
