"""Two (or more) ranks on ONE GPU through gloo (plumbing check of the sharded loops): every rank owns a voxel shard of the same window;
the sharded damping_iter / lm_steps must reproduce the single-factor run on the whole window."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from voxel_slam_amd import synth, vxba, dist as vdist

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
COLLECTIVE = os.environ.get("VXBA_TWO_RANK_COLLECTIVE", "hook")      # "hook": torch.distributed through the host callback; "peer": one-shot mailbox all-reduce (hipIpc)
def attach(fac):
    if COLLECTIVE == "peer":
        vdist.attach_peer(fac)
        return None
    return vdist.attach_allreduce(fac)
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
sc = synth.make_scene(win_size=10, pts_per_scan=40_000, n_voxels=6000, p_obs=0.9, seed=99, rot_sigma_deg=0.3, trans_sigma=0.08)
lo, hi = vdist.shard_bounds(sc.n_voxels, world, rank)
full = vxba.LidarFactor(10); full.push_voxels(sc.clusters, sc.fix, sc.coe); full.evaluate_only_residual(sc.poses_init)
ref = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, full, max_iter=6)
full.evaluate_only_residual(sc.poses_init); full.snapshot_cache()
ref_p, ref_r, ref_s = full.lm_steps(sc.poses_init, 12, 3)
f = vxba.LidarFactor(10); f.push_voxels(sc.clusters[lo:hi], sc.fix[lo:hi], sc.coe[lo:hi])
keep = attach(f)
f.evaluate_only_residual(sc.poses_init)
got = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=6)
et, er = synth.pose_errors(got["poses"], ref["poses"])
print("rank %d damping_iter: trace accept %s vs %s, pose diff %.2e %.2e, resis %s vs %s" % (rank, got["trace"][:, 6], ref["trace"][:, 6], et, er, got["resis"], ref["resis"]), flush=True)
f.evaluate_only_residual(sc.poses_init); f.snapshot_cache()
p, r, s = f.lm_steps(sc.poses_init, 12, 3)
et, er = synth.pose_errors(p, ref_p)
assert s == ref_s
print("rank %d lm_steps: stats %s vs %s, pose diff %.2e %.2e, resis %s vs %s" % (rank, s, ref_s, et, er, r, ref_r), flush=True)

# an easier window (steps get accepted) for the bench-mode loop and the LiDAR-inertial shell (host loop between the sweeps)
sc = synth.make_scene(win_size=10, pts_per_scan=40_000, n_voxels=6000, p_obs=0.9, seed=199)
lo, hi = vdist.shard_bounds(sc.n_voxels, world, rank)
full = vxba.LidarFactor(10); full.push_voxels(sc.clusters, sc.fix, sc.coe); full.evaluate_only_residual(sc.poses_init); full.snapshot_cache()
ref_p, ref_r, ref_s = full.lm_steps(sc.poses_init, 12, 3)
f = vxba.LidarFactor(10); f.push_voxels(sc.clusters[lo:hi], sc.fix[lo:hi], sc.coe[lo:hi])
keep2 = attach(f)
f.evaluate_only_residual(sc.poses_init); f.snapshot_cache()
p, r, s = f.lm_steps(sc.poses_init, 12, 3)
et, er = synth.pose_errors(p, ref_p)
print("rank %d lm_steps_easy: stats %s vs %s, pose diff %.2e %.2e, resis %s vs %s" % (rank, s, ref_s, et, er, r, ref_r), flush=True)
iw = synth.make_imu(sc, seed=101)
def factors():
    out = []
    for gyr, acc, dts in iw.samples:
        fac = vxba.IMU_PRE(iw.states_init[0, 15:18], iw.states_init[0, 18:21])
        for g, a, dt in zip(gyr, acc, dts):
            fac.add_imu(g, a, dt, iw.noise_meas, iw.noise_walk)
        out.append(fac)
    return out
full.evaluate_only_residual(sc.poses_init); f.evaluate_only_residual(sc.poses_init)
ref_li = vxba.LI_BA_Optimizer().damping_iter(iw.states_init, full, factors(), max_iter=4)
got_li = vxba.LI_BA_Optimizer().damping_iter(iw.states_init, f, factors(), max_iter=4)
et, er = synth.pose_errors(got_li["states"][:, :12], ref_li["states"][:, :12])
print("rank %d li_damping_iter: trace accept %s vs %s, pose diff %.2e %.2e, vbias diff %.2e" % (rank, got_li["trace"][:, 6], ref_li["trace"][:, 6], et, er,
      np.abs(got_li["states"][:, 12:21] - ref_li["states"][:, 12:21]).max()), flush=True)

# a wide window (sparse-incidence sweeps, host LM) sharded the same way
scw = synth.make_scene(win_size=20, pts_per_scan=6000, n_voxels=2400, p_obs=0.25, seed=123, rot_sigma_deg=0.1, trans_sigma=0.03)
lo, hi = vdist.shard_bounds(scw.n_voxels, world, rank)
fullw = vxba.LidarFactor(20); fullw.push_voxels(scw.clusters, scw.fix, scw.coe); fullw.evaluate_only_residual(scw.poses_init)
refw = vxba.Lidar_BA_Optimizer().damping_iter(scw.poses_init, fullw, max_iter=4)
fw = vxba.LidarFactor(20); fw.push_voxels(scw.clusters[lo:hi], scw.fix[lo:hi], scw.coe[lo:hi])
keepw = vdist.attach_allreduce(fw)      # wide windows always go through the collective library (2.9 MB buffers)
fw.evaluate_only_residual(scw.poses_init)
gotw = vxba.Lidar_BA_Optimizer().damping_iter(scw.poses_init, fw, max_iter=4)
et, er = synth.pose_errors(gotw["poses"], refw["poses"])
print("rank %d wide damping_iter: trace accept %s vs %s, pose diff %.2e %.2e" % (rank, gotw["trace"][:, 6], refw["trace"][:, 6], et, er), flush=True)
if COLLECTIVE == "peer":
    assert f.peer_status() == 0
dist.barrier()
