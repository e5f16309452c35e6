"""Two (or more) ranks on ONE GPU through gloo (plumbing check of the sharded loops): every rank owns a voxel shard of the same window;
the sharded damping_iter / lm_steps must reproduce the single-factor run on the whole window."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from voxel_slam_amd import synth, vxba, dist as vdist

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
sc = synth.make_scene(win_size=10, pts_per_scan=40_000, n_voxels=6000, p_obs=0.9, seed=99, rot_sigma_deg=0.3, trans_sigma=0.08)
lo, hi = vdist.shard_bounds(sc.n_voxels, world, rank)
full = vxba.LidarFactor(10); full.push_voxels(sc.clusters, sc.fix, sc.coe); full.evaluate_only_residual(sc.poses_init)
ref = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, full, max_iter=6)
full.evaluate_only_residual(sc.poses_init); full.snapshot_cache()
ref_p, ref_r, ref_s = full.lm_steps(sc.poses_init, 12, 3)
f = vxba.LidarFactor(10); f.push_voxels(sc.clusters[lo:hi], sc.fix[lo:hi], sc.coe[lo:hi])
keep = vdist.attach_allreduce(f)
f.evaluate_only_residual(sc.poses_init)
got = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=6)
et, er = synth.pose_errors(got["poses"], ref["poses"])
print("rank %d damping_iter: trace accept %s vs %s, pose diff %.2e %.2e, resis %s vs %s" % (rank, got["trace"][:, 6], ref["trace"][:, 6], et, er, got["resis"], ref["resis"]), flush=True)
f.evaluate_only_residual(sc.poses_init); f.snapshot_cache()
p, r, s = f.lm_steps(sc.poses_init, 12, 3)
et, er = synth.pose_errors(p, ref_p)
print("rank %d lm_steps: stats %s vs %s, pose diff %.2e %.2e, resis %s vs %s" % (rank, s, ref_s, et, er, r, ref_r), flush=True)
dist.barrier()
