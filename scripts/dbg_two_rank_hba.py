"""Two (or more) ranks on ONE GPU through gloo: dist.hierarchical_ba_sharded (bottom-level windows round-robin, submaps all-gathered, the
wide top-level window voxel-sharded on the device by root-voxel hash, its packed system all-reduced through the host hook) against the
single-process hba.hierarchical_ba on the same session.  Launched by tests/test_gpu_two_rank.py through torch.distributed.run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from voxel_slam_amd import synth, vxba, hba, dist as vdist

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
K, wd, mg = int(os.environ.get("HBA_K", "45")), int(os.environ.get("HBA_WD", "6")), int(os.environ.get("HBA_MG", "3"))
xyz, fp, poses, gt = synth.make_scans(win_size=K, pts_per_scan=5000, extent=24.0, noise=0.005, seed=synth.MASTER_SEED + 950 + K, rot_sigma_deg=0.1, trans_sigma=0.02)
clouds = [xyz[fp[i]:fp[i + 1]].astype(np.float32) for i in range(K)]
coarse = vxba.VoxelizeParams(voxel_size=2.0, max_layer=2, min_points=10, min_eigen_value=0.02, eigen_ratio=(1 / 9, 1 / 9, 1 / 9, 1 / 9))
fine = vxba.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
got = vdist.hierarchical_ba_sharded(clouds, poses, coarse, fine, wdsize=wd, mgsize=mg, top_max_iter=2)
ref = hba.hierarchical_ba(clouds, poses, coarse, fine, wdsize=wd, mgsize=mg, top_max_iter=2)
et, er = synth.pose_errors(got["submap_poses"], ref["submap_poses"])
# every rank's poses, gathered: they must be the same bits
allp = [None] * world
dist.all_gather_object(allp, got["submap_poses"].tobytes())
same = all(b == allp[0] for b in allp)
nv = [None] * world
dist.all_gather_object(nv, [r["n_voxels"] for r in got["top_rounds"]])
tot = np.sum(np.asarray(nv), axis=0).tolist()
print("rank %d hba_sharded: pose diff %.2e %.2e, same bits on all ranks %s, top voxels per round %s sum %s vs %s, submap sizes equal %s, edges %d %d vs %d %d, windows %s" % (
    rank, et, er, same, nv[rank], tot, [r["n_voxels"] for r in ref["top_rounds"]], got["submap_sizes"] == ref["submap_sizes"],
    len(got["edges1"]), len(got["edges2"]), len(ref["edges1"]), len(ref["edges2"]), got["windows_of_rank"]), flush=True)
dist.destroy_process_group()
