"""Two (or more) ranks on ONE GPU through gloo: dist.hba_pass -- the any-N driver over the C-ABI halves of the pass (vxba_hba_bottom over the rank's
windows, the packed submaps all-gathered, vxba_hba_top voxel-sharded on the device by root-voxel hash with its packed system all-reduced through the
host hook) -- against the one-rank pass (HbaSession.run_pass = vxba_hba_pass) on the same session, with the time of both.
Launched by tests/test_gpu_two_rank.py through torch.distributed.run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from voxel_slam_amd import synth, vxba, hba, dist as vdist

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
K, wd, mg = int(os.environ.get("HBA_K", "45")), int(os.environ.get("HBA_WD", "6")), int(os.environ.get("HBA_MG", "3"))
pts = int(os.environ.get("HBA_PTS", "5000"))
xyz, fp, poses, gt = synth.make_scans(win_size=K, pts_per_scan=pts, extent=24.0, noise=0.005, seed=synth.MASTER_SEED + 950 + K, rot_sigma_deg=0.1, trans_sigma=0.02)
clouds = [xyz[fp[i]:fp[i + 1]].astype(np.float32) for i in range(K)]
coarse = vxba.VoxelizeParams(voxel_size=2.0, max_layer=2, min_points=10, min_eigen_value=0.02, eigen_ratio=(1 / 9, 1 / 9, 1 / 9, 1 / 9))
fine = vxba.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
ses = vxba.HbaSession(device=0); ses.add_keyframes(clouds)
ctx = {}
nthr = int(os.environ.get("HBA_THREADS", "2"))
got = vdist.hba_pass(ses, poses, coarse, fine, wdsize=wd, mgsize=mg, top_max_iter=2, n_threads=nthr, ctx=ctx)
dist.barrier(); torch.cuda.synchronize()
t0 = time.perf_counter()
ph = dict(bottom=0.0, exchange=0.0, top=0.0)
for _ in range(2):
    got = vdist.hba_pass(ses, poses, coarse, fine, wdsize=wd, mgsize=mg, top_max_iter=2, n_threads=nthr, ctx=ctx)
    for k in ph:
        ph[k] += got["phase_s"][k] / 2
torch.cuda.synchronize(); dist.barrier()
t_n = (time.perf_counter() - t0) / 2
# the one-rank pass, on rank 0 alone (the others wait: one GPU)
ref, t_1 = None, 0.0
if rank == 0:
    one = vxba.HbaSession(device=0); one.add_keyframes(clouds)
    ref = one.run_pass(poses, coarse, fine, wdsize=wd, mgsize=mg, top_max_iter=2, n_threads=nthr * world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        ref = one.run_pass(poses, coarse, fine, wdsize=wd, mgsize=mg, top_max_iter=2, n_threads=nthr * world)
    torch.cuda.synchronize()
    t_1 = (time.perf_counter() - t0) / 2
    # ... and its bottom half alone (the part the ranks split)
    tb = time.perf_counter()
    for _ in range(2):
        one.bottom(poses, coarse, fine, wd, mg, True, w_first=0, w_stride=1, n_threads=nthr * world)
    torch.cuda.synchronize()
    t_1b = (time.perf_counter() - tb) / 2
    one.close()
else:
    t_1b = 0.0
box = [ref, t_1, t_1b]
dist.broadcast_object_list(box, src=0)
ref, t_1, t_1b = box
et, er = synth.pose_errors(got["submap_poses"], ref["submap_poses"])
allp = [None] * world
dist.all_gather_object(allp, got["submap_poses"].tobytes())
same = all(b == allp[0] for b in allp)
nv = [None] * world
dist.all_gather_object(nv, [r["n_voxels"] for r in got["top_rounds"]])
sharded = len(got["submap_ids"]) > vxba.MAX_WIN
tot = np.sum(np.asarray(nv), axis=0).tolist() if sharded else nv[0]
print("rank %d hba_sharded: pose diff %.2e %.2e, same bits on all ranks %s, top voxels per round %s sum %s vs %s, submap sizes equal %s, edges %d %d vs %d %d, windows %s, pass %.4f s on %d ranks vs %.4f s on one (ratio %.3f); bottom half %.4f s vs %.4f s on one (bottom ratio %.3f), exchange %.4f s, top %.4f s" % (
    rank, et, er, same, nv[rank], tot, [r["n_voxels"] for r in ref["top_rounds"]], got["submap_sizes"] == ref["submap_sizes"],
    len(got["edges1"]), len(got["edges2"]), len(ref["edges1"]), len(ref["edges2"]), got["windows_of_rank"], t_n, world, t_1, t_n / max(t_1, 1e-9), ph["bottom"], t_1b, ph["bottom"] / max(t_1b, 1e-9), ph["exchange"], ph["top"]), flush=True)
ctx.clear(); ses.close()
dist.destroy_process_group()
