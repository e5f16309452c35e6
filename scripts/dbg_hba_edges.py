"""Development: degenerate sessions through the hierarchical pass below the C ABI against the Python orchestration (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from voxel_slam_amd import hba, synth, vxba

coarse = vxba.VoxelizeParams(voxel_size=2.0, max_layer=2, min_points=10, min_eigen_value=0.02, eigen_ratio=(1 / 9, 1 / 9, 1 / 9, 1 / 9))
fine = vxba.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))


def session(K, seed):
    xyz, fp, poses, _ = synth.make_scans(win_size=K, pts_per_scan=4000, extent=24.0, noise=0.005, seed=synth.MASTER_SEED + seed, rot_sigma_deg=0.1, trans_sigma=0.02)
    return [xyz[fp[i]:fp[i + 1]].astype(np.float32) for i in range(K)], poses


def run(name, clouds, poses, wd, mg, threads=3):
    try:
        py = hba.hierarchical_ba(clouds, poses, coarse, fine, wdsize=wd, mgsize=mg, top_max_iter=2)
    except Exception as e:   # noqa: BLE001
        py = None; print(name, "python path raised:", repr(e)[:200])
    try:
        ses = vxba.HbaSession(); ses.add_keyframes(clouds)
        got = ses.run_pass(poses, coarse, fine, wdsize=wd, mgsize=mg, top_max_iter=2, n_threads=threads)
        ses.close()
    except Exception as e:   # noqa: BLE001
        got = None; print(name, "C pass raised:", repr(e)[:200])
    if py is not None and got is not None:
        et, er = synth.pose_errors(got["submap_poses"], py["submap_poses"])
        print(name, "ok: submaps", len(got["submap_ids"]), "sizes differ by", int(np.max(np.abs(np.asarray(got["submap_sizes"]) - np.asarray(py["submap_sizes"])))),
              "edges", len(got["edges1"]), len(py["edges1"]), len(got["edges2"]), len(py["edges2"]), "pose diff %.2e m %.2e rad" % (et, er))


cl, ps = session(25, 1)
cl[7] = np.zeros((0, 3), np.float32)
run("one empty keyframe", cl, ps, 10, 5)
cl, ps = session(30, 2)
for i in range(10, 20):
    cl[i] = np.zeros((0, 3), np.float32)
run("a window of empty keyframes", cl, ps, 10, 5)
cl, ps = session(10, 3)
run("K == wdsize (one submap)", cl, ps, 10, 5)
cl, ps = session(12, 4)
run("wdsize 2", cl, ps, 2, 1)
cl, ps = session(26, 5)
run("stride larger than the window", cl, ps, 4, 7)
cl, ps = session(20, 6)
run("eight threads, four windows", cl, ps, 5, 5, threads=8)
