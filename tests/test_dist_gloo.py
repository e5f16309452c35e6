"""N > 1 host logic on CPUs: world_size-2 gloo run of the voxel-sharded LM driver (product code: voxel_slam_amd.dist +
the C-ABI's host-only vxba_damping_iter_generic), with the CPU oracle as each rank's shard evaluator, against the
single-process oracle on the whole window."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, outdir):
    import torch.distributed as dist
    from tests import _oracle as O
    from voxel_slam_amd import dist as vdist, synth

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sc = synth.make_scene(win_size=5, pts_per_scan=5000, n_voxels=401, p_obs=0.9, fix_frac=0.2, seed=17,
                          rot_sigma_deg=0.2, trans_sigma=0.03)
    lo, hi = vdist.shard_bounds(sc.n_voxels, world, rank)
    f = O.Oracle(sc.win_size)
    f.push_voxels(sc.clusters[lo:hi], sc.fix[lo:hi], sc.coe[lo:hi])
    f.evaluate_only_residual(sc.poses_init)          # seed this shard's cache
    out = vdist.damping_iter_sharded(sc.win_size, sc.poses_init, f.acc_evaluate2, f.evaluate_only_residual, max_iter=4)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), poses=out["poses"], trace=out["trace"], resis=out["resis"], hess=out["hess"],
             lo=lo, hi=hi)
    dist.destroy_process_group()


def test_shard_bounds_follow_the_reference_rule():
    from voxel_slam_amd.dist import shard_bounds
    for V, world in [(401, 2), (50000, 8), (7, 3), (5, 5)]:
        cuts = [shard_bounds(V, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == V
        assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))


def test_two_rank_sharded_lm_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    from tests import _oracle as O
    from voxel_slam_amd import synth

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 200, 200, 401)
    # after the all-reduce both ranks hold the same system: identical decisions and poses, bit for bit
    assert np.array_equal(r0["poses"], r1["poses"]) and np.array_equal(r0["trace"], r1["trace"])

    sc = synth.make_scene(win_size=5, pts_per_scan=5000, n_voxels=401, p_obs=0.9, fix_frac=0.2, seed=17,
                          rot_sigma_deg=0.2, trans_sigma=0.03)
    f = O.Oracle(sc.win_size)
    f.push_voxels(sc.clusters, sc.fix, sc.coe)
    f.evaluate_only_residual(sc.poses_init)
    ref = f.damping_iter(sc.poses_init, max_iter=4, thd_num=2)
    assert np.array_equal(r0["trace"][:, 6:], ref["trace"][:, 6:])
    assert np.allclose(r0["trace"][:, :6], ref["trace"][:, :6], rtol=1e-8, atol=1e-12)
    et, er = synth.pose_errors(r0["poses"], ref["poses"])
    assert et < 1e-10 and er < 1e-10
    assert np.allclose(r0["hess"], ref["hess"], rtol=1e-10, atol=1e-10 * np.abs(ref["hess"]).max())


def test_generic_driver_single_process_equals_oracle_lm():
    """The host LM shell of the C ABI against the oracle's damping_iter on identical sweeps."""
    from tests import _oracle as O
    from voxel_slam_amd import synth, vxba
    sc = synth.make_scene(win_size=6, pts_per_scan=4000, n_voxels=300, seed=23, rot_sigma_deg=0.2, trans_sigma=0.03)
    fa = O.Oracle(sc.win_size); fa.push_voxels(sc.clusters, sc.fix, sc.coe); fa.evaluate_only_residual(sc.poses_init)
    fb = O.Oracle(sc.win_size); fb.push_voxels(sc.clusters, sc.fix, sc.coe); fb.evaluate_only_residual(sc.poses_init)
    ref = fa.damping_iter(sc.poses_init, max_iter=8, thd_num=2)
    got = vxba.damping_iter_generic(sc.win_size, sc.poses_init, fb.acc_evaluate2, fb.evaluate_only_residual, max_iter=8)
    assert np.array_equal(got["trace"][:, 6:], ref["trace"][:, 6:])
    assert np.allclose(got["trace"], ref["trace"], rtol=1e-9, atol=1e-13)
    assert np.allclose(got["poses"], ref["poses"], rtol=0, atol=1e-12)
    assert got["is_converge"] == ref["is_converge"]
