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


def _scaling_worker(rank, world, port, outdir, scaling):
    """What bench.py does per rank for --scaling weak / strong, with the oracle standing in for the GPU shard."""
    import torch.distributed as dist
    from tests import _oracle as O
    from voxel_slam_amd import dist as vdist

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    kw = dict(win_size=4, pts_per_scan=3000, n_voxels=240, seed=91, rot_sigma_deg=0.1, trans_sigma=0.02)
    sc = vdist.rank_scene(kw, scaling, rank, world)
    f = O.Oracle(sc.win_size)
    f.push_voxels(O.build_clusters(sc.points_body, sc.cell_ptr).reshape(sc.win_size, sc.n_voxels, 10).transpose(1, 0, 2), sc.fix, sc.coe)   # from the re-bucketed points
    f.evaluate_only_residual(sc.poses_init)
    out = vdist.damping_iter_sharded(sc.win_size, sc.poses_init, f.acc_evaluate2, f.evaluate_only_residual, max_iter=3)
    np.savez(os.path.join(outdir, f"{scaling}{rank}.npz"), poses=out["poses"], trace=out["trace"], n_voxels=sc.n_voxels, clusters=sc.clusters,
             poses_init=sc.poses_init)
    dist.destroy_process_group()


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_rank_weak_and_strong_scaling_shards(tmp_path, scaling):
    """bench.py --scaling: strong = one window split by the reference's shard rule (the two shards ARE the window), weak = every rank
    its own window-sized set of voxels of one shared trajectory (the global window is the union).  Either way the sharded LM equals
    the single-process LM on the global window."""
    import torch.multiprocessing as mp
    from tests import _oracle as O
    from voxel_slam_amd import dist as vdist, synth

    world, port = 2, _free_port()
    mp.spawn(_scaling_worker, args=(world, port, str(tmp_path), scaling), nprocs=world, join=True)
    r = [np.load(tmp_path / f"{scaling}{k}.npz") for k in range(world)]
    assert np.array_equal(r[0]["poses"], r[1]["poses"]) and np.array_equal(r[0]["poses_init"], r[1]["poses_init"])
    kw = dict(win_size=4, pts_per_scan=3000, n_voxels=240, seed=91, rot_sigma_deg=0.1, trans_sigma=0.02)
    if scaling == "strong":
        full = vdist.rank_scene(kw, "strong", 0, 1)
        assert int(r[0]["n_voxels"]) + int(r[1]["n_voxels"]) == full.n_voxels == 240
        assert np.array_equal(np.concatenate([r[0]["clusters"], r[1]["clusters"]]), full.clusters)
        clusters, fix, coe = full.clusters, full.fix, full.coe
    else:
        assert int(r[0]["n_voxels"]) == int(r[1]["n_voxels"]) == 240 and not np.array_equal(r[0]["clusters"], r[1]["clusters"])
        parts = [vdist.rank_scene(kw, "weak", k, world) for k in range(world)]
        clusters = np.concatenate([p.clusters for p in parts]); fix = np.concatenate([p.fix for p in parts]); coe = np.concatenate([p.coe for p in parts])
    f = O.Oracle(4)
    f.push_voxels(clusters, fix, coe)
    f.evaluate_only_residual(r[0]["poses_init"])
    ref = f.damping_iter(r[0]["poses_init"], max_iter=3, thd_num=2)
    assert np.array_equal(r[0]["trace"][:, 6:], ref["trace"][:, 6:])
    et, er = synth.pose_errors(r[0]["poses"], ref["poses"])
    assert et < 1e-10 and er < 1e-10


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


def _eight_rank_worker(rank, world, port, outdir):
    """One rank of an 8-way voxel-sharded window (BASELINE configs[3]'s shape at 1/100 of its size): shard by the reference's rule, count the
    collectives the sharded LM issues."""
    import torch.distributed as dist
    from tests import _oracle as O
    from voxel_slam_amd import dist as vdist, synth

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sc = synth.make_scene(win_size=10, pts_per_scan=10_000, n_voxels=4003, seed=404, rot_sigma_deg=0.05, trans_sigma=0.02)
    lo, hi = vdist.shard_bounds(sc.n_voxels, world, rank)
    f = O.Oracle(sc.win_size)
    f.push_voxels(sc.clusters[lo:hi], sc.fix[lo:hi], sc.coe[lo:hi])
    f.evaluate_only_residual(sc.poses_init)
    counts = {"all_reduce": 0, "elements": []}
    real = dist.all_reduce

    def counting(t, *a, **k):
        counts["all_reduce"] += 1
        counts["elements"].append(int(t.numel()))
        return real(t, *a, **k)
    dist.all_reduce = counting
    try:
        out = vdist.damping_iter_sharded(sc.win_size, sc.poses_init, f.acc_evaluate2, f.evaluate_only_residual, max_iter=3)
    finally:
        dist.all_reduce = real
    np.savez(os.path.join(outdir, f"r{rank}.npz"), poses=out["poses"], trace=out["trace"], lo=lo, hi=hi, n_coll=counts["all_reduce"],
             elements=np.asarray(counts["elements"]))
    dist.destroy_process_group()


def test_eight_rank_sharded_window_matches_single_process(tmp_path):
    """world_size 8 (the node BASELINE configs[3] / configs[4] are quoted on): shard bounds by the reference's rule int(part i) .. int(part (i+1))
    (voxel_map.hpp:318-321), every rank ends on the same poses bit for bit, the result equals the single-process LM on the whole window,
    and the host-driven loop exchanges exactly one packed [Hess | JacT | residual] buffer per Hessian sweep plus one scalar per residual
    sweep (the device-resident loop folds the two into ONE all-reduce of (6W)^2 + 6W + 2 doubles per iteration; that count is
    checked on the GPU, tests/test_gpu_parity.py::test_two_voxel_shards_with_a_real_cross_shard_sum)."""
    import torch.multiprocessing as mp
    from tests import _oracle as O
    from voxel_slam_amd import synth

    world, port = 8, _free_port()
    mp.spawn(_eight_rank_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(world)]
    V, W = 4003, 10
    part = 1.0 * V / world
    for k in range(world):
        assert (int(r[k]["lo"]), int(r[k]["hi"])) == (int(part * k), int(part * (k + 1)) if k + 1 < world else V)
        assert np.array_equal(r[k]["poses"], r[0]["poses"]) and np.array_equal(r[k]["trace"], r[0]["trace"])
        assert int(r[k]["n_coll"]) == int(r[0]["n_coll"])
    n_hess = int(r[0]["trace"][:, 7].sum())                      # iterations that recomputed the Hessian
    n_iter = r[0]["trace"].shape[0]
    el = r[0]["elements"]
    n = 6 * W
    assert (el == n * n + n + 1).sum() == n_hess and (el == 1).sum() == n_iter     # the packed buffer carries residual1 with it
    assert int(r[0]["n_coll"]) == n_hess + n_iter
    sc = synth.make_scene(win_size=10, pts_per_scan=10_000, n_voxels=V, seed=404, rot_sigma_deg=0.05, trans_sigma=0.02)
    f = O.Oracle(W)
    f.push_voxels(sc.clusters, sc.fix, sc.coe)
    f.evaluate_only_residual(sc.poses_init)
    ref = f.damping_iter(sc.poses_init, max_iter=3, thd_num=4)
    assert np.array_equal(r[0]["trace"][:, 6:], ref["trace"][:, 6:])
    et, er = synth.pose_errors(r[0]["poses"], ref["poses"])
    assert et < 1e-10 and er < 1e-10
