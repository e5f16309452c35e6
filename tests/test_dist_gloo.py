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


# ------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: hierarchical global BA over the ranks (dist.hierarchical_ba_sharded), the oracle standing in for each rank's GPU
# ------------------------------------------------------------------------------------------------------------------------------
HBA_CASE = dict(K=45, wdsize=6, mgsize=3, pts=3000)


def _hba_session():
    from voxel_slam_amd import synth, vxba
    c = HBA_CASE
    xyz, fp, poses, gt = synth.make_scans(win_size=c["K"], pts_per_scan=c["pts"], extent=24.0, noise=0.005, seed=synth.MASTER_SEED + 995,
                                          rot_sigma_deg=0.1, trans_sigma=0.02)
    clouds = [xyz[fp[i]:fp[i + 1]].astype(np.float32) for i in range(c["K"])]
    coarse = vxba.VoxelizeParams(voxel_size=2.0, max_layer=2, min_points=10, min_eigen_value=0.02, eigen_ratio=(1 / 9, 1 / 9, 1 / 9, 1 / 9))
    fine = vxba.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
    return clouds, poses, gt, coarse, fine


class _OracleOpt:
    def damping_iter(self, xs, f, max_iter=4):
        return f.damping_iter(xs, max_iter=max_iter, thd_num=2)


def _oracle_voxelize(W, shard=None):
    """The oracle's OctreeGBA walk; with shard = (index, count) only the root voxels that hash to `index` are kept -- what
    vxba_voxelize_push does on a rank's GPU with VoxelizeParams.sharded(index, count)."""
    from tests import _oracle as O
    from voxel_slam_amd import dist as vdist

    def go(xyz, fp, xs, params):
        r = O.voxelize(W, xyz, fp, xs, params.as_array())
        ids = r["node_id"]
        order = np.lexsort((ids, (ids & np.uint64(7)).astype(np.int64)))      # push order of the GPU path: by layer, ascending id inside a layer
        if shard is not None:
            keep = vdist.root_shard(ids[order] >> np.uint64(16), shard[1]) == shard[0]
            order = order[keep]
        n = order.size
        f = O.Oracle(W)
        if n:
            f.push_voxels(r["clusters"][order], np.zeros((n, 10)), np.ones(n), r["eig_val"][order], r["eig_vec"][order], r["merged"][order])
        return f, n
    return go


def _hba_worker(rank, world, port, outdir):
    import torch.distributed as dist
    from tests import _oracle as O
    from voxel_slam_amd import dist as vdist, hba

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    clouds, poses, gt, coarse, fine = _hba_session()
    c = HBA_CASE
    n_coll = {"packed": 0}

    class ShardedOpt:          # the LM shell over this rank's shard: sweeps by the oracle, sums by all-reduce (what the wide factor + RCCL do on a GPU)
        def damping_iter(self, xs, f, max_iter=4):
            W = xs.shape[0]
            zero = (np.zeros((6 * W, 6 * W)), np.zeros(6 * W), 0.0)

            def hess(x):
                n_coll["packed"] += 1
                return f.acc_evaluate2(x) if f.size() else zero

            return vdist.damping_iter_sharded(W, xs, hess, lambda x: f.evaluate_only_residual(x) if f.size() else 0.0, max_iter=max_iter)

    def bottom_refine(xyz, fp, xs):
        return hba.window_refine(xyz, fp, xs, coarse, fine, max_iter=1, optimizer=_OracleOpt(), voxelize=_oracle_voxelize(xs.shape[0]))   # the closing window has its own size

    def top_refine(xyz, fp, xs, si, sc):
        return hba.window_refine(xyz, fp, xs, coarse, fine, max_iter=2, optimizer=ShardedOpt(), voxelize=_oracle_voxelize(xs.shape[0], (si, sc)))

    out = vdist.hierarchical_ba_sharded(clouds, poses, coarse, fine, wdsize=c["wdsize"], mgsize=c["mgsize"], top_max_iter=2, bottom_refine=bottom_refine,
                                        top_refine=top_refine, downsample=O.down_sampling_voxel)
    np.savez(os.path.join(outdir, f"hba{rank}.npz"), submap_poses=out["submap_poses"], submap_sizes=np.asarray(out["submap_sizes"]),
             n_edges=np.asarray([len(out["edges1"]), len(out["edges2"])]), windows=np.asarray(out["windows_of_rank"]),
             top_voxels=np.asarray([r["n_voxels"] for r in out["top_rounds"]]), top_resis=np.asarray([r["resis"] for r in out["top_rounds"]]),
             edge2_v6=np.asarray([e["v6"] for e in out["edges2"]]), n_packed=n_coll["packed"])
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_hierarchical_ba_over_ranks_matches_single_process(tmp_path, world):
    """configs[4]'s schedule on `world` ranks: bottom-level windows round-robin (replicas), submaps all-gathered, the top-level window
    voxel-sharded by root-voxel hash with one all-reduce of the packed system per sweep.  Every rank must end on the same submap poses
    BIT FOR BIT; the shards must partition the single-process factor set exactly (voxel counts add up, round by round); poses and edge
    weights must equal the single-process run (hba.hierarchical_ba on the oracle) to round-off -- the shards sum in another order."""
    import torch.multiprocessing as mp
    from tests import _oracle as O
    from voxel_slam_amd import hba, synth

    port = _free_port()
    mp.spawn(_hba_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"hba{k}.npz") for k in range(world)]
    clouds, poses, gt, coarse, fine = _hba_session()
    c = HBA_CASE
    ref = hba.hierarchical_ba(clouds, poses, coarse, fine, wdsize=c["wdsize"], mgsize=c["mgsize"], top_max_iter=2, optimizer=_OracleOpt(),
                              voxelize=_oracle_voxelize, downsample=O.down_sampling_voxel)
    S = len(hba.windows(c["K"], c["wdsize"], c["mgsize"]))                                          # full windows + the closing one
    assert sorted(int(w) for k in range(world) for w in r[k]["windows"]) == list(range(S))          # every window ran exactly once
    for k in range(world):
        assert [int(w) % world for w in r[k]["windows"]] == [k] * len(r[k]["windows"])
        assert np.array_equal(r[k]["submap_poses"], r[0]["submap_poses"])                             # bit for bit across ranks
        assert np.array_equal(r[k]["submap_sizes"], np.asarray(ref["submap_sizes"]))                 # the bottom level is the single-process one
        assert np.array_equal(r[k]["n_edges"], r[0]["n_edges"]) and int(r[k]["n_packed"]) == int(r[0]["n_packed"])
        assert np.array_equal(r[k]["top_resis"], r[0]["top_resis"])
    # the shards partition the factor set of every top-level round
    assert np.array_equal(sum(r[k]["top_voxels"] for k in range(world)), np.asarray([x["n_voxels"] for x in ref["top_rounds"]]))
    assert all((r[k]["top_voxels"] > 0).all() for k in range(world)) or world == 8
    assert [int(x) for x in r[0]["n_edges"]] == [len(ref["edges1"]), len(ref["edges2"])]
    et, er = synth.pose_errors(r[0]["submap_poses"], ref["submap_poses"])
    assert et < 1e-9 and er < 1e-9, (et, er)
    assert np.allclose(r[0]["top_resis"], np.asarray([x["resis"] for x in ref["top_rounds"]]), rtol=1e-9)
    assert np.allclose(r[0]["edge2_v6"], np.asarray([e["v6"] for e in ref["edges2"]]), rtol=1e-6)


def test_root_shard_is_the_device_function():
    """dist.root_shard (numpy) against the definition in csrc/vxba_voxelize.h: ((root48 * 0x9E3779B97F4A7C15) mod 2^64 >> 32) mod count."""
    from voxel_slam_amd.dist import root_shard
    rng = np.random.default_rng(5)
    roots = rng.integers(0, 1 << 48, size=2000, dtype=np.uint64)
    for count in (2, 3, 8):
        want = np.array([(((int(x) * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)) >> 32) % count for x in roots])
        got = root_shard(roots, count)
        assert np.array_equal(got, want)
        assert np.bincount(got, minlength=count).min() > 2000 // count * 0.7          # spreads evenly
