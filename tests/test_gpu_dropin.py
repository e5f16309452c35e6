"""The drop-in, proven on the reference's own text: oracle/_ref/libdropin.so is the reference's voxel_map.hpp / loop_refine.hpp with
the ONE edit INTEGRATION.md describes (the block LidarFactor ... LI_BA_OptimizerGravity, voxel_map.hpp:108-864, replaced by
`#include "vxba_voxel_map.hpp"`; done on a build-time copy, `make -C oracle dropin`), compiled with the same driver as libref.so
and linked against libvxba.so.  So OctoTree::tras_opt pushes into the MI355X factor, OctoTree::margi reads pcr_adds / eig_values /
eig_vectors back from it (from five threads), OctreeGBA_multi_recut copies and concatenates factors with the reference's insert
idiom, and the three optimizers are called with the reference's signatures -- and everything must come out as from the untouched
reference (libref.so) on the same inputs."""
import numpy as np
import pytest

from tests import _ref
from tests.test_oracle_octree import PRM, point_vars, to_world
from voxel_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def RD():
    R, D = _ref.backend(), _ref.dropin()
    if R is None or D is None:
        pytest.skip("oracle/_ref/libref.so / libdropin.so not available")
    assert "vxba_voxel_map.hpp" in D.BACKEND_NAME
    return R, D


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-300))


@pytest.mark.parametrize("kw,iters", [
    (dict(win_size=5, pts_per_scan=1500, n_voxels=96, seed=9001, rot_sigma_deg=0.2, trans_sigma=0.03), 4),
    (dict(win_size=10, pts_per_scan=2500, n_voxels=130, p_obs=0.7, fix_frac=0.3, seed=9002, rot_sigma_deg=0.1, trans_sigma=0.02), 4),
    (dict(win_size=6, pts_per_scan=3000, n_voxels=200, seed=4242, rot_sigma_deg=2.5, trans_sigma=0.25), 8),
    (dict(win_size=5, pts_per_scan=20_000, n_voxels=5_000, seed=synth.MASTER_SEED + 1), 3),
])
def test_lidar_ba_optimizer_call_site(RD, kw, iters):
    """LidarFactor::push_voxel x V, evaluate_only_residual, acc_evaluate2, Lidar_BA_Optimizer::damping_iter(xs, voxhess, &hess, resis, up, true)"""
    R, D = RD
    sc = synth.make_scene(**kw)
    coe = np.linspace(0.5, 1.5, sc.n_voxels)
    out = {}
    for name, B in (("ref", R), ("dropin", D)):
        f = B.Oracle(sc.win_size)
        f.push_voxels(sc.clusters, sc.fix, coe)
        assert f.size() == sc.n_voxels
        r0 = f.evaluate_only_residual(sc.poses_init)
        H, J, r = f.acc_evaluate2(sc.poses_init)
        lm = f.damping_iter(sc.poses_init, max_iter=iters, thd_num=2)
        ev, U, m = f.read_cache()
        out[name] = (r0, H, J, r, lm, ev, m)
    a, b = out["dropin"], out["ref"]
    assert np.isclose(a[0], b[0], rtol=1e-10) and rel(a[1], b[1]) < 1e-10 and rel(a[2], b[2]) < 1e-10 and np.isclose(a[3], b[3], rtol=1e-10)
    ta, tb = a[4]["trace"], b[4]["trace"]
    assert ta.shape == tb.shape and np.array_equal(ta[:, 6:], tb[:, 6:])       # the is_display line of every iteration, parsed by the hook
    assert np.allclose(ta[:, :2], tb[:, :2], rtol=1e-9) and np.allclose(ta[:, 2:4], tb[:, 2:4], rtol=1e-6)
    et, er = synth.pose_errors(a[4]["poses"], b[4]["poses"])
    assert et < 1e-7 and er < 1e-7 and a[4]["is_converge"] == b[4]["is_converge"]
    assert rel(a[4]["hess"], b[4]["hess"]) < 1e-8 and np.allclose(a[4]["resis"], b[4]["resis"], rtol=1e-9)
    assert rel(a[6], b[6]) < 1e-9


@pytest.mark.parametrize("W,V,pts", [(5, 500, 6000), (10, 1500, 20000)])
def test_li_optimizer_call_sites(RD, W, V, pts):
    """LI_BA_Optimizer::damping_iter(x_buf, voxhess, imu_pre_buf, &hess) and LI_BA_OptimizerGravity::damping_iter(x_buf, voxhess,
    imu_pre_buf, resis, &hess, n) with the reference's IMUST / IMU_PRE structs."""
    R, D = RD
    sc = synth.make_scene(win_size=W, pts_per_scan=pts, n_voxels=V, seed=500 + W)
    iw = synth.make_imu(sc, seed=501 + W)
    blobs = R.imu_preintegrate(iw.samples, iw.noise_meas, iw.noise_walk, iw.states_init[0, 15:18], iw.states_init[0, 18:21])
    res = {}
    for name, B in (("ref", R), ("dropin", D)):
        f = B.Oracle(W); f.push_voxels(sc.clusters, sc.fix, sc.coe); f.evaluate_only_residual(sc.poses_init)
        H, J, r = B.li_divide_thread(f, iw.states_init, blobs, 5, 1e-4)
        r2 = B.li_only_residual(f, iw.states_init, blobs, 5, 1e-4)
        li = B.li_damping_iter(f, iw.states_init, blobs, max_iter=3, imu_coef=1e-4)
        f.evaluate_only_residual(sc.poses_init)
        lg = B.li_damping_iter_gravity(f, iw.states_init, blobs, max_iter=2, imu_coef=1e-4)
        res[name] = (H, J, r, r2, li, lg)
    a, b = res["dropin"], res["ref"]
    assert rel(a[0], b[0]) < 1e-6 and rel(a[1], b[1]) < 1e-6 and np.isclose(a[2], b[2], rtol=1e-8) and np.isclose(a[3], b[3], rtol=1e-8)
    for k in (4, 5):
        et, er = synth.pose_errors(a[k]["states"][:, :12], b[k]["states"][:, :12])
        assert et < 1e-7 and er < 1e-7 and np.allclose(a[k]["states"][:, 12:], b[k]["states"][:, 12:], atol=1e-6)
        assert np.allclose(a[k]["imus"][:, 67:79], b[k]["imus"][:, 67:79], atol=1e-7) and rel(a[k]["hess"], b[k]["hess"]) < 1e-5
    assert np.allclose(a[5]["resis"], b[5]["resis"], rtol=1e-8)


def by_id(lv):
    o = np.argsort(lv["node_id"], kind="stable")
    return {k: (v[o] if isinstance(v, np.ndarray) and v.shape[:1] == o.shape else v) for k, v in lv.items()}


def test_local_mapping_cycle_on_the_reference_octree(RD):
    """voxelslam.cpp:1609-1700 scan by scan: cut_voxel_multi -> multi_recut (OctoTree::tras_opt -> LidarFactor::push_voxel) ->
    Lidar_BA_Optimizer::damping_iter -> multi_margi (OctoTree::margi reads vox_opt.pcr_adds[opt_state] ...) -> ring shift.  The tree is
    the reference's in both runs; the factor and the optimizer are the reference's in one and the MI355X drop-in in the other."""
    R, D = RD
    S, win, pts, seed = 9, 4, 12000, 6
    xyz, fp, poses_gt, _ = synth.make_scans(win_size=S, pts_per_scan=pts, seed=synth.MASTER_SEED + 900 + seed)
    rng = np.random.default_rng(seed)
    var = point_vars(xyz.shape[0], seed)
    kw = dict(PRM); kw["max_points"] = 60
    maps = [R.LocalMapOracle(win_size=win, **kw), D.LocalMapOracle(win_size=win, **kw)]
    facs = [R.Oracle(win), D.Oracle(win)]
    xbs = [[], []]
    win_count = windows = 0
    for k in range(S):
        pose = poses_gt[k].copy(); pose[9:12] += rng.normal(0, 0.01, 3)
        s = slice(fp[k], fp[k + 1])
        win_count += 1
        for m, f, xb in zip(maps, facs, xbs):
            xb.append(pose.copy())
            f.clear()
            m.cut_voxel(win_count - 1, xyz[s], var[s], to_world(xb[-1], xyz[s]))
            m.recut(win_count, np.stack(xb), f)
        assert facs[0].size() == facs[1].size()
        if win_count < win:
            continue
        windows += 1
        outs = [f.damping_iter(np.stack(xb), max_iter=3, thd_num=2) for f, xb in zip(facs, xbs)]
        assert np.array_equal(outs[0]["trace"][:, 6:], outs[1]["trace"][:, 6:])
        et, er = synth.pose_errors(outs[1]["poses"], outs[0]["poses"])
        assert et < 1e-7 and er < 1e-7
        for m, f, xb, out in zip(maps, facs, xbs, outs):
            m.margi(win_count, out["poses"], f)
            m.slide(1)
            xb[:] = [p for p in out["poses"][1:]]
        win_count -= 1
        a, b = by_id(maps[0].leaves()), by_id(maps[1].leaves())
        assert np.array_equal(a["node_id"], b["node_id"])
        for key in ("isexist", "is_plane", "has_sw", "n_point_fix", "n_points", "in_slide", "last_num", "layer"):
            assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a["pcr_add"][:, 9], b["pcr_add"][:, 9]) and np.array_equal(a["pcr_fix"][:, 9], b["pcr_fix"][:, 9])
        assert rel(a["pcr_add"], b["pcr_add"]) < 1e-9 and rel(a["pcr_fix"], b["pcr_fix"]) < 1e-9
        upd = a["is_plane"] & (a["last_num"] == a["pcr_add"][:, 9]) & (a["last_num"] > 0)
        sgn = np.sign(np.sum(a["normal"][upd] * b["normal"][upd], axis=1))
        assert np.allclose(a["normal"][upd], b["normal"][upd] * sgn[:, None], atol=1e-7) and np.allclose(a["center"][upd], b["center"][upd], atol=1e-8)
    assert windows == 6


def test_octree_gba_multi_recut_on_the_dropin_factor(RD):
    """voxelslam.cpp:2374-2384: OctreeGBA::cut_voxel, OctreeGBA_multi_recut(oct_map, voxhess, 2) -- which copy-constructs per-thread factors
    and concatenates them with X.insert(X.end(), other.X.begin(), other.X.end()) -- then the factor content read back through the
    reference's member names (plvec_voxels[a][i], eig_values[a], eig_vectors[a], pcr_adds[a])."""
    R, D = RD
    W = 5
    xyz, fp, poses, _ = synth.make_scans(win_size=W, pts_per_scan=6000, seed=synth.MASTER_SEED + 812)
    xyz = xyz.astype(np.float32).astype(np.float64)
    params = np.array([1.0, 2, 10, 0.02, 1 / 16, 1 / 16, 1 / 9, 1 / 9, 0.12, 0, 0, 0, 0, 2], dtype=np.float64)
    r, d = R.voxelize(W, xyz, fp, poses, params), D.voxelize(W, xyz, fp, poses, params)
    assert r["node_id"].shape[0] == d["node_id"].shape[0] > 100
    key = lambda v: np.lexsort((v["merged"][:, 8], v["merged"][:, 7], v["merged"][:, 6], v["merged"][:, 9]))
    ir, idd = key(r), key(d)
    assert np.array_equal(r["clusters"][ir], d["clusters"][idd]) and np.array_equal(r["merged"][ir], d["merged"][idd])
    assert np.array_equal(r["eig_val"][ir], d["eig_val"][idd]) and np.array_equal(r["eig_vec"][ir], d["eig_vec"][idd])
