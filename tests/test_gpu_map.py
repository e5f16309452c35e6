"""The device-resident incremental local map (vxba_map_*, csrc/vxba_map.hip) against the CPU checkers: scan by
scan through cut_voxel_multi -> multi_recut (+ tras_opt straight into the GPU factor) -> Lidar_BA_Optimizer::damping_iter -> multi_margi
(reading the optimiser's cache on the device) -> ring shift, side by side with the oracle's LocalMap (pinned to the reference's OctoTree
by tests/test_ref_pin.py) and, where libref.so travelled with the snapshot, with the reference's own OctoTree.  After every stage: the
same leaves (ids, layers), the same flags and point counts, BIT-IDENTICAL window clusters, world clusters / plane records equal to
rounding, the same factor set handed to the BA, poses within 1e-7 (contract 1e-4).  Also against the committed golden of the reference."""
import os

import numpy as np
import pytest

from tests import _oracle as O
from tests import _ref
from tests.test_oracle_octree import PRM, point_vars, to_world
from voxel_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vx():
    from voxel_slam_amd import vxba
    vxba.load_library()
    return vxba


def by_id(lv):
    o = np.argsort(lv["node_id"], kind="stable")
    return {k: (v[o] if isinstance(v, np.ndarray) and v.shape[:1] == o.shape else v) for k, v in lv.items()}


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-300))


def same_structure(a, b, stage):
    assert np.array_equal(a["node_id"], b["node_id"]), stage
    for key in ("layer", "isexist", "is_plane", "has_sw", "n_point_fix", "in_slide", "last_num"):
        assert np.array_equal(np.asarray(a[key]).astype(np.int64), np.asarray(b[key]).astype(np.int64)), (stage, key)
    assert np.array_equal(a["n_points"], b["n_points"]), stage
    assert np.array_equal(a["pcr_add"][:, 9], b["pcr_add"][:, 9]) and np.array_equal(a["pcr_fix"][:, 9], b["pcr_fix"][:, 9]), stage


def checker_backends():
    out = [("oracle", O)]
    R = _ref.backend()
    if R is not None:
        out.append(("reference", R))
    return out


@pytest.mark.parametrize("backend", [n for n, _ in checker_backends()])
@pytest.mark.parametrize("S,win,pts,seed,max_points", [(9, 4, 12000, 6, 60), (12, 5, 8000, 11, 100), (10, 3, 20000, 3, 40)])
def test_device_map_evolves_like_the_checker(vx, backend, S, win, pts, seed, max_points):
    evolve(vx, backend, S, win, pts, seed, max_points)


def test_device_map_with_the_fix_point_pool_compacted_on_the_way(vx, monkeypatch):
    """The pool of marginalised points is a bump allocator that is compacted when its cursor has run far ahead of what is live
    (vxba_map_fix_pool, include/vxba.h).  With the threshold lowered to 2000 points the same sessions compact several times; every
    comparison against the checker -- among them the fix-point sums of the children of leaves subdivided AFTER a compaction -- must
    hold unchanged, and the pool must end smaller than the sum of what was ever allocated."""
    monkeypatch.setenv("VXBA_MAP_FIX_COMPACT_AT", "2000")
    total = 0
    for S, win, pts, seed, max_points in [(12, 5, 8000, 11, 100), (10, 3, 20000, 3, 40)]:
        mg = evolve(vx, "oracle", S, win, pts, seed, max_points)
        fp = mg.fix_pool()
        assert fp["compactions"] >= 2 and 0 < fp["cursor"] <= fp["capacity"], fp
        total += fp["compactions"]
    monkeypatch.delenv("VXBA_MAP_FIX_COMPACT_AT")
    mg = evolve(vx, "oracle", 9, 4, 12000, 6, 60)
    assert mg.fix_pool()["compactions"] == 0          # default threshold: 4M points


def evolve(vx, backend, S, win, pts, seed, max_points):
    B = dict(checker_backends())[backend]
    xyz, fp, poses_gt, _ = synth.make_scans(win_size=S, pts_per_scan=pts, seed=synth.MASTER_SEED + 900 + seed)
    rng = np.random.default_rng(seed)
    var = point_vars(xyz.shape[0], seed)
    kw = dict(PRM); kw["max_points"] = max_points
    mo, mg = B.LocalMapOracle(win_size=win, **kw), vx.LocalMap(win_size=win, **kw)
    fo, fg = B.Oracle(win), vx.LidarFactor(win)
    xo, xg = [], []
    win_count = windows = 0
    subdivided = capped = 0
    for k in range(S):
        pose = poses_gt[k].copy(); pose[9:12] += rng.normal(0, 0.01, 3)
        s = slice(fp[k], fp[k + 1])
        win_count += 1
        for m, f, xb in ((mo, fo, xo), (mg, fg, xg)):
            xb.append(pose.copy())
            f.clear()
            m.cut_voxel(win_count - 1, xyz[s], var[s], to_world(xb[-1], xyz[s]))
        a, b = by_id(mo.leaves()), mg.leaves()
        assert np.array_equal(a["node_id"], b["node_id"]) and np.array_equal(a["n_points"], b["n_points"])
        assert np.array_equal(a["pcrs_local"], b["pcrs_local"])                                   # running sums continued bit for bit
        if windows == 0:
            assert np.array_equal(a["pcr_add"], b["pcr_add"])
        assert rel(a["pcr_add"], b["pcr_add"]) < 1e-9 and rel(a["cov_add"], b["cov_add"]) < 1e-9
        mo.recut(win_count, np.stack(xo), fo)
        npush = mg.recut(win_count, np.stack(xg), fg)
        a, b = by_id(mo.leaves()), mg.leaves()
        same_structure(a, b, ("recut", k))
        assert np.array_equal(a["pcrs_local"], b["pcrs_local"])                                   # children of a subdivision: the reference's push order
        assert np.array_equal(a["opt_state"] >= 0, b["opt_state"] >= 0) and fo.size() == fg.size() == npush
        assert rel(a["pcr_add"], b["pcr_add"]) < 1e-9 and rel(a["pcr_fix"], b["pcr_fix"]) < 1e-9 and rel(a["cov_add"], b["cov_add"]) < 1e-9
        if windows == 0:
            assert np.array_equal(a["pcr_add"], b["pcr_add"]) and np.array_equal(a["pcr_fix"], b["pcr_fix"])
        pl = a["is_plane"]
        vb2 = np.sum((a["pcr_add"][pl, 6:9] / a["pcr_add"][pl, 9:10]) ** 2, axis=1, keepdims=True)
        assert np.all(np.abs(a["eig_val"][pl] - b["eig_val"][pl]) <= 1e-12 * (vb2 + 1.0))
        subdivided = max(subdivided, int((a["layer"] > 0).sum()))
        if win_count < win:
            continue
        windows += 1
        assert fo.size() > 100
        # the factor on the device holds what tras_opt would have pushed: same voxels (matched by node id), same content
        fa = np.nonzero(a["opt_state"] >= 0)[0]; fb = np.nonzero(b["opt_state"] >= 0)[0]
        assert np.array_equal(a["node_id"][fa], b["node_id"][fb])
        cl_dev = fg.read_clusters()
        assert np.array_equal(cl_dev[b["opt_state"][fb]], a["pcrs_local"][fa])
        oo = fo.damping_iter(np.stack(xo), max_iter=3, thd_num=2)
        gg = vx.Lidar_BA_Optimizer().damping_iter(np.stack(xg), fg, max_iter=3)
        assert np.array_equal(oo["trace"][:, 6:], gg["trace"][:, 6:])
        et, er = synth.pose_errors(gg["poses"], oo["poses"])
        assert et < 1e-7 and er < 1e-7, (et, er)
        mo.margi(win_count, oo["poses"], fo); mg.margi(win_count, gg["poses"], fg)
        mo.slide(1); mg.slide(1)
        xo[:] = [p for p in oo["poses"][1:]]; xg[:] = [p for p in gg["poses"][1:]]
        win_count -= 1
        a, b = by_id(mo.leaves()), mg.leaves()
        same_structure(a, b, ("margi", k))
        assert rel(a["pcr_add"], b["pcr_add"]) < 1e-9 and rel(a["pcr_fix"], b["pcr_fix"]) < 1e-9
        upd = a["is_plane"] & (a["last_num"] == a["pcr_add"][:, 9]) & (a["last_num"] > 0)
        sgn = np.sign(np.sum(a["normal"][upd] * b["normal"][upd], axis=1))
        assert np.allclose(a["normal"][upd], b["normal"][upd] * sgn[:, None], atol=1e-7) and np.allclose(a["center"][upd], b["center"][upd], atol=1e-8)
        # plane_var = cov(normal, centre): the normal's sign is the eigen-solver's choice, and the normal-centre blocks flip with it
        pvb = b["plane_var"][upd].copy()
        pvb[:, :3, 3:] *= sgn[:, None, None]; pvb[:, 3:, :3] *= sgn[:, None, None]
        assert np.allclose(a["radius"][upd], b["radius"][upd], rtol=1e-6) and rel(a["plane_var"][upd], pvb) < 1e-5
        ca, cb = mo.counts(), mg.counts()
        assert ca == cb, (ca, cb)
        capped = max(capped, int((a["pcr_fix"][:, 9] >= max_points).sum()))
    assert windows == S - win + 1 and subdivided > 50 and capped > 0
    return mg


def test_device_map_reproduces_the_reference_octree_golden(vx):
    """tests/golden/localmap_cycle.npz: leaf tables of the reference's OctoTree after every window (generated from libref.so)."""
    from tests.test_golden import LOCALMAP, check_localmap_window, localmap_replay
    g = np.load(LOCALMAP)
    n = 0
    opt = vx.Lidar_BA_Optimizer()
    for w, lv, lm in localmap_replay(g, vx.LocalMap, vx.LidarFactor, lambda f, xs: opt.damping_iter(xs, f, max_iter=3)):
        check_localmap_window(g, w, lv, lm, tol=1e-9)
        n += 1
    assert n == int(g["windows"])
    assert np.array_equal(lv["pcrs_local"], g["last_pcrs_local"])


def test_thread_num_quirk_and_errors(vx):
    """upstream's `if(g_size < thd_num) return;`: a scan touching fewer roots than threads is not pushed; recut / margi return early."""
    kw = dict(PRM)
    m = vx.LocalMap(win_size=3, thread_num=5, **kw)
    o = O.LocalMapOracle(win_size=3, thread_num=5, **kw)
    rng = np.random.default_rng(0)
    pts = rng.uniform(0.1, 0.9, (200, 3)) + np.array([3.0, 2.0, 0.0])       # one root voxel only
    var = point_vars(200, 1)
    pose = np.concatenate([np.eye(3).reshape(9), np.zeros(3)])
    f = vx.LidarFactor(3); fo = O.Oracle(3)
    for mm, ff in ((m, f), (o, fo)):
        mm.cut_voxel(0, pts, var, pts)
        mm.recut(1, pose[None], ff)
    assert m.counts() == o.counts() and f.size() == fo.size() == 0
    a, b = by_id(o.leaves()), m.leaves()
    assert np.array_equal(a["node_id"], b["node_id"]) and not b["has_sw"].any() and np.array_equal(a["pcr_add"], b["pcr_add"])
    with pytest.raises(vx.VxbaError):
        m.cut_voxel(0, pts, var, pts + 1e6)                                     # outside the +-32768 voxel range
    with pytest.raises(vx.VxbaError):
        m.recut(1, pose[None], vx.LidarFactor(4))                               # win_size mismatch


def test_whole_scan_cycle_on_the_device_matches_the_oracle_cycle(vx):
    """A scan cycle that moves only the raw scan up and the poses down: var_init -> lio_state_estimation against the plane map the
    tree exported on the device -> pvec_update (resident) -> cut_voxel on the resident scan -> recut + tras_opt into the factor -> BA ->
    margi on the factor's device cache -> ring shift -> plane export.  Free-running next to the same cycle on the oracle (whose odometry
    gets the full plane map re-built from its tree every scan): the incremental export must keep the two plane maps equivalent."""
    from tests.test_gpu_local_mapping_cycle import lio_leaf_args
    S, win, pts, seed = 10, 4, 20000, 7
    xyz, fp, poses_gt, _ = synth.make_scans(win_size=S, pts_per_scan=pts, seed=synth.MASTER_SEED + 900 + seed)
    rng = np.random.default_rng(seed)
    mo, mg = O.LocalMapOracle(win_size=win, **PRM), vx.LocalMap(win_size=win, **PRM)
    fo, fg = O.Oracle(win), vx.LidarFactor(win)
    ge = vx.LioEstimator(PRM["voxel_size"], PRM["max_layer"])
    xo, xg = [], []
    win_count = estimated = 0
    cov = np.eye(15) * 1e-4
    for k in range(S):
        s = slice(fp[k], fp[k + 1])
        scan32 = xyz[s].astype(np.float32)
        prior = np.concatenate([poses_gt[k][:9], poses_gt[k][9:12] + rng.normal(0, 0.02, 3), np.zeros(9), [0, 0, -9.8]])
        oe = O.LioOracle(PRM["voxel_size"], PRM["max_layer"])
        oe.var_init(scan32); ge.var_init(scan32)
        so, sg, co, cg = prior, prior, cov, cov
        lv = mo.leaves() if k else None
        if lv is not None and (lv["is_plane"] & (lv["last_num"] > 0)).sum() > 200:
            oe.map_update(*lio_leaf_args(lv))
            ro = oe.lio_state_estimation(prior, cov); rg = ge.lio_state_estimation(prior, cov)
            assert ro["iterations"] == rg["iterations"] and abs(ro["match_num"] - rg["match_num"]) <= 3 and ro["match_num"] > 0.3 * pts
            et, er = synth.pose_errors(rg["state"][None, :12], ro["state"][None, :12])
            assert et < 1e-7 and er < 1e-7, (k, et, er)
            so, co, sg, cg = ro["state"], ro["cov"], rg["state"], rg["cov"]
            estimated += 1
        pw_o, var_o = oe.pvec_update(so, co)
        ge.pvec_update(sg, cg, resident=True)
        pnt_body, _ = oe.read_points()
        win_count += 1
        xo.append(so[:12].copy()); xg.append(sg[:12].copy())
        fo.clear(); fg.clear()
        mo.cut_voxel(win_count - 1, pnt_body, var_o, pw_o)
        mg.cut_voxel_lio(win_count - 1, ge)
        mo.recut(win_count, np.stack(xo), fo); mg.recut(win_count, np.stack(xg), fg)
        a, b = by_id(mo.leaves()), mg.leaves()
        same_structure(a, b, ("recut", k))
        assert rel(a["pcrs_local"], b["pcrs_local"]) < 1e-9 and fo.size() == fg.size()
        if win_count >= win:
            oo = fo.damping_iter(np.stack(xo), max_iter=3, thd_num=2)
            gg = vx.Lidar_BA_Optimizer().damping_iter(np.stack(xg), fg, max_iter=3)
            assert np.array_equal(oo["trace"][:, 6:], gg["trace"][:, 6:])
            et, er = synth.pose_errors(gg["poses"], oo["poses"])
            assert et < 1e-7 and er < 1e-7, (k, et, er)
            mo.margi(win_count, oo["poses"], fo); mg.margi(win_count, gg["poses"], fg)
            mo.slide(1); mg.slide(1)
            xo[:] = [p for p in oo["poses"][1:]]; xg[:] = [p for p in gg["poses"][1:]]
            win_count -= 1
            same_structure(by_id(mo.leaves()), mg.leaves(), ("margi", k))
        n_exp = mg.export_planes(ge)
        assert n_exp > 0
    assert estimated >= 4


# ------------------------------------------------------------------------------------------------------------------------------------------
# release of far-away roots (vxba_map_release; voxelslam.cpp:1503-1523) on a long drive
# ------------------------------------------------------------------------------------------------------------------------------------------
def _corridor_scan(k, pts, rng):
    """Scan k of a drive along x (0.5 m per scan): floor, two walls and a cross wall every 10 m, sensor range 6 m; body frame = world - position."""
    x0 = 0.5 * k
    n = pts // 4
    u = rng.uniform(x0 - 6, x0 + 6, size=(3, n)); v = rng.uniform(0, 1, size=(4, n))
    floor = np.stack([u[0], -4 + 8 * v[0], np.full(n, -1.5)], 1)
    wall1 = np.stack([u[1], np.full(n, -4.0), -1.5 + 4 * v[1]], 1)
    wall2 = np.stack([u[2], np.full(n, 4.0), -1.5 + 4 * v[2]], 1)
    kx = np.round(rng.uniform(x0 - 6, x0 + 6, size=n) / 10.0) * 10.0 + 0.3
    cross = np.stack([kx, -4 + 3 * v[3], -1.5 + 4 * rng.uniform(0, 1, n)], 1)
    w = np.concatenate([floor, wall1, wall2, cross]) + rng.normal(0, 0.01, size=(4 * n, 3))
    p = np.array([x0, 0.0, 0.0])
    pose = np.concatenate([np.eye(3).reshape(-1), p])
    return w - p, pose


def test_release_of_far_roots_bounds_the_map_on_a_long_drive(vx):
    """2000 scans along a 1 km corridor, window 5, release every 100 scans of every root not marginalised for 60 journeys (metres here; 700
    upstream).  Two maps side by side, one released and one not: (1) the released map's leaves are, field for field and BIT FOR BIT, the
    leaves of the unreleased map under the roots it kept -- compaction is pure relocation, also of the fix-point pool; (2) its root count,
    node count and device memory stay bounded while the other map's grow with the distance; (3) the factor it hands to the BA is the same."""
    win, pts, S = 5, 4000, 2000
    rng = np.random.default_rng(77)
    kw = dict(PRM); kw["max_points"] = 60
    ma, mb = vx.LocalMap(win_size=win, **kw), vx.LocalMap(win_size=win, **kw)      # a: released, b: grows
    fa, fb = vx.LidarFactor(win), vx.LidarFactor(win)
    xs = []
    win_count = 0
    jour = 0.0
    hist = []
    released_total = 0
    for k in range(S):
        body, pose = _corridor_scan(k, pts, rng)
        var = point_vars(body.shape[0], k)
        xs.append(pose)
        win_count += 1
        wld = to_world(pose, body)
        na = nb = 0
        for m, f in ((ma, fa), (mb, fb)):
            f.clear()
            m.cut_voxel(win_count - 1, body, var, wld)
            n = m.recut(win_count, np.stack(xs), f)
            if m is ma:
                na = n
            else:
                nb = n
        if win_count >= win:
            # same root set under the window on both maps => the same factor voxels (ids are compared through the leaves below)
            assert na == nb > 0, (k, na, nb)
            for m, f in ((ma, fa), (mb, fb)):
                f.evaluate_only_residual(np.stack(xs))      # the cache margi reads (no BA here: poses are exact)
                m.set_journey(jour)
                m.margi(win_count, np.stack(xs), f)
                m.slide(1)
            xs = xs[1:]; win_count -= 1
            jour += 0.5
        if k % 100 == 99:
            r = ma.release(jour, 60)
            released_total += r["roots"]
            ca, cb = ma.counts(), mb.counts()
            hist.append((k, ca["roots"], cb["roots"], ma.device_bytes()["total"], mb.device_bytes()["total"], r["roots"], r["nodes"]))
            la, lb = ma.leaves(), mb.leaves()
            keep = np.isin(lb["node_id"] >> np.uint64(16), np.unique(la["node_id"] >> np.uint64(16)))
            assert keep.sum() == la["node_id"].size, (k, int(keep.sum()), la["node_id"].size)
            assert released_total == 0 or keep.sum() < lb["node_id"].size
            for key, va in la.items():
                if isinstance(va, np.ndarray) and va.shape[:1] == la["node_id"].shape:
                    assert np.array_equal(va, lb[key][keep]), (k, key)
    assert released_total > 500
    roots_a = np.array([h[1] for h in hist]); roots_b = np.array([h[2] for h in hist]); bytes_a = np.array([h[3] for h in hist]); bytes_b = np.array([h[4] for h in hist])
    assert roots_b[-1] > 5 * roots_a[-1]                                   # the unreleased map holds the whole corridor
    assert roots_a[5:].max() <= 1.3 * roots_a[5:].min() + 50              # steady state: what the last 60 m + the window hold
    assert bytes_a[5:].max() <= 1.6 * bytes_a[5:].min()                    # memory follows the map: bounded ...
    assert bytes_b[-1] > 2 * bytes_a[-1]                                   # ... while the other pool only grows
    fp = ma.fix_pool()
    assert fp["compactions"] >= 10 and fp["cursor"] <= fp["capacity"]
