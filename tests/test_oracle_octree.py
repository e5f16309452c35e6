"""The incremental local map of the oracle (oracle/vxo_octree.hpp: OctoTree + cut_voxel_multi / multi_recut / multi_margi, SURVEY 8 row
f2) pinned through what the algorithm must conserve and through the batch construction that is already pinned:
 - all scans cut first, one recut (motion_init's build) == the batch voxeliser with OctoTree's criteria, clusters bit for bit;
 - cov_add of an undivided root == the sum of Bf_var over its points; plane records == plane_update of the leaf's own quantities;
 - through a sliding window: point counts conserved leaf by leaf, world cluster == fix + sum of transformed window clusters,
   subdivision is sticky, the slot ring returns after win_size shifts, the max_points cap stops the fix cluster and frees its points;
 - tras_opt hands the optimiser exactly the leaves that carry a plane; with the BA in the loop margi adopts the optimiser's cache;
 - upstream's `if(g_size < thd_num) return` quirk.
No reference vectors exist for this path (parity unpinned)."""
import numpy as np
import pytest

from tests import _oracle as O
from voxel_slam_amd import synth

PRM = dict(voxel_size=1.0, max_layer=2, min_point=(20, 20, 15, 10), min_eigen_value=0.02, plane_eigen_value_thre=(0.25, 0.25, 0.25, 0.25))


def to_world(pose, pnt):
    """R * p + t with the oracle's own association (vxo_linalg.hpp operator*), so that world points agree bit for bit."""
    R = pose[:9].reshape(3, 3).T
    x, y, z = pnt[:, 0], pnt[:, 1], pnt[:, 2]
    return np.stack([(R[r, 0] * x + R[r, 1] * y + R[r, 2] * z) + pose[9 + r] for r in range(3)], axis=1)


def point_vars(n, seed):
    rng = np.random.default_rng(seed)
    M = rng.normal(size=(n, 3, 3)) * 0.01
    return M @ np.transpose(M, (0, 2, 1)) + np.eye(3) * 1e-5


def transform_cluster(cl, pose):
    R = pose[:9].reshape(3, 3).T; t = pose[9:12]
    P = np.array([[cl[0], cl[1], cl[2]], [cl[1], cl[3], cl[4]], [cl[2], cl[4], cl[5]]]); v = cl[6:9]; N = cl[9]
    Rv = R @ v
    Pw = R @ P @ R.T + np.outer(Rv, t) + np.outer(t, Rv) + N * np.outer(t, t)
    return np.array([Pw[0, 0], Pw[0, 1], Pw[0, 2], Pw[1, 1], Pw[1, 2], Pw[2, 2], *(Rv + N * t), N])


@pytest.mark.parametrize("W,pts,seed", [(4, 6000, 1), (7, 8000, 2)])
def test_all_scans_then_one_recut_equals_the_batch_build(W, pts, seed):
    xyz, fp, poses, _ = synth.make_scans(win_size=W, pts_per_scan=pts, seed=synth.MASTER_SEED + 800 + seed)
    params = np.array([1.0, 2, 20, 0.02, 0.25, 0.25, 0.25, 0.25, 0.12, 20, 20, 15, 10, 0], dtype=np.float64)
    ref = O.voxelize(W, xyz, fp, poses, params)
    m = O.LocalMapOracle(win_size=W, **PRM)
    var = point_vars(xyz.shape[0], seed)
    pw_all = np.zeros_like(xyz)
    for i in range(W):
        s = slice(fp[i], fp[i + 1])
        pw_all[s] = to_world(poses[i], xyz[s])
        m.cut_voxel(i, xyz[s], var[s], pw_all[s])
    f = O.Oracle(W)
    m.recut(W, poses, f)
    lv = m.leaves()
    fac = lv["opt_state"] >= 0
    assert f.size() == fac.sum() == ref["node_id"].shape[0] > 50
    order = np.argsort(lv["node_id"][fac])
    assert np.array_equal(lv["node_id"][fac][order], ref["node_id"])
    assert np.array_equal(lv["pcrs_local"][fac][order], ref["clusters"])          # body-frame clusters, bit for bit
    assert np.array_equal(lv["pcr_add"][fac][order], ref["merged"])
    assert np.array_equal(lv["eig_val"][fac][order], ref["eig_val"])
    assert np.array_equal(np.sort(lv["opt_state"][fac]), np.arange(fac.sum()))     # tras_opt numbered them in push order
    ev, U, merged = f.read_cache()
    assert np.array_equal(merged[lv["opt_state"][fac]], lv["pcr_add"][fac])
    # cov_add of a root that was never divided: sum of Bf_var over its points in push order (frame by frame, scan order inside)
    root = (lv["layer"] == 0) & lv["has_sw"]
    assert root.sum() > 10
    cell = np.floor(pw_all / 1.0).astype(np.int64)
    key = ((cell[:, 0] + 32768).astype(np.uint64) << np.uint64(32)) | ((cell[:, 1] + 32768).astype(np.uint64) << np.uint64(16)) | (cell[:, 2] + 32768).astype(np.uint64)
    for a in np.nonzero(root)[0][:12]:
        sel = np.nonzero(key == (lv["node_id"][a] >> np.uint64(16)))[0]
        assert sel.size == lv["pcr_add"][a, 9]
        ca = O.cov_add_build(pw_all[sel], var[sel], np.array([0, sel.size]))[0]
        assert np.allclose(lv["cov_add"][a], ca, rtol=1e-12, atol=1e-18)
        assert np.array_equal(O.build_clusters(np.ascontiguousarray(pw_all[sel]), np.array([0, sel.size]))[0], lv["pcr_add"][a])


def run_stream(S, win, pts, seed, max_points=100, with_ba=False, perturb=0.0):
    """The local-mapping loop of voxelslam.cpp:1592-1700 on S scans: yields the state after every margi."""
    xyz, fp, poses_gt, _ = synth.make_scans(win_size=S, pts_per_scan=pts, seed=synth.MASTER_SEED + 900 + seed)
    rng = np.random.default_rng(seed)
    var = point_vars(xyz.shape[0], seed)
    m = O.LocalMapOracle(win_size=win, max_points=max_points, **PRM)
    f = O.Oracle(win)
    x_buf, scans = [], []
    win_count = 0
    for k in range(S):
        pose = poses_gt[k].copy()
        if perturb:
            pose[9:12] += rng.normal(0, perturb, 3)
        s = slice(fp[k], fp[k + 1])
        win_count += 1
        x_buf.append(pose); scans.append(k)
        f.clear()
        m.cut_voxel(win_count - 1, xyz[s], var[s], to_world(pose, xyz[s]))
        xs = np.stack(x_buf)
        m.recut(win_count, xs, f)
        before = m.leaves()
        if win_count >= win:
            if with_ba and f.size() > 0:
                out = f.damping_iter(xs, max_iter=3, thd_num=2)
                xs = out["poses"]; x_buf = [p for p in xs]
            m.margi(win_count, xs, f)
            yield dict(map=m, factor=f, leaves=m.leaves(), before=before, x_buf=np.stack(x_buf), scans=list(scans), step=k, gt=poses_gt)
            m.slide(1)
            x_buf.pop(0); scans.pop(0)
            win_count -= 1


def internal_prefixes(node_id, layer):
    """ids of the subdivided ancestors of every leaf (root / layer-1 prefixes)."""
    out = set()
    for i, l in zip(node_id.tolist(), layer.tolist()):
        root = i >> 16; path = (i >> 7) & 0x1FF
        if l >= 1: out.add((root, 0, 0))
        if l >= 2: out.add((root, 1, path >> 6))
    return out


def test_sliding_window_conserves_points_and_clusters():
    win = 5
    prev_internal = set()
    prev_ids = None
    steps = born_with_fix = 0
    for st in run_stream(S=12, win=win, pts=4000, seed=3):
        lv = st["leaves"]; xs = st["x_buf"]
        steps += 1
        # children created by a subdivision of a leaf that already held marginalised points start with their share of them (fix_divide),
        # and their world cluster is again fix + window
        if prev_ids is not None:
            for a in np.nonzero((lv["layer"] > 0) & (lv["pcr_fix"][:, 9] > 0))[0]:
                if int(lv["node_id"][a]) not in prev_ids:
                    born_with_fix += 1
                    acc = lv["pcr_fix"][a].copy()
                    for i in range(win):
                        if lv["pcrs_local"][a, i, 9] > 0:
                            acc += transform_cluster(lv["pcrs_local"][a, i], xs[i])
                    assert np.allclose(acc, lv["pcr_add"][a], rtol=1e-9, atol=1e-9)
        prev_ids = set(lv["node_id"].tolist())
        # leaf by leaf: points in the world cluster = marginalised points + points still in the window
        assert np.array_equal(lv["pcr_add"][:, 9], lv["pcr_fix"][:, 9] + lv["pcrs_local"][:, :, 9].sum(axis=1))
        # slot 0 has just been marginalised in every leaf margi visited
        live = lv["in_slide"] & lv["has_sw"] & lv["isexist"]
        assert live.sum() > 50 and np.all(lv["pcrs_local"][live][:, 0, 9] == 0)
        # world cluster = fix + sum of window clusters moved by the window poses (slots 1.. hold scans 1..)
        for a in np.nonzero(live)[0][:40]:
            acc = lv["pcr_fix"][a].copy()
            for i in range(1, win):
                if lv["pcrs_local"][a, i, 9] > 0:
                    acc += transform_cluster(lv["pcrs_local"][a, i], xs[i])
            assert np.allclose(acc, lv["pcr_add"][a], rtol=1e-9, atol=1e-9)
        # points kept for a later subdivision exist only above the finest layer, one per point of the slot's cluster
        fine = lv["layer"] == 2
        assert np.all(lv["n_points"][fine] == 0) and np.all(lv["n_point_fix"][fine] == 0)
        coarse = (lv["layer"] < 2) & lv["has_sw"]
        assert np.array_equal(lv["n_points"][coarse], lv["pcrs_local"][coarse][:, :, 9].astype(np.int64))
        # subdivision is sticky
        now = internal_prefixes(lv["node_id"], lv["layer"])
        assert prev_internal <= now
        prev_internal = now
        # voxels whose content is entirely marginalised left the slide map and gave their window back
        gone = ~lv["in_slide"]
        assert not lv["has_sw"][gone].any()
        # plane records are plane_update of the leaf's own cluster / eigen-decomposition / cov_add at the time of the update
        fresh = live & lv["is_plane"] & (lv["last_num"] == lv["pcr_add"][:, 9]) & (lv["pcr_fix"][:, 9] < 100)
        if fresh.any():
            pl = O.plane_update(lv["pcr_add"][fresh], lv["eig_val"][fresh], lv["eig_vec"][fresh], lv["cov_add"][fresh])
            assert np.allclose(pl["center"], lv["center"][fresh], rtol=1e-14) and np.allclose(pl["normal"], lv["normal"][fresh], atol=1e-15)
            assert np.allclose(pl["plane_var"], lv["plane_var"][fresh], rtol=1e-9, atol=1e-16) and np.array_equal(pl["radius"].astype(np.float32), lv["radius"][fresh].astype(np.float32))
        assert st["map"].counts()["mp0"] == (steps - 1) % win
    assert steps == 8 and born_with_fix > 100


def test_tras_opt_hands_over_the_planes_and_margi_adopts_the_optimisers_cache():
    for st in run_stream(S=9, win=4, pts=5000, seed=4, with_ba=True, perturb=0.01):
        b = st["before"]; f = st["factor"]
        fac = b["opt_state"] >= 0
        expect = b["isexist"] & b["is_plane"] & b["has_sw"] & (b["eig_val"][:, 0] / np.where(b["eig_val"][:, 1] != 0, b["eig_val"][:, 1], 1) <= 0.12)
        assert f.size() == fac.sum() > 30 and np.array_equal(fac, expect)
        lv = st["leaves"]
        ev, U, merged = f.read_cache()
        ids_before = {int(i): int(o) for i, o in zip(b["node_id"][fac], b["opt_state"][fac])}
        hit = 0
        for a, nid in enumerate(lv["node_id"].tolist()):
            if nid in ids_before and lv["pcr_fix"][a, 9] < 100:
                o = ids_before[nid]
                # pcr_add = the optimiser's merged cluster (slot 0 stays inside it: it moved into pcr_fix), eigen-decomposition likewise
                assert np.array_equal(lv["pcr_add"][a], merged[o]) and np.array_equal(lv["eig_val"][a], ev[o])
                hit += 1
        assert hit > 30 and np.all(lv["opt_state"] == -1)
        assert np.array_equal(lv["pcr_add"][:, 9], lv["pcr_fix"][:, 9] + lv["pcrs_local"][:, :, 9].sum(axis=1))


def test_max_points_caps_the_fix_cluster():
    cap = 40
    seen_capped = released = 0
    capped_before = {}
    for st in run_stream(S=14, win=4, pts=6000, seed=5, max_points=cap):
        lv = st["leaves"]
        capped = lv["pcr_fix"][:, 9] >= cap
        seen_capped = max(seen_capped, int(capped.sum()))
        for a in np.nonzero(capped)[0]:
            nid = int(lv["node_id"][a])
            if nid in capped_before and lv["in_slide"][a] and lv["has_sw"][a]:
                # a margi that finds the fix cluster at the cap frees the stored points and stops feeding it
                assert lv["n_point_fix"][a] == 0 and lv["pcr_fix"][a, 9] == capped_before[nid]
                released += 1
        capped_before = {int(i): n for i, n in zip(lv["node_id"][capped].tolist(), lv["pcr_fix"][capped][:, 9].tolist())}
        assert np.array_equal(lv["pcr_add"][:, 9], lv["pcr_fix"][:, 9] + lv["pcrs_local"][:, :, 9].sum(axis=1))
    assert seen_capped > 20 and released > 20
    # a capped fix cluster grows at most by one scan's worth past the cap (the scan that crossed it)
    assert lv["pcr_fix"][capped][:, 9].max() < cap + 6000


def test_fewer_touched_roots_than_threads_pushes_nothing():
    m = O.LocalMapOracle(win_size=3, thread_num=5, **PRM)
    pts = np.array([[0.2, 0.2, 0.2], [1.2, 0.3, 0.1], [2.5, 0.5, 0.5], [0.3, 0.4, 0.1]])      # 3 roots < 5 threads
    m.cut_voxel(0, pts, np.tile(np.eye(3) * 1e-4, (4, 1, 1)), pts)
    lv = m.leaves()
    assert m.counts()["roots"] == 3 and m.counts()["slide"] == 3 and np.all(lv["pcr_add"][:, 9] == 0)
    m2 = O.LocalMapOracle(win_size=3, thread_num=2, **PRM)
    m2.cut_voxel(0, pts, np.tile(np.eye(3) * 1e-4, (4, 1, 1)), pts)
    assert sorted(m2.leaves()["pcr_add"][:, 9].tolist()) == [1, 1, 2]


def test_cut_voxel_as_sort_and_segmented_continuation():
    """The data-parallel formulation planned for the device-resident map (DESIGN.md section 7, step ii), spelled out in numpy and checked
    against the tree walk: leaf id per point by descending a FLAT table of node ids (root key by the float-typed index, octants by strict
    '>' against centres rebuilt with the float quarter lengths), one STABLE sort of the scan by leaf id, then every touched leaf continues
    its running sums over its segment in scan order.  Same leaves (including the ones the scan creates), and pcr_add / pcrs_local equal to
    the oracle's per-point `allocate` bit for bit."""
    win = 4
    states = list(run_stream(S=7, win=win, pts=6000, seed=8))
    m = states[-1]["map"]                                   # a map with subdivided voxels, fix clusters and a shifted slot ring
    before = m.leaves()
    xyz, fp, poses, _ = synth.make_scans(win_size=8, pts_per_scan=6000, seed=synth.MASTER_SEED + 900 + 8)
    s = slice(fp[7], fp[8])
    pnt = xyz[s]; pw = to_world(poses[7], pnt); var = point_vars(pnt.shape[0], 99)
    ord_ = win - 1                                          # the scan enters the last window position
    m.cut_voxel(ord_, pnt, var, pw)
    after = m.leaves()

    vs = PRM["voxel_size"]; max_layer = PRM["max_layer"]
    leaf_ids = set(before["node_id"].tolist())
    # --- kernel 1: leaf id per point -------------------------------------------------------------------------------------------
    loc = (pw / vs).astype(np.float32)
    loc = np.where(loc < 0, loc - np.float32(1), loc).astype(np.int64)
    root48 = ((loc[:, 0] + 32768) << 32) | ((loc[:, 1] + 32768) << 16) | (loc[:, 2] + 32768)
    internal = set()                                         # subdivided nodes = proper prefixes of the leaves
    for i in leaf_ids:
        r, p, l = i >> 16, (i >> 7) & 0x1FF, i & 7
        if l >= 1: internal.add((r << 16) | 0)
        if l >= 2: internal.add((r << 16) | ((p & 0x1C0) << 7) | 1)
    ids = np.zeros(pnt.shape[0], dtype=np.uint64)
    for i in range(pnt.shape[0]):
        r = int(root48[i]); path = 0; layer = 0
        centre = (0.5 + loc[i]) * vs; ql = np.float32(vs / 4.0)
        while ((r << 16) | (path << 7) | layer) in internal:   # a node that is neither leaf nor internal is allocated here, as a leaf
            oct_ = [int(pw[i, k] > centre[k]) for k in range(3)]
            path |= (4 * oct_[0] + 2 * oct_[1] + oct_[2]) << (3 * (2 - layer))
            centre = centre + (2 * np.array(oct_) - 1) * float(ql); ql = np.float32(ql / np.float32(2)); layer += 1
        ids[i] = (r << 16) | (path << 7) | layer
    # --- kernel 2: stable sort by leaf id; kernel 3: one lane per touched leaf continues the sums in scan order ---------------------
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    starts = np.concatenate([[0], np.nonzero(sid[1:] != sid[:-1])[0] + 1, [sid.size]])
    row_before = {int(i): a for a, i in enumerate(before["node_id"].tolist())}
    row_after = {int(i): a for a, i in enumerate(after["node_id"].tolist())}
    assert set(row_after) == set(row_before) | set(sid.tolist())          # the same leaves exist afterwards, new ones included
    new_leaves = 0
    for a, b in zip(starts[:-1], starts[1:]):
        nid = int(sid[a]); seg = order[a:b]
        if nid in row_before:
            add = before["pcr_add"][row_before[nid]].copy(); loc_cl = before["pcrs_local"][row_before[nid], ord_].copy()
        else:
            add = np.zeros(10); loc_cl = np.zeros(10); new_leaves += 1
        for q in seg:                                          # PointCluster::push: P += v v^T (upper triangle), v += v, N += 1
            for acc, v in ((add, pw[q]), (loc_cl, pnt[q])):
                acc[0] += v[0] * v[0]; acc[1] += v[0] * v[1]; acc[2] += v[0] * v[2]; acc[3] += v[1] * v[1]; acc[4] += v[1] * v[2]; acc[5] += v[2] * v[2]
                acc[6] += v[0]; acc[7] += v[1]; acc[8] += v[2]; acc[9] += 1
        ra = row_after[nid]
        assert np.array_equal(add, after["pcr_add"][ra]) and np.array_equal(loc_cl, after["pcrs_local"][ra, ord_])
        kept = 0 if after["layer"][ra] == max_layer else seg.size
        assert after["n_points"][ra, ord_] == (before["n_points"][row_before[nid], ord_] if nid in row_before else 0) + kept
    assert new_leaves > 20 and len(starts) - 1 > 500
    # untouched leaves did not change
    untouched = [i for i in row_before if i not in set(sid.tolist())]
    assert np.array_equal(before["pcr_add"][[row_before[i] for i in untouched]], after["pcr_add"][[row_after[i] for i in untouched]])


def push_seq(acc, v):
    """PointCluster::push on the packed (P upper triangle, v, N) record, the oracle's operation order."""
    acc[0] += v[0] * v[0]; acc[1] += v[0] * v[1]; acc[2] += v[0] * v[2]; acc[3] += v[1] * v[1]; acc[4] += v[1] * v[2]; acc[5] += v[2] * v[2]
    acc[6] += v[0]; acc[7] += v[1]; acc[8] += v[2]; acc[9] += 1


def node_centre(nid, vs):
    """Centre and float quarter length of a node from its id, by the recurrences of cut_voxel / allocate."""
    r = nid >> 16; path = (nid >> 7) & 0x1FF; layer = nid & 7
    loc = np.array([((r >> 32) & 0xFFFF) - 32768, ((r >> 16) & 0xFFFF) - 32768, (r & 0xFFFF) - 32768], dtype=np.float64)
    centre = (0.5 + loc) * vs; ql = np.float32(vs / 4.0)
    for l in range(layer):
        o = (path >> (3 * (2 - l))) & 7
        centre = centre + (2 * np.array([(o >> 2) & 1, (o >> 1) & 1, o & 1]) - 1) * float(ql); ql = np.float32(ql / np.float32(2))
    return centre, ql


def test_recut_layer_by_layer_as_rebucketing():
    """Step iii of the planned device formulation in numpy against the recursive `recut`: layer by layer, every leaf is
    judged (point floor, eigen-decomposition, plane_judge); the ones that split hand their points to children by octant -- the fix points
    first, then the window's scans in window order, each in stored order -- which is one stable sort of that sequence by child; the
    children's running sums over their segments then equal the recursive walk bit for bit, flags included."""
    win = 4
    states = list(run_stream(S=7, win=win, pts=6000, seed=9))
    m = states[-1]["map"]; x_buf = [p for p in states[-1]["x_buf"][1:]]
    xyz, fp, poses, _ = synth.make_scans(win_size=8, pts_per_scan=6000, seed=synth.MASTER_SEED + 900 + 9)
    s = slice(fp[7], fp[8])
    pw = to_world(poses[7], xyz[s])
    m.cut_voxel(win - 1, xyz[s], point_vars(pw.shape[0], 5), pw)
    x_buf.append(poses[7].copy()); xs = np.stack(x_buf); win_count = win
    before = m.leaves()
    pts_of = {}
    for a in np.nonzero((before["layer"] < 2) & before["has_sw"] & before["in_slide"])[0]:
        nid = int(before["node_id"][a])
        pts_of[nid] = [m.leaf_points(nid, -1)[0]] + [m.leaf_points(nid, i)[0] for i in range(win_count)]
    f = O.Oracle(win)
    m.recut(win_count, xs, f)
    after = m.leaves()
    row_after = {int(i): a for a, i in enumerate(after["node_id"].tolist())}

    vs = PRM["voxel_size"]; max_layer = PRM["max_layer"]
    # the work list of a layer: leaves as records (id, pcr_add, pcr_fix, window clusters, point lists [fix, scan 0, scan 1, ...])
    work = []
    for a in np.nonzero(before["in_slide"])[0]:
        nid = int(before["node_id"][a])
        work.append(dict(id=nid, add=before["pcr_add"][a].copy(), fix=before["pcr_fix"][a].copy(), loc=before["pcrs_local"][a].copy(),
                         isexist=bool(before["isexist"][a]), has_sw=bool(before["has_sw"][a]), was_plane=bool(before["is_plane"][a]), pts=pts_of.get(nid)))
    split_total = 0
    checked = 0
    while work:
        nxt = []
        ev_all, _ = O.plane_fit(np.stack([w["add"] for w in work]))
        for w, ev in zip(work, ev_all):
            layer = w["id"] & 7
            N = w["add"][9]
            ra = row_after.get(w["id"])
            if N <= PRM["min_point"][layer]:
                assert ra is not None and not after["is_plane"][ra]; checked += 1
                continue
            if not w["isexist"] or not w["has_sw"]:
                assert ra is not None and after["is_plane"][ra] == w["was_plane"]; checked += 1
                continue
            plane = bool(ev[0] < PRM["min_eigen_value"] and ev[0] / ev[2] < PRM["plane_eigen_value_thre"][layer])
            if plane or layer >= max_layer:
                assert ra is not None and bool(after["is_plane"][ra]) == plane; checked += 1
                assert np.array_equal(after["pcr_add"][ra], w["add"]) and np.array_equal(after["pcr_fix"][ra], w["fix"]) and np.array_equal(after["pcrs_local"][ra], w["loc"])
                continue
            # --- split: one sequence [fix | scan 0 | scan 1 | ...], child id per element, stable sort, segmented sums ---------------
            assert ra is None                                                   # the node is internal afterwards
            split_total += 1
            centre, ql = node_centre(w["id"], vs)
            seq_w, seq_b, seq_src = [], [], []
            for src, P in enumerate(w["pts"]):                                  # src 0 = fix (world), src i+1 = scan i of the window (body)
                if P.shape[0] == 0:
                    continue
                W_ = P if src == 0 else to_world(xs[src - 1], P)
                seq_w.append(W_); seq_b.append(P); seq_src.append(np.full(P.shape[0], src))
            W_ = np.concatenate(seq_w); B_ = np.concatenate(seq_b); SRC = np.concatenate(seq_src)
            octant = 4 * (W_[:, 0] > centre[0]) + 2 * (W_[:, 1] > centre[1]) + (W_[:, 2] > centre[2])
            order = np.argsort(octant, kind="stable")
            for o in np.unique(octant):
                seg = order[octant[order] == o]
                r = w["id"] >> 16; path = ((w["id"] >> 7) & 0x1FF) | (int(o) << (3 * (2 - layer)))
                cid = (r << 16) | (path << 7) | (layer + 1)
                child = dict(id=cid, add=np.zeros(10), fix=np.zeros(10), loc=np.zeros((win, 10)), isexist=False, has_sw=False, was_plane=False,
                             pts=[[] for _ in range(win_count + 1)])
                for q in seg:
                    if SRC[q] == 0:                                             # push_fix: pcr_fix and pcr_add take the world point
                        push_seq(child["fix"], W_[q]); push_seq(child["add"], W_[q])
                    else:                                                       # push: window cluster takes the body point, pcr_add the world point
                        push_seq(child["loc"][SRC[q] - 1], B_[q]); push_seq(child["add"], W_[q])
                        child["isexist"] = True; child["has_sw"] = True
                    if layer + 1 < max_layer:
                        child["pts"][SRC[q]].append(W_[q] if SRC[q] == 0 else B_[q])
                child["pts"] = [np.array(p).reshape(-1, 3) for p in child["pts"]]
                nxt.append(child)
        work = nxt
    assert split_total > 20 and checked > 1000
