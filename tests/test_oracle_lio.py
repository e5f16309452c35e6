"""CPU oracle of the odometry's point-to-plane update (oracle/vxo_lio.hpp) against independent statements of the same
mathematics: a brute-force numpy matcher, closed forms of calcBodyVar, finite differences for plane_update, and the
EKF's contraction towards the pose the scan was taken at.  (The reference holds no golden vectors for this path.)"""
import numpy as np
import pytest

from tests import _oracle as O
from voxel_slam_amd import synth


def unpack(st):
    return st[:9].reshape(3, 3).T, st[9:12]


def brute_force_match(pm, pnt, var, state, cov):
    """Leaf by box containment, tests in float64.  Returns (plane index or -1, margin) per point; margin = distance of the
    closest decision (voxel face / gate) from flipping, so that callers can skip the knife-edge points."""
    R, p = unpack(state)
    w = pnt @ R.T + p
    rot_var, tsl_var = cov[:3, :3], cov[3:6, 3:6]
    lo = pm.box_center - pm.box_half[:, None]; hi = pm.box_center + pm.box_half[:, None]
    out = np.full(len(w), -1); margin = np.full(len(w), np.inf)
    for i, x in enumerate(w):
        inside = np.all((x > lo) & (x <= hi), axis=1)
        face = np.min(np.abs(np.concatenate([x - lo[inside], hi[inside] - x]))) if inside.any() else np.min(np.abs(x / pm.voxel_size - np.round(x / pm.voxel_size)))
        margin[i] = face
        k = np.nonzero(inside)[0]
        if k.size == 0:
            continue
        k = k[0]
        if not pm.is_plane[k]:
            continue
        d = x - pm.center[k]; n = pm.normal[k]
        dp = abs(n @ d)
        range_dis = d @ d - dp * dp
        ph = np.array([[0, -pnt[i][2], pnt[i][1]], [pnt[i][2], 0, -pnt[i][0]], [-pnt[i][1], pnt[i][0], 0]])
        vw = R @ var[i] @ R.T + ph @ rot_var @ ph.T + tsl_var
        J = np.concatenate([d, -n])
        sig = J @ pm.plane_var[k] @ J + n @ vw @ n
        margin[i] = min(margin[i], abs(range_dis - 9 * pm.radius[k]) / (9 * pm.radius[k]), abs(dp - 3 * np.sqrt(sig)) / (3 * np.sqrt(sig)))
        if range_dis <= 9 * pm.radius[k] and dp < 3 * np.sqrt(sig):
            out[i] = k
    return out, margin


def planes_only_index(pm):
    """Plane ids as the oracle hands them out (order of insertion, every leaf counted)."""
    return np.arange(len(pm.layer))


@pytest.mark.parametrize("max_layer,voxel_size", [(2, 1.0), (1, 2.0), (3, 0.5), (0, 1.0)])
def test_match_agrees_with_brute_force(max_layer, voxel_size):
    pm = synth.make_plane_map(n_roots=300, extent=4, voxel_size=voxel_size, max_layer=max_layer, seed=11 + max_layer)
    sc = synth.make_lio_scan(pm, n_points=1500, seed=12 + max_layer)
    o = O.LioOracle(voxel_size, max_layer); o.map_update(*pm.args()); o.var_init(sc.xyz)
    pnt, var = o.read_points()
    got = o.sweep(sc.state_init, sc.cov, want_points=True)
    ref, margin = brute_force_match(pm, pnt, var, sc.state_init, sc.cov)
    ok = margin > 1e-5
    assert ok.mean() > 0.98
    assert np.array_equal(got["plane_of_point"][ok], ref[ok])
    assert got["match_num"] == (got["plane_of_point"] >= 0).sum() and got["match_num"] > 300
    # the sums, recomputed from the association
    R, p = unpack(sc.state_init)
    HTH = np.zeros((6, 6)); HTz = np.zeros(6); nnt = np.zeros((3, 3))
    for i in np.nonzero(got["plane_of_point"] >= 0)[0]:
        k = got["plane_of_point"][i]; n = pm.normal[k]
        w = R @ pnt[i] + p
        jac = np.concatenate([np.cross(pnt[i], R.T @ n), n])
        Rinv = 1.0 / (0.0005 + got["sigma_of_point"][i])
        HTH += Rinv * np.outer(jac, jac); HTz -= Rinv * jac * (n @ (w - pm.center[k])); nnt += np.outer(n, n)
    assert np.allclose(got["HTH"], HTH, rtol=1e-10) and np.allclose(got["HTz"], HTz, rtol=1e-9, atol=1e-6) and np.allclose(got["nnt"], nnt, rtol=1e-10)


def test_float_typed_voxel_index_is_reproduced():
    """`loc[j] = wld[j] / voxel_size` is a float upstream (voxel_map.hpp:1678-1683): a point 1e-7 below a voxel face is looked up in
    the NEXT voxel, and below zero `-= 1` shifts exact integers one further down.  The restatement keeps both."""
    z6 = np.zeros((6, 6)); one = np.float64(np.float32(0.2))
    loc = np.array([[3, 0, 0], [2, 0, 0], [-2, 0, 0], [-1, 0, 0]]); layer = np.zeros(4, dtype=np.int32); path = np.zeros(4, dtype=np.int32)
    center = loc + 0.5; normal = np.tile([0.0, 0.0, 1.0], (4, 1))
    pv = np.tile(np.eye(6) * 1e-2, (4, 1, 1))
    o = O.LioOracle(1.0, 2); o.map_update(loc, layer, path, center, normal, pv, np.full(4, 10.0))
    pnt = np.array([[2.99999999, 0.5, 0.5], [2.9999, 0.5, 0.5], [-1.0, 0.5, 0.5], [-1.0001, 0.5, 0.5]])
    o.set_points(pnt, np.tile(np.eye(3) * 1e-4, (4, 1, 1)))
    st = np.concatenate([np.eye(3).reshape(9), np.zeros(15)])
    r = o.sweep(st, np.eye(15) * 1e-4, want_points=True)
    assert list(r["plane_of_point"]) == [0, 1, 2, 2]
    del z6, one


def test_node_cache_matches_reference_semantics():
    """octos[i] keeps the matched leaf across the iterations of one call: a point that drifts out of the leaf's box is looked up
    afresh, one that stays is tested against the cached leaf only."""
    pm = synth.make_plane_map(n_roots=200, extent=3, seed=21)
    sc = synth.make_lio_scan(pm, n_points=3000, seed=22)
    o = O.LioOracle(pm.voxel_size, pm.max_layer); o.map_update(*pm.args()); o.var_init(sc.xyz)
    a = o.sweep(sc.state_init, sc.cov, reset_cache=True, want_points=True)
    b = o.sweep(sc.state_gt, sc.cov, reset_cache=False, want_points=True)      # cached
    c = o.sweep(sc.state_gt, sc.cov, reset_cache=True, want_points=True)       # fresh walk
    # away from voxel faces the cache cannot change the answer
    assert (b["plane_of_point"] != c["plane_of_point"]).mean() < 1e-3
    assert a["match_num"] > 0 and b["match_num"] >= a["match_num"]


def test_state_estimation_contracts_towards_truth_and_follows_the_schedule():
    pm = synth.make_plane_map(n_roots=1500, seed=31)
    sc = synth.make_lio_scan(pm, n_points=20000, seed=32)
    o = O.LioOracle(pm.voxel_size, pm.max_layer); o.map_update(*pm.args()); o.var_init(sc.xyz)
    res = o.lio_state_estimation(sc.state_init, sc.cov)
    e0 = synth.pose_errors(sc.state_init[None, :12], sc.state_gt[None, :12]); e1 = synth.pose_errors(res["state"][None, :12], sc.state_gt[None, :12])
    assert e1[0] < 0.1 * e0[0] and e1[1] < 0.1 * e0[1]
    assert res["ok"] and 2 <= res["iterations"] <= 4 and res["match_num"] > 0.7 * 20000
    # posterior covariance shrinks on the observed block, velocity / bias blocks only through correlations
    assert np.all(np.diag(res["cov"])[:6] < np.diag(sc.cov)[:6]) and np.all(np.linalg.eigvalsh(0.5 * (res["cov"] + res["cov"].T)) > 0)
    # a prior so tight that the first step is below the convergence thresholds: one rematch follows and the loop ends
    # (iterations == 2, voxelslam.cpp:934-946)
    assert o.lio_state_estimation(sc.state_gt, np.eye(15) * 1e-12)["iterations"] == 2
    # v, bg, ba move only through the prior's correlations: with a block-diagonal prior they stay put
    cov = np.eye(15) * 1e-4
    res3 = o.lio_state_estimation(sc.state_init, cov)
    assert np.allclose(res3["state"][12:21], sc.state_init[12:21], atol=1e-12)
    # a degenerate scene (every normal along z) fails the eigenvalue test (voxelslam.cpp:951-957)
    flat = synth.make_plane_map(n_roots=150, extent=3, seed=33)
    flat.normal[:] = [0.0, 0.0, 1.0]
    sc2 = synth.make_lio_scan(flat, n_points=3000, seed=34)
    o2 = O.LioOracle(flat.voxel_size, flat.max_layer); o2.map_update(*flat.args()); o2.var_init(sc2.xyz)
    assert not o2.lio_state_estimation(sc2.state_init, sc2.cov)["ok"]


def test_calc_body_var_closed_form():
    """var = range_var d d^T + range^2 sin^2(beam) (I - d d^T): eigenvalues known in closed form."""
    rng = np.random.default_rng(5)
    xyz = rng.uniform(-30, 30, size=(200, 3)).astype(np.float32)
    xyz[0, 2] = 0.0          # the reference nudges an exactly-zero z (voxelslam.hpp:166-167)
    o = O.LioOracle(); o.var_init(xyz, dept_err=0.02, beam_err=0.05)
    pnt, var = o.read_points()
    assert pnt[0, 2] == 0.0001 and np.allclose(pnt[1:], xyz[1:].astype(np.float64))
    for i in range(200):
        r = np.linalg.norm(pnt[i]); d = pnt[i] / r
        expect = np.float32(0.02) ** 2 * np.outer(d, d) + r * r * np.sin(np.float32(0.05) * 0.017453293) ** 2 * (np.eye(3) - np.outer(d, d))
        assert np.allclose(var[i], expect, rtol=1e-5, atol=1e-12)   # `range` is a float upstream
    # extrinsic: rotates the covariance, moves the point
    Rx = synth.rodrigues(np.array([0.1, 0.2, -0.3])); px = np.array([0.05, -0.02, 0.1])
    o.var_init(xyz, ext_R=Rx, ext_p=px)
    p2, v2 = o.read_points()
    assert np.allclose(p2, pnt @ Rx.T + px) and np.allclose(v2, Rx @ var @ Rx.T, rtol=1e-12, atol=1e-18)


def test_pvec_update_matches_numpy():
    pm = synth.make_plane_map(n_roots=50, extent=2, seed=41)
    sc = synth.make_lio_scan(pm, n_points=300, seed=42)
    o = O.LioOracle(); o.var_init(sc.xyz)
    pnt, var = o.read_points()
    R, p = unpack(sc.state_init)
    pw, vw = o.pvec_update(sc.state_init, sc.cov)
    hat = lambda v: np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    assert np.allclose(pw, pnt @ R.T + p)
    for i in range(0, 300, 17):
        assert np.allclose(vw[i], R @ var[i] @ R.T + hat(pnt[i]) @ sc.cov[:3, :3] @ hat(pnt[i]).T + sc.cov[3:6, 3:6], rtol=1e-12)


def test_plane_update_is_first_order_covariance_propagation():
    """plane_var = J cov_add J^T with J = d(normal, center) / d(P_sym6, v) (voxel_map.hpp:1118-1146): checked by finite differences."""
    rng = np.random.default_rng(9)
    n_pts = 60
    nrm = np.array([0.2, -0.3, 0.93]); nrm /= np.linalg.norm(nrm)
    a = np.cross(nrm, [1, 0, 0]); a /= np.linalg.norm(a); b = np.cross(nrm, a)
    pts = np.array([3.0, -2.0, 1.0]) + rng.uniform(-0.5, 0.5, (n_pts, 1)) * a + rng.uniform(-0.5, 0.5, (n_pts, 1)) * b + rng.normal(0, 0.02, (n_pts, 1)) * nrm

    def cluster(P6, v):
        return np.concatenate([P6, v, [n_pts]])

    def fit(P6, v):
        P = np.array([[P6[0], P6[1], P6[2]], [P6[1], P6[3], P6[4]], [P6[2], P6[4], P6[5]]])
        c = v / n_pts
        lam, U = np.linalg.eigh(P / n_pts - np.outer(c, c))
        return lam, U

    P = pts.T @ pts; P6 = np.array([P[0, 0], P[0, 1], P[0, 2], P[1, 1], P[1, 2], P[2, 2]]); v = pts.sum(axis=0)
    ev, U = O.plane_fit(cluster(P6, v)[None])
    out = O.plane_update(cluster(P6, v)[None], ev, U, np.eye(9)[None])
    n0 = out["normal"][0]
    Jfd = np.zeros((3, 9))
    x0 = np.concatenate([P6, v])
    for k in range(9):
        h = 1e-6 * max(1.0, abs(x0[k]))
        xp = x0.copy(); xp[k] += h; xm = x0.copy(); xm[k] -= h
        up = fit(xp[:6], xp[6:])[1][:, 0]; um = fit(xm[:6], xm[6:])[1][:, 0]
        up *= np.sign(up @ n0); um *= np.sign(um @ n0)
        Jfd[:, k] = (up - um) / (2 * h)
    pv = out["plane_var"][0]
    assert np.allclose(pv[:3, :3], Jfd @ Jfd.T, rtol=2e-4, atol=1e-12)
    assert np.allclose(pv[:3, 3:], Jfd[:, 6:] / n_pts, rtol=2e-4, atol=1e-12) and np.allclose(pv[3:, :3], pv[:3, 3:].T)
    assert np.allclose(pv[3:, 3:], np.eye(3) / n_pts ** 2)
    assert np.allclose(out["center"][0], v / n_pts) and out["radius"][0] == np.float32(ev[0, 2])
    # cov_add: sum over points of B var B^T with B = d(P_sym6, v)/d(point)
    var = np.tile(np.diag([1e-4, 2e-4, 3e-4]), (n_pts, 1, 1))
    ca = O.cov_add_build(pts, var, np.array([0, n_pts]))[0]
    expect = np.zeros((9, 9))
    for x, V in zip(pts, var):
        B = np.array([[2 * x[0], 0, 0], [x[1], x[0], 0], [x[2], 0, x[0]], [0, 2 * x[1], 0], [0, x[2], x[1]], [0, 0, 2 * x[2]], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
        expect += B @ V @ B.T
    assert np.allclose(ca, expect, rtol=1e-12)


def test_down_sampling_voxel_running_mean():
    rng = np.random.default_rng(77)
    xyz = rng.uniform(-20, 20, size=(20000, 3)).astype(np.float32)
    out = O.down_sampling_voxel(xyz, 0.5)
    # one point per occupied voxel (float-typed index as upstream), each inside its voxel, close to the plain mean
    loc = (xyz.astype(np.float64) / 0.5).astype(np.float32); loc = np.where(loc < 0, loc - np.float32(1.0), loc).astype(np.int64)
    uniq, inv, cnt = np.unique(loc, axis=0, return_inverse=True, return_counts=True)
    assert out.shape[0] == uniq.shape[0]
    mean = np.zeros((uniq.shape[0], 3)); np.add.at(mean, inv.ravel(), xyz.astype(np.float64)); mean /= cnt[:, None]
    assert np.allclose(out, mean, atol=2e-5)                         # np.unique sorts rows lexicographically = the oracle's order
    # voxel sizes below 1 mm leave the cloud alone (tools.hpp:203)
    assert np.array_equal(O.down_sampling_voxel(xyz[:50], 1e-4), xyz[:50])
