"""Odometry point-to-plane update on the GPU (vxba_lio.hip) against the CPU oracle (oracle/vxo_lio.hpp): per-point plane
association (integer work: exact), the sweep sums, var_init / pvec_update, and the whole lio_state_estimation."""
import numpy as np
import pytest

from tests import _oracle as O
from voxel_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vx():
    from voxel_slam_amd import vxba
    vxba.load_library()
    return vxba


def both(vx, pm, sc, raw=True):
    o = O.LioOracle(pm.voxel_size, pm.max_layer); o.map_update(*pm.args()); o.var_init(sc.xyz)
    g = vx.LioEstimator(pm.voxel_size, pm.max_layer); g.map_update(*pm.args())
    if raw:
        g.var_init(sc.xyz)
    else:
        g.set_points(*o.read_points())
    return o, g


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def check_sweep(a, b, tol=1e-10):
    assert a["match_num"] == b["match_num"]
    assert rel(a["HTH"], b["HTH"]) < tol and rel(a["HTz"], b["HTz"]) < 1e3 * tol and rel(a["nnt"], b["nnt"]) < tol
    assert np.array_equal(a["HTH"], a["HTH"].T)


# the last case has the shape of a real scan -- many returns per plane (60k points on ~1.3k planes), records shared through L2
@pytest.mark.parametrize("max_layer,voxel_size,n_roots,n_points", [(2, 1.0, 3000, 30000), (1, 2.0, 400, 8000), (3, 0.5, 2500, 20000), (0, 1.0, 500, 5000), (2, 1.0, 300, 60000)])
def test_sweep_matches_oracle(vx, max_layer, voxel_size, n_roots, n_points):
    pm = synth.make_plane_map(n_roots=n_roots, extent=8 if n_roots > 1000 else 4, voxel_size=voxel_size, max_layer=max_layer, seed=2000 + max_layer)
    sc = synth.make_lio_scan(pm, n_points=n_points, seed=2100 + max_layer)
    o, g = both(vx, pm, sc, raw=False)
    assert g.map_size()[1] == int(pm.is_plane.sum()) and g.scan_size() == n_points
    ro = o.sweep(sc.state_init, sc.cov, want_points=True); rg = g.sweep(sc.state_init, sc.cov, want_points=True)
    # which points matched, and with what variance (the exact leaf identity is the next test's subject)
    ids_o = ro["plane_of_point"]; ids_g = rg["plane_of_point"]
    assert np.array_equal(ids_o >= 0, ids_g >= 0)
    m = ids_o >= 0
    assert np.allclose(ro["sigma_of_point"][m], rg["sigma_of_point"][m], rtol=1e-11)
    check_sweep(rg, ro)
    # second sweep at another pose WITH the node cache of the first, then a fresh one
    ro2 = o.sweep(sc.state_gt, sc.cov, reset_cache=False, want_points=True); rg2 = g.sweep(sc.state_gt, sc.cov, reset_cache=False, want_points=True)
    assert np.array_equal(ro2["plane_of_point"] >= 0, rg2["plane_of_point"] >= 0)
    check_sweep(rg2, ro2)
    ro3 = o.sweep(sc.state_gt, sc.cov, reset_cache=True); rg3 = g.sweep(sc.state_gt, sc.cov, reset_cache=True)
    check_sweep(rg3, ro3)
    # bitwise reproducible
    rg4 = g.sweep(sc.state_gt, sc.cov, reset_cache=True)
    assert np.array_equal(rg4["HTH"], rg3["HTH"]) and np.array_equal(rg4["HTz"], rg3["HTz"])


def test_plane_association_is_exact_including_the_float_quirks(vx):
    """Every point lands on the same leaf as in the reference walk: identify leaves by (centre, normal) of the matched plane."""
    pm = synth.make_plane_map(n_roots=4000, extent=9, seed=2200)
    sc = synth.make_lio_scan(pm, n_points=60000, seed=2201)
    # a batch of points sitting within float rounding of voxel faces, and exact negative integers
    rng = np.random.default_rng(3)
    face = rng.integers(-8, 8, size=(4000, 3)).astype(np.float64) + rng.choice([-1e-7, -1e-8, 0.0, 1e-8, 1e-7, 0.5], size=(4000, 3))
    o, g = both(vx, pm, sc, raw=False)
    pnt, var = o.read_points()
    st = np.concatenate([np.eye(3).reshape(9), np.zeros(15)])
    pnt = pnt @ sc.state_gt[:9].reshape(3, 3) + sc.state_gt[9:12]       # world coordinates, looked up under the identity pose
    pnt2 = np.concatenate([pnt, face]); var2 = np.concatenate([var, np.tile(np.eye(3) * 1e-2, (4000, 1, 1))])
    o.set_points(pnt2, var2); g.set_points(pnt2, var2)
    big = np.eye(15) * 1e-2                                 # wide gates: almost every point inside a plane leaf matches
    ro = o.sweep(st, big, want_points=True); rg = g.sweep(st, big, want_points=True)
    assert ro["match_num"] > 30000
    # translate both id spaces to the caller's leaf list
    carry = np.nonzero(pm.is_plane == 1)[0]
    leaf_o = ro["plane_of_point"]
    leaf_g = np.where(rg["plane_of_point"] >= 0, -2, -1)
    # GPU records are handed out by atomics inside one update call: recover the leaf from the record's centre
    leaf_of_record = records_to_leaves(g, pm, carry)
    leaf_g[rg["plane_of_point"] >= 0] = leaf_of_record[rg["plane_of_point"][rg["plane_of_point"] >= 0]]
    assert np.array_equal(leaf_o, leaf_g)
    assert np.array_equal(ro["sigma_of_point"] > 0, rg["sigma_of_point"] > 0)


def records_to_leaves(g, pm, carry):
    """record index -> leaf index, by probing each plane with a point on its centre under an identity pose."""
    n_keep = g.scan_size()
    keep = g.read_points()
    g.set_points(pm.center[carry], np.tile(np.eye(3) * 1e-2, (carry.size, 1, 1)))
    st = np.concatenate([np.eye(3).reshape(9), np.zeros(15)])
    r = g.sweep(st, np.eye(15) * 1e-2, want_points=True)
    assert np.all(r["plane_of_point"] >= 0)               # a plane's own centre always matches it
    out = np.full(r["plane_of_point"].max() + 1, -1)
    out[r["plane_of_point"]] = carry
    g.set_points(*keep)
    assert g.scan_size() == n_keep
    return out


def test_var_init_and_pvec_update(vx):
    pm = synth.make_plane_map(n_roots=300, extent=4, seed=2300)
    sc = synth.make_lio_scan(pm, n_points=20000, seed=2301)
    sc.xyz[:5, 2] = 0.0
    Rx = synth.rodrigues(np.array([0.01, -0.02, 0.03])); px = np.array([0.04, 0.02, -0.03])
    o = O.LioOracle(); g = vx.LioEstimator()
    o.var_init(sc.xyz, Rx, px, 0.02, 0.05); g.var_init(sc.xyz, Rx, px, 0.02, 0.05)
    po, vo = o.read_points(); pg, vg = g.read_points()
    assert np.allclose(pg, po, rtol=1e-15, atol=1e-15) and np.allclose(vg, vo, rtol=1e-12, atol=1e-20)
    wo, wvo = o.pvec_update(sc.state_init, sc.cov); wg, wvg = g.pvec_update(sc.state_init, sc.cov)
    assert np.allclose(wg, wo, rtol=1e-14, atol=1e-14) and np.allclose(wvg, wvo, rtol=1e-11, atol=1e-18)


@pytest.mark.parametrize("device_ekf", ["1", "0"])
@pytest.mark.parametrize("seed,n_roots,n_points,raw", [(2400, 3000, 40000, True), (2410, 800, 6000, False), (2420, 6000, 100000, True), (2430, 400, 100000, True)])
def test_state_estimation_matches_oracle(vx, seed, n_roots, n_points, raw, device_ekf):
    """device_ekf: the 15-dimensional EKF algebra between the sweeps as a kernel, all iterations enqueued up front (default), or on the
    host with a round trip per iteration (vxba_lio_set_option(VXBA_LIO_OPT_DEVICE_EKF, 0))."""
    pm = synth.make_plane_map(n_roots=n_roots, extent=10, seed=seed)
    sc = synth.make_lio_scan(pm, n_points=n_points, seed=seed + 1)
    o, g = both(vx, pm, sc, raw=raw)
    g.set_option("device_ekf", int(device_ekf))
    ref = o.lio_state_estimation(sc.state_init, sc.cov); got = g.lio_state_estimation(sc.state_init, sc.cov)
    assert got["iterations"] == ref["iterations"] and got["ok"] == ref["ok"] and got["match_num"] == ref["match_num"]
    for a, b in zip(got["sweeps"], ref["sweeps"]):
        check_sweep(a, b, tol=1e-9)
    et, er = synth.pose_errors(got["state"][None, :12], ref["state"][None, :12])
    assert et < 1e-9 and er < 1e-9, (et, er)                       # contract: 1e-4 m / 1e-4 rad
    assert np.allclose(got["state"][12:], ref["state"][12:], atol=1e-10)
    assert rel(got["cov"], ref["cov"]) < 1e-8 and abs(got["min_eig"] - ref["min_eig"]) < 1e-8 * ref["min_eig"]
    e0 = synth.pose_errors(sc.state_init[None, :12], sc.state_gt[None, :12]); e1 = synth.pose_errors(got["state"][None, :12], sc.state_gt[None, :12])
    assert e1[0] < 0.2 * e0[0]


def test_map_upsert_remove_clear_and_growth(vx):
    pm = synth.make_plane_map(n_roots=2500, extent=8, seed=2500)
    sc = synth.make_lio_scan(pm, n_points=20000, seed=2501)
    n = len(pm.layer)
    o, g = both(vx, pm, sc, raw=False)
    full = g.sweep(sc.state_init, sc.cov)
    # the same map inserted in many small batches (forces table growth + rehash) gives the same sums
    g2 = vx.LioEstimator(pm.voxel_size, pm.max_layer); g2.set_points(*o.read_points())
    cuts = [0, 10, 700, 701, 3000, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        g2.map_update(pm.loc[a:b], pm.layer[a:b], pm.path[a:b], pm.center[a:b], pm.normal[a:b], pm.plane_var[a:b], pm.radius[a:b], pm.is_plane[a:b])
    r2 = g2.sweep(sc.state_init, sc.cov)
    assert r2["match_num"] == full["match_num"] and rel(r2["HTH"], full["HTH"]) < 1e-12
    assert g2.map_size() == g.map_size()
    # updating every plane in place (same leaves, new centres) reuses the records ...
    before = g2.map_size()
    g2.map_update(pm.loc, pm.layer, pm.path, pm.center + 1e-3, pm.normal, pm.plane_var, pm.radius, pm.is_plane)
    assert g2.map_size() == before
    o2 = O.LioOracle(pm.voxel_size, pm.max_layer); o2.map_update(pm.loc, pm.layer, pm.path, pm.center + 1e-3, pm.normal, pm.plane_var, pm.radius, pm.is_plane)
    o2.set_points(*o.read_points())
    check_sweep(g2.sweep(sc.state_init, sc.cov), o2.sweep(sc.state_init, sc.cov))
    # ... removing half of the planes equals a map built without them
    drop = np.arange(n) % 2 == 0
    isp = pm.is_plane.copy(); isp[drop] = 0
    g2.map_update(pm.loc[drop], pm.layer[drop], pm.path[drop], pm.center[drop], pm.normal[drop], pm.plane_var[drop], pm.radius[drop], isp[drop])
    o3 = O.LioOracle(pm.voxel_size, pm.max_layer); o3.map_update(pm.loc, pm.layer, pm.path, pm.center + 1e-3, pm.normal, pm.plane_var, pm.radius, isp)
    o3.set_points(*o.read_points())
    check_sweep(g2.sweep(sc.state_init, sc.cov), o3.sweep(sc.state_init, sc.cov))
    # clear: nothing matches, the handle stays usable
    g2.map_clear()
    assert g2.map_size() == (0, 0) and g2.sweep(sc.state_init, sc.cov)["match_num"] == 0
    g2.map_update(*pm.args())
    check_sweep(g2.sweep(sc.state_init, sc.cov), o.sweep(sc.state_init, sc.cov))


def test_lio_edge_cases_and_errors(vx):
    g = vx.LioEstimator(1.0, 2)
    st = np.concatenate([np.eye(3).reshape(9), np.zeros(15)])
    # empty scan, empty map
    r = g.sweep(st, np.eye(15) * 1e-4)
    assert r["match_num"] == 0 and not r["HTH"].any()
    g.set_points(np.array([[0.5, 0.5, 0.5]]), np.eye(3)[None] * 1e-4)
    assert g.sweep(st, np.eye(15) * 1e-4)["match_num"] == 0
    res = g.lio_state_estimation(st, np.eye(15) * 1e-4)          # no match: the prior is returned, flagged degenerate
    assert not res["ok"] and res["match_num"] == 0 and np.allclose(res["state"], st) and np.allclose(res["cov"], np.eye(15) * 1e-4)
    with pytest.raises(vx.VxbaError):
        g.map_update(np.array([[1 << 21, 0, 0]]), [0], [0], np.zeros((1, 3)), np.array([[0, 0, 1.0]]), np.zeros((1, 6, 6)), [0.1])
    with pytest.raises(vx.VxbaError):
        g.map_update(np.array([[0, 0, 0]]), [3], [0], np.zeros((1, 3)), np.array([[0, 0, 1.0]]), np.zeros((1, 6, 6)), [0.1])
    with pytest.raises(vx.VxbaError):
        g.map_update(np.array([[0, 0, 0]]), [1], [9], np.zeros((1, 3)), np.array([[0, 0, 1.0]]), np.zeros((1, 6, 6)), [0.1])
    with pytest.raises(vx.VxbaError):
        vx.LioEstimator(1.0, 4)
    with pytest.raises(vx.VxbaError):
        g.lio_state_estimation(st, np.zeros((15, 15)))


def test_plane_records_built_on_the_gpu_match_oracle(vx):
    """The producers of the plane map: world points + covariances per leaf -> cluster (K1) -> eigen-decomposition (K4) -> cov_add
    (sum of Bf_var) -> plane_update, all on the GPU, against the oracle; then a state estimation on the map so built."""
    pm = synth.make_plane_map(n_roots=1200, extent=6, seed=2600)
    carry = np.nonzero(pm.is_plane == 1)[0]
    rng = np.random.default_rng(2601)
    counts = rng.integers(12, 60, size=carry.size)
    cell_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    k = np.repeat(carry, counts)
    n = pm.normal[k]
    a = np.cross(n, np.array([0.31, 0.52, 0.79])); a /= np.linalg.norm(a, axis=1, keepdims=True); b = np.cross(n, a)
    h = pm.box_half[k][:, None] * 0.9
    world = pm.center[k] + rng.uniform(-1, 1, (k.size, 1)) * h * a + rng.uniform(-1, 1, (k.size, 1)) * h * b + rng.normal(0, 0.02, (k.size, 1)) * n
    # world covariances of those points as pvec_update leaves them (seen from the origin under a small pose covariance)
    st = np.concatenate([np.eye(3).reshape(9), np.zeros(15)])
    o = O.LioOracle(); o.var_init(world.astype(np.float32))
    g = vx.LioEstimator(); g.var_init(world.astype(np.float32))
    cov = np.eye(15) * 1e-6
    pw_o, var_o = o.pvec_update(st, cov); pw_g, var_g = g.pvec_update(st, cov)
    assert np.allclose(pw_g, pw_o, rtol=1e-14) and np.allclose(var_g, var_o, rtol=1e-11, atol=1e-20)
    cl_g = vx.build_clusters(pw_g, cell_ptr); cl_o = O.build_clusters(pw_o, cell_ptr)
    assert np.array_equal(cl_g, cl_o)
    ev_g, U_g = vx.plane_fit(cl_g); ev_o, U_o = O.plane_fit(cl_o)
    assert np.allclose(ev_g, ev_o, rtol=1e-9, atol=1e-13)
    ca_g = vx.cov_add_build(pw_g, var_g, cell_ptr); ca_o = O.cov_add_build(pw_o, var_o, cell_ptr)
    assert np.allclose(ca_g, ca_o, rtol=1e-12, atol=1e-22) and np.allclose(ca_g, np.transpose(ca_g, (0, 2, 1)), rtol=1e-12, atol=1e-22)
    # plane_update is quadratic in the eigenvectors: signs cancel, so both sides may use their own decomposition
    pl_g = vx.plane_update(cl_g, ev_g, U_g, ca_g); pl_o = O.plane_update(cl_o, ev_o, U_o, ca_o)
    sgn = np.sign(np.sum(pl_g["normal"] * pl_o["normal"], axis=1))
    assert np.all(np.abs(sgn) == 1) and np.allclose(pl_g["normal"] * sgn[:, None], pl_o["normal"], atol=1e-9)
    assert np.allclose(pl_g["center"], pl_o["center"], rtol=1e-15) and np.array_equal(pl_g["radius"], pl_o["radius"])
    S = np.ones((carry.size, 6)); S[:, :3] = sgn[:, None]
    pv_g = pl_g["plane_var"] * S[:, :, None] * S[:, None, :]
    scale = np.abs(pl_o["plane_var"]).max(axis=(1, 2), keepdims=True)
    assert np.all(np.abs(pv_g - pl_o["plane_var"]) <= 1e-7 * scale)       # 1/(lambda_0 - lambda_k) amplifies the eigen-solvers' 1e-13
    # the fitted planes recover the generating ones, and a scan is localised against the GPU-built map
    assert np.median(np.abs(np.sum(pl_g["normal"] * pm.normal[carry], axis=1))) > 0.99
    g2 = vx.LioEstimator(pm.voxel_size, pm.max_layer); o2 = O.LioOracle(pm.voxel_size, pm.max_layer)
    for e, pl in ((g2, pl_g), (o2, pl_o)):
        e.map_update(pm.loc[carry], pm.layer[carry], pm.path[carry], pl["center"], pl["normal"], pl["plane_var"], pl["radius"])
    sc = synth.make_lio_scan(pm, n_points=15000, seed=2602)
    g2.var_init(sc.xyz); o2.var_init(sc.xyz)
    rg = g2.lio_state_estimation(sc.state_init, sc.cov); ro = o2.lio_state_estimation(sc.state_init, sc.cov)
    assert rg["iterations"] == ro["iterations"] and abs(rg["match_num"] - ro["match_num"]) <= 2
    et, er = synth.pose_errors(rg["state"][None, :12], ro["state"][None, :12])
    assert et < 1e-6 and er < 1e-6
    e0 = synth.pose_errors(sc.state_init[None, :12], sc.state_gt[None, :12]); e1 = synth.pose_errors(rg["state"][None, :12], sc.state_gt[None, :12])
    assert e1[0] < 0.3 * e0[0]


def test_down_sampling_voxel_is_bit_identical(vx):
    """The voxel-grid filter run on every raw scan and every merged submap: same voxels, same float running means, bit for bit."""
    rng = np.random.default_rng(2700)
    for n, size, scale in ((200_000, 0.1, 30.0), (50_000, 0.125, 5.0), (3000, 2.0, 100.0), (1, 0.5, 1.0)):
        xyz = (rng.normal(size=(n, 3)) * scale).astype(np.float32)
        xyz[: n // 10] = np.round(xyz[: n // 10] / size) * size            # points sitting on voxel faces, also at negative integers
        got = vx.down_sampling_voxel(xyz, size); ref = O.down_sampling_voxel(xyz, size)
        assert got.shape == ref.shape and got.shape[0] <= n
        assert np.array_equal(got, ref)
    # dense duplicates: long running-mean chains
    xyz = np.repeat(rng.uniform(-1, 1, (50, 3)).astype(np.float32), 400, axis=0) + rng.normal(0, 1e-3, (20000, 3)).astype(np.float32)
    assert np.array_equal(vx.down_sampling_voxel(xyz, 0.5), O.down_sampling_voxel(xyz, 0.5))
    # pass-through below 1 mm, empty input, index range
    assert np.array_equal(vx.down_sampling_voxel(xyz[:100], 1e-4), xyz[:100])
    assert vx.down_sampling_voxel(np.zeros((0, 3), dtype=np.float32), 0.5).shape == (0, 3)
    with pytest.raises(vx.VxbaError):
        vx.down_sampling_voxel(np.array([[3e6, 0, 0]], dtype=np.float32), 0.5)


def test_map_random_update_sequences_match_a_rebuilt_oracle_map(vx):
    """Random upsert / remove / re-insert batches (table growth and re-hashing in between, records reused and appended): after every
    batch the GPU map must behave like an oracle map rebuilt from the surviving leaves."""
    pm = synth.make_plane_map(n_roots=3000, extent=8, seed=2800)
    sc = synth.make_lio_scan(pm, n_points=12000, seed=2801)
    n = len(pm.layer)
    rng = np.random.default_rng(2802)
    g = vx.LioEstimator(pm.voxel_size, pm.max_layer); g.var_init(sc.xyz)
    pnt, var = g.read_points()
    alive = np.zeros(n, dtype=bool)
    center = pm.center.copy()
    order = rng.permutation(n)
    cuts = [0, 5, 400, 1500, 1501, 4000, n]
    for step, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        batch = order[a:b]
        # new leaves, plus an in-place update of some old ones and the removal of others
        upd = rng.choice(np.nonzero(alive)[0], size=min(200, int(alive.sum())), replace=False) if alive.any() else np.zeros(0, dtype=int)
        rem = rng.choice(np.nonzero(alive)[0], size=min(150, int(alive.sum())), replace=False) if alive.any() else np.zeros(0, dtype=int)
        rem = np.setdiff1d(rem, upd)
        center[upd] += 1e-3
        idx = np.concatenate([batch, upd, rem]).astype(int)
        isp = pm.is_plane[idx].copy(); isp[len(batch) + len(upd):] = 0
        g.map_update(pm.loc[idx], pm.layer[idx], pm.path[idx], center[idx], pm.normal[idx], pm.plane_var[idx], pm.radius[idx], isp)
        alive[batch] = pm.is_plane[batch] == 1
        alive[rem] = False
        keep = np.nonzero(alive)[0]
        o = O.LioOracle(pm.voxel_size, pm.max_layer)
        o.map_update(pm.loc[keep], pm.layer[keep], pm.path[keep], center[keep], pm.normal[keep], pm.plane_var[keep], pm.radius[keep])
        o.set_points(pnt, var)
        ro = o.sweep(sc.state_gt, sc.cov, want_points=True); rg = g.sweep(sc.state_gt, sc.cov, want_points=True)
        assert np.array_equal(ro["plane_of_point"] >= 0, rg["plane_of_point"] >= 0), step
        if ro["match_num"]:
            check_sweep(rg, ro)
    assert g.map_size()[0] <= 3000


def test_leaf_stats_from_the_resident_scan_match_the_staged_chain(vx):
    """cut_voxel's per-leaf increments (PointCluster of the new points, cov_add) computed from the world points / covariances pvec_update
    left on the device, with only the host tree's bucketing crossing the boundary: identical to pvec_update -> gather -> K1 -> cov_add on
    the oracle, for a shuffled bucketing with empty leaves and points that belong to no leaf."""
    pm = synth.make_plane_map(n_roots=1500, extent=8, seed=2700)
    sc = synth.make_lio_scan(pm, n_points=40000, seed=2701)
    g = vx.LioEstimator(pm.voxel_size, pm.max_layer); o = O.LioOracle(pm.voxel_size, pm.max_layer)
    with pytest.raises(vx.VxbaError):
        g.leaf_stats(np.array([0, 0]), np.zeros(0, dtype=np.int32))                    # no scan, no world points yet
    g.var_init(sc.xyz); o.var_init(sc.xyz)
    with pytest.raises(vx.VxbaError):
        g.leaf_stats(np.array([0, 1]), np.zeros(1, dtype=np.int32))                    # scan loaded but pvec_update not called
    cov = sc.cov
    pw_o, var_o = o.pvec_update(sc.state_gt, cov)
    pw_g = g.pvec_update(sc.state_gt, cov, with_var=False)
    assert np.allclose(pw_g, pw_o, rtol=1e-14, atol=1e-14)
    rng = np.random.default_rng(2702)
    n = pw_g.shape[0]
    cell = np.floor(pw_g / pm.voxel_size).astype(np.int64)
    keep = rng.random(n) < 0.9                                                           # a tenth of the points land in no leaf
    idx = np.nonzero(keep)[0]
    perm = idx[np.lexsort((rng.random(idx.size), cell[idx, 2], cell[idx, 1], cell[idx, 0]))]   # bucketed by voxel, shuffled inside a bucket
    _, first = np.unique(cell[perm], axis=0, return_index=True)
    starts = np.sort(first)
    cell_ptr = np.concatenate([starts, [perm.size]]).astype(np.int64)
    cell_ptr = np.insert(cell_ptr, [3, 3, len(cell_ptr) - 1], [cell_ptr[3], cell_ptr[3], cell_ptr[-1]])   # three empty leaves
    cl_g, ca_g = g.leaf_stats(cell_ptr, perm)
    cl_o = O.build_clusters(np.ascontiguousarray(pw_g[perm]), cell_ptr)
    ca_o = O.cov_add_build(np.ascontiguousarray(pw_o[perm]), np.ascontiguousarray(var_o[perm]), cell_ptr)
    assert np.array_equal(cl_g, cl_o)                                                    # sequential unfused sums: bit-identical
    scale = np.maximum(np.abs(ca_o).max(axis=(1, 2), keepdims=True), 1e-300)
    assert np.all(np.abs(ca_g - ca_o) <= 1e-11 * scale)
    empty = np.diff(cell_ptr) == 0
    assert empty.sum() == 3 and np.all(cl_g[empty] == 0) and np.all(ca_g[empty] == 0)
    with pytest.raises(vx.VxbaError):
        g.leaf_stats(np.array([0, 1]), np.array([n], dtype=np.int32))                  # index outside the scan
    g.var_init(sc.xyz[:100])                                                             # a new scan invalidates the resident world points
    with pytest.raises(vx.VxbaError):
        g.leaf_stats(np.array([0, 1]), np.zeros(1, dtype=np.int32))
