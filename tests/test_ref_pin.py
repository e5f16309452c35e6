"""Pins the oracle (oracle/vxo_*.hpp, the restatement every GPU parity test checks against) to THE REFERENCE'S OWN CODE:
oracle/_ref/libref.so = /root/reference/VoxelSLAM/src/{tools,preintegration,voxel_map}.hpp, unmodified, compiled where they lie
(`make -C oracle ref`, oracle/ref_capi.cpp) against the Eigen/PCL/ROS API shim under oracle/shim/ (or a real Eigen when one is
installed).  Both libraries export the same vxo_* surface, so every case runs the same inputs through both and compares outputs.

What "equal" means here: the two differ only in the association of a few sums (the restatement accumulates some 3-term products in
a different order than the reference's expression text evaluated by the shim), so cluster sums, eigen-decompositions, gradients and
accept/reject decisions come out identical and Hessians / poses agree to a few ulps -- the assertions below say exactly how far.

"Bit-identical" in this file therefore means: bit-identical to the reference's text AS THE SHIM EVALUATES IT -- every Eigen expression eagerly,
left to right, in natural order.  Real Eigen 3.3.7 may associate some products differently (lazy expression templates, vectorised reductions);
the image has no Eigen to run the pin against, and a real one is picked up automatically when installed (oracle/Makefile).  The three numerical
algorithms upstream takes from Eigen (SelfAdjointEigenSolver, LDLT, inverse) are the oracle's restatements on BOTH sides of these comparisons,
checked against LAPACK in tests/test_oracle_math.py (the round-5 review's caveat).

libref.so is git-ignored but travels to the GPU box with the snapshot; where it is neither prebuilt nor buildable (no
/root/reference and nothing under oracle/_ref/) the module is skipped, and the committed tests/golden/*.npz -- generated FROM
libref.so by tests/golden/make_golden.py -- carry the pin instead (tests/test_golden.py).
"""
import numpy as np
import pytest

from tests import _oracle as O
from tests import _ref
from tests.test_oracle_octree import PRM, point_vars, to_world
from voxel_slam_amd import synth

R = _ref.backend()
pytestmark = pytest.mark.skipif(R is None, reason="oracle/_ref/libref.so not available (needs /root/reference or a prebuilt copy)")

EPS = np.finfo(np.float64).eps


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


# ---- building blocks (tools.hpp:51-133, 304-365; voxel_map.hpp:91-106, 1118-1146, 1161-1163) -------------------------------------
def test_so3_helpers_bit_identical():
    rng = np.random.default_rng(1)
    for a in list(rng.normal(0, 1.0, (50, 3))) + list(rng.normal(0, 1e-6, (10, 3))) + [np.zeros(3), np.array([1e-12, 0, 0])]:
        Ro, Rr = O.exp_so3(a), R.exp_so3(a)
        assert np.array_equal(Ro, Rr)
        lo = np.zeros(3); lr = np.zeros(3)
        O.lib().vxo_log(np.ascontiguousarray(Ro.T.reshape(9)), lo); R.lib().vxo_log(np.ascontiguousarray(Rr.T.reshape(9)), lr)
        assert np.array_equal(lo, lr)
        assert np.array_equal(O.jr(a), R.jr(a))
        assert np.array_equal(O.jr_inv(Ro), R.jr_inv(Rr))


def test_cluster_push_transform_planefit_bit_identical():
    sc = synth.make_scene(win_size=3, pts_per_scan=3000, n_voxels=150, seed=77)
    co, cr = O.build_clusters(sc.points_body, sc.cell_ptr), R.build_clusters(sc.points_body, sc.cell_ptr)
    assert np.array_equal(co, cr) and co[:, 9].sum() == sc.points_body.shape[0]
    for k in range(0, co.shape[0], 7):
        assert np.array_equal(O.cluster_transform(co[k], sc.poses_init[k % 3]), R.cluster_transform(cr[k], sc.poses_init[k % 3]))
    big = co[co[:, 9] >= 5]
    (evo, Uo), (evr, Ur) = O.plane_fit(big), R.plane_fit(big)
    assert np.array_equal(evo, evr) and np.array_equal(Uo, Ur)


def test_cov_add_and_plane_update_match():
    rng = np.random.default_rng(5)
    n_cells, per = 40, 30
    pts = rng.normal(0, 1, (n_cells * per, 3)) * np.array([1.0, 1.0, 0.02]) + rng.uniform(-20, 20, (n_cells, 1, 3)).repeat(per, 1).reshape(-1, 3)
    var = point_vars(pts.shape[0], 3)
    cp = np.arange(0, n_cells * per + 1, per)
    ca_o, ca_r = O.cov_add_build(pts, var, cp), R.cov_add_build(pts, var, cp)
    assert rel(ca_o, ca_r) < 4 * EPS
    cl = O.build_clusters(pts, cp)
    ev, U = O.plane_fit(cl)
    po, pr = O.plane_update(cl, ev, U, ca_o), R.plane_update(cl, ev, U, ca_o)
    assert np.array_equal(po["center"], pr["center"]) and np.array_equal(po["normal"], pr["normal"]) and np.array_equal(po["radius"], pr["radius"])
    assert rel(po["plane_var"], pr["plane_var"]) < 1e-12


# ---- LidarFactor + Lidar_BA_Optimizer (voxel_map.hpp:109-444) -------------------------------------------------------------------------
CASES = {
    "w5_dense": dict(win_size=5, pts_per_scan=1500, n_voxels=96, p_obs=1.0, fix_frac=0.0, seed=9001, rot_sigma_deg=0.2, trans_sigma=0.03),
    "w10_sparse_fix": dict(win_size=10, pts_per_scan=2500, n_voxels=130, p_obs=0.7, fix_frac=0.3, seed=9002, rot_sigma_deg=0.1, trans_sigma=0.02),
    "w3_ragged": dict(win_size=3, pts_per_scan=400, n_voxels=41, p_obs=0.8, fix_frac=0.5, seed=9003, rot_sigma_deg=0.3, trans_sigma=0.05),
    "cfg1": dict(win_size=5, pts_per_scan=20_000, n_voxels=5_000, seed=synth.MASTER_SEED + 1),
}


def run_factor(B, sc, coe, max_iter, thd_num):
    f = B.Oracle(sc.win_size)
    f.push_voxels(sc.clusters, sc.fix, coe)
    r0 = f.evaluate_only_residual(sc.poses_init)
    ev, U, m = f.read_cache()
    H, J, r = f.acc_evaluate2(sc.poses_init)
    Hd, Jd, rd = f.divide_thread(sc.poses_init, thd_num=thd_num)
    lm = f.damping_iter(sc.poses_init, max_iter=max_iter, thd_num=thd_num)
    evf, Uf, mf = f.read_cache()
    return dict(r0=r0, ev=ev, U=U, m=m, H=H, J=J, r=r, Hd=Hd, Jd=Jd, rd=rd, lm=lm, evf=evf, Uf=Uf, mf=mf)


@pytest.mark.parametrize("name", list(CASES))
def test_lidar_factor_and_lm_match_the_reference(name):
    sc = synth.make_scene(**CASES[name])
    coe = np.linspace(0.5, 1.5, sc.n_voxels)
    o, r = run_factor(O, sc, coe, 4, 2), run_factor(R, sc, coe, 4, 2)
    # K2: residual, eigen-decomposition and merged clusters bit for bit
    assert o["r0"] == r["r0"] and np.array_equal(o["ev"], r["ev"]) and np.array_equal(o["U"], r["U"]) and np.array_equal(o["m"], r["m"])
    # K3: gradient and cached residual bit for bit; Hessian to a couple of ulps of its largest entry
    assert o["r"] == r["r"] and rel(o["J"], r["J"]) < 8 * EPS
    assert rel(o["H"], r["H"]) < 8 * EPS and rel(o["Hd"], r["Hd"]) < 8 * EPS and rel(o["Jd"], r["Jd"]) < 8 * EPS and abs(o["rd"] - r["rd"]) <= 4 * EPS * abs(r["rd"])
    # LM: same number of iterations, same accept/reject and Hessian-recompute flags, same damping schedule, poses to 1e-13
    to, tr = o["lm"]["trace"], r["lm"]["trace"]
    assert to.shape == tr.shape and np.array_equal(to[:, 6:8], tr[:, 6:8])
    assert np.allclose(to[:, :4], tr[:, :4], rtol=1e-12, atol=0) and np.allclose(to[:, 5], tr[:, 5], rtol=1e-9)
    assert np.allclose(to[:, 4], tr[:, 4], rtol=0, atol=1e-13 * abs(tr[0, 0]))          # q = residual1 - residual2 (cancellation)
    assert o["lm"]["is_converge"] == r["lm"]["is_converge"]
    et, er = synth.pose_errors(o["lm"]["poses"], r["lm"]["poses"])
    assert et < 1e-13 and er < 1e-13
    assert rel(o["lm"]["hess"], r["lm"]["hess"]) < 1e-12 and np.allclose(o["lm"]["resis"], r["lm"]["resis"], rtol=1e-13)
    assert np.allclose(o["evf"], r["evf"], rtol=1e-9, atol=1e-15) and rel(o["mf"], r["mf"]) < 1e-13


def test_lm_with_rejected_steps_matches_the_reference():
    """A window whose first steps are rejected (large perturbation): the reject branch -- u *= v, v *= 2, no Hessian recompute,
    cache left at the rejected trial state -- is the reference's."""
    sc = synth.make_scene(win_size=6, pts_per_scan=6000, n_voxels=400, seed=4242, rot_sigma_deg=2.5, trans_sigma=0.25)
    coe = np.ones(sc.n_voxels)
    o, r = run_factor(O, sc, coe, 8, 2), run_factor(R, sc, coe, 8, 2)
    to, tr = o["lm"]["trace"], r["lm"]["trace"]
    assert to.shape == tr.shape and np.array_equal(to[:, 6:8], tr[:, 6:8])
    assert (tr[:, 6] == 0).sum() >= 1 and (tr[:, 6] == 1).sum() >= 1, "case must exercise both branches"
    assert np.allclose(to[:, :4], tr[:, :4], rtol=1e-10)
    et, er = synth.pose_errors(o["lm"]["poses"], r["lm"]["poses"])
    assert et < 1e-12 and er < 1e-12
    assert rel(o["mf"], r["mf"]) < 1e-12


# ---- IMU_PRE + LI_BA_Optimizer + LI_BA_OptimizerGravity (preintegration.hpp:11-303, voxel_map.hpp:446-864) --------------------------------
def li_scene(W, V, pts, seed):
    sc = synth.make_scene(win_size=W, pts_per_scan=pts, n_voxels=V, seed=seed)
    iw = synth.make_imu(sc, seed=seed + 1)
    bg, ba = iw.states_init[0, 15:18], iw.states_init[0, 18:21]
    return sc, iw, bg, ba


@pytest.mark.parametrize("W,V,pts", [(5, 500, 6000), (10, 1500, 20000), (2, 200, 3000)])
def test_inertial_half_matches_the_reference(W, V, pts):
    sc, iw, bg, ba = li_scene(W, V, pts, 500 + W)
    bo = O.imu_preintegrate(iw.samples, iw.noise_meas, iw.noise_walk, bg, ba)
    br = R.imu_preintegrate(iw.samples, iw.noise_meas, iw.noise_walk, bg, ba)
    assert rel(bo[:, :79], br[:, :79]) < 1e-13                      # deltas, bias Jacobians, dtime
    assert rel(bo[:, 79:], br[:, 79:]) < 1e-12                      # covariance propagation (225)
    for i in range(W - 1):
        ro, Jo, go = O.imu_evaluate(br[i], iw.states_init[i], iw.states_init[i + 1])
        rr, Jr, gr = R.imu_evaluate(br[i], iw.states_init[i], iw.states_init[i + 1])
        # the information matrix is the inverse of a covariance with condition ~1e9: products with it agree to ~1e-9 at best
        assert np.isclose(ro, rr, rtol=1e-8) and rel(Jo, Jr) < 1e-8 and rel(go, gr) < 1e-8
        rog, Jog, gog = O.imu_evaluate_g(br[i], iw.states_init[i], iw.states_init[i + 1])
        rrg, Jrg, grg = R.imu_evaluate_g(br[i], iw.states_init[i], iw.states_init[i + 1])
        assert np.isclose(rog, rrg, rtol=1e-8) and rel(Jog, Jrg) < 1e-8 and rel(gog, grg) < 1e-8
    fo = O.Oracle(W); fo.push_voxels(sc.clusters, sc.fix, sc.coe); fo.evaluate_only_residual(sc.poses_init)
    fr = R.Oracle(W); fr.push_voxels(sc.clusters, sc.fix, sc.coe); fr.evaluate_only_residual(sc.poses_init)
    Ho, Jo, ro = O.li_divide_thread(fo, iw.states_init, br, thd_num=5, imu_coef=1e-4)
    Hr, Jr, rr = R.li_divide_thread(fr, iw.states_init, br, thd_num=5, imu_coef=1e-4)
    assert rel(Ho, Hr) < 1e-8 and rel(Jo, Jr) < 1e-8 and np.isclose(ro, rr, rtol=1e-9)
    assert np.isclose(O.li_only_residual(fo, iw.states_init, br, 5, 1e-4), R.li_only_residual(fr, iw.states_init, br, 5, 1e-4), rtol=1e-9)
    # LI_BA_OptimizerGravity::divide_thread / only_residual under their own names (voxel_map.hpp:673-773)
    Hog, Jog, rog = O.li_divide_thread_gravity(fo, iw.states_init, br, thd_num=5, imu_coef=1e-4)
    Hrg, Jrg, rrg = R.li_divide_thread_gravity(fr, iw.states_init, br, thd_num=5, imu_coef=1e-4)
    assert Hog.shape == (15 * W + 3, 15 * W + 3) and rel(Hog, Hrg) < 1e-8 and rel(Jog, Jrg) < 1e-8 and np.isclose(rog, rrg, rtol=1e-9)
    assert np.abs(Hrg[-3:, :-3]).max() > 0 and np.allclose(Hrg[:15 * W, :15 * W][:6, :6], Hr[:6, :6], rtol=1e-9)    # gravity rows present; the pose blocks are the 15W system's
    assert np.isclose(O.li_only_residual_gravity(fo, iw.states_init, br, 5, 1e-4), R.li_only_residual_gravity(fr, iw.states_init, br, 5, 1e-4), rtol=1e-9)
    # LI_BA_Optimizer::damping_iter (3 iterations upstream)
    oo = O.li_damping_iter(fo, iw.states_init, br, max_iter=3, thd_num=5, imu_coef=1e-4)
    rr_ = R.li_damping_iter(fr, iw.states_init, br, max_iter=3, thd_num=5, imu_coef=1e-4)
    et, er = synth.pose_errors(oo["states"][:, :12], rr_["states"][:, :12])
    assert et < 1e-9 and er < 1e-9 and np.allclose(oo["states"][:, 12:], rr_["states"][:, 12:], atol=1e-8)
    assert np.allclose(oo["imus"][:, 67:79], rr_["imus"][:, 67:79], atol=1e-9)       # dbg / dba and their _buf copies
    assert rel(oo["hess"], rr_["hess"]) < 1e-7
    # gravity variant (2 iterations by default upstream)
    fo.evaluate_only_residual(sc.poses_init); fr.evaluate_only_residual(sc.poses_init)
    og = O.li_damping_iter_gravity(fo, iw.states_init, br, max_iter=2, thd_num=5, imu_coef=1e-4)
    rg = R.li_damping_iter_gravity(fr, iw.states_init, br, max_iter=2, thd_num=5, imu_coef=1e-4)
    et, er = synth.pose_errors(og["states"][:, :12], rg["states"][:, :12])
    assert et < 1e-9 and er < 1e-9 and np.allclose(og["states"][:, 12:], rg["states"][:, 12:], atol=1e-8)
    assert np.allclose(og["resis"], rg["resis"], rtol=1e-9) and rel(og["hess"], rg["hess"]) < 1e-7


# ---- the incremental local map: OctoTree / cut_voxel_multi / recut / tras_opt / margi (voxel_map.hpp:896-1639) ------------------------------
def by_id(lv):
    order = np.argsort(lv["node_id"], kind="stable")
    return {k: (v[order] if isinstance(v, np.ndarray) and v.shape[:1] == order.shape else v) for k, v in lv.items()}


def factor_arrays(lv):
    fac = np.nonzero(lv["opt_state"] >= 0)[0]
    return fac[np.argsort(lv["opt_state"][fac])]


@pytest.mark.parametrize("S,win,pts,seed", [(9, 4, 12000, 6), (12, 5, 8000, 11)])
def test_local_map_evolves_like_the_reference_octree(S, win, pts, seed):
    """Scan by scan through cut_voxel_multi -> multi_recut (+ tras_opt) -> damping_iter -> multi_margi -> ring shift, once on the
    restatement (oracle/vxo_octree.hpp + vxo_ba.hpp) and once on the reference's OctoTree + LidarFactor + Lidar_BA_Optimizer.  After
    every stage the two maps hold the same leaves (ids, layers, flags, point counts, stored-point counts), bit-identical body-frame
    window clusters, world clusters / plane records equal to rounding, and the factors handed to the BA are the same set."""
    xyz, fp, poses_gt, _ = synth.make_scans(win_size=S, pts_per_scan=pts, seed=synth.MASTER_SEED + 900 + seed)
    rng = np.random.default_rng(seed)
    var = point_vars(xyz.shape[0], seed)
    kw = dict(PRM); kw["max_points"] = 60          # small enough for the fix-cluster cap (voxel_map.hpp:1247-1277) to engage
    mo, mr = O.LocalMapOracle(win_size=win, **kw), R.LocalMapOracle(win_size=win, **kw)
    fo, fr = O.Oracle(win), R.Oracle(win)
    xo, xr = [], []
    win_count = windows = 0
    capped = subdivided = 0
    for k in range(S):
        pose = poses_gt[k].copy(); pose[9:12] += rng.normal(0, 0.01, 3)
        s = slice(fp[k], fp[k + 1])
        win_count += 1
        for m, f, xb in ((mo, fo, xo), (mr, fr, xr)):
            xb.append(pose.copy())
            f.clear()
            m.cut_voxel(win_count - 1, xyz[s], var[s], to_world(xb[-1], xyz[s]))
        a, b = by_id(mo.leaves()), by_id(mr.leaves())
        assert np.array_equal(a["node_id"], b["node_id"]) and np.array_equal(a["pcrs_local"], b["pcrs_local"]) and np.array_equal(a["n_points"], b["n_points"])
        # world clusters are bit-identical until the first BA: from then on they carry the optimiser's merged clusters, which differ in
        # the last bits because the two factors list their voxels in different orders (hash order vs insertion order)
        if windows == 0:
            assert np.array_equal(a["pcr_add"], b["pcr_add"])
        assert rel(a["pcr_add"], b["pcr_add"]) < 1e-11 and rel(a["cov_add"], b["cov_add"]) < 1e-11
        mo.recut(win_count, np.stack(xo), fo); mr.recut(win_count, np.stack(xr), fr)
        a, b = by_id(mo.leaves()), by_id(mr.leaves())
        assert np.array_equal(a["node_id"], b["node_id"]) and np.array_equal(a["layer"], b["layer"])
        for key in ("isexist", "is_plane", "has_sw", "n_point_fix", "n_points", "in_slide"):
            assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a["opt_state"] >= 0, b["opt_state"] >= 0) and fo.size() == fr.size()
        assert np.array_equal(a["pcrs_local"], b["pcrs_local"]) and np.array_equal(a["pcr_fix"][:, 9], b["pcr_fix"][:, 9])
        assert rel(a["pcr_add"], b["pcr_add"]) < 1e-11 and rel(a["pcr_fix"], b["pcr_fix"]) < 1e-11
        pl = a["is_plane"]
        if windows == 0:
            assert np.array_equal(a["eig_val"][pl], b["eig_val"][pl]) and np.array_equal(a["eig_vec"][pl], b["eig_vec"][pl])
        # afterwards: cov = P/N - c c^T cancels at |c|^2 eps, so eigenvalues of clusters that differ in the last bits agree to ~1e-12 absolute
        assert np.allclose(a["eig_val"][pl], b["eig_val"][pl], rtol=1e-9, atol=1e-11)
        subdivided = max(subdivided, int((a["layer"] > 0).sum()))
        if win_count < win:
            continue
        windows += 1
        assert fo.size() > 100
        # the factor voxels come out in different orders (insertion order vs unordered_map order): same set, same content
        ia, ib = factor_arrays(a), factor_arrays(b)
        assert np.array_equal(np.sort(a["node_id"][ia]), np.sort(b["node_id"][ib]))
        oo = fo.damping_iter(np.stack(xo), max_iter=3, thd_num=2)
        rr = fr.damping_iter(np.stack(xr), max_iter=3, thd_num=2)
        assert np.array_equal(oo["trace"][:, 6:8], rr["trace"][:, 6:8])
        et, er = synth.pose_errors(oo["poses"], rr["poses"])
        assert et < 1e-11 and er < 1e-11
        for m, f, xb, out in ((mo, fo, xo, oo), (mr, fr, xr, rr)):
            m.margi(win_count, out["poses"], f)
            m.slide(1)
            xb[:] = [p for p in out["poses"][1:]]
        win_count -= 1
        a, b = by_id(mo.leaves()), by_id(mr.leaves())
        assert np.array_equal(a["node_id"], b["node_id"])
        for key in ("isexist", "is_plane", "has_sw", "n_point_fix", "n_points", "in_slide", "last_num"):
            assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a["pcr_add"][:, 9], b["pcr_add"][:, 9]) and np.array_equal(a["pcr_fix"][:, 9], b["pcr_fix"][:, 9])
        assert rel(a["pcr_add"], b["pcr_add"]) < 1e-11 and rel(a["pcr_fix"], b["pcr_fix"]) < 1e-11
        upd = a["is_plane"] & (a["last_num"] == a["pcr_add"][:, 9]) & (a["last_num"] > 0)
        sgn = np.sign(np.sum(a["normal"][upd] * b["normal"][upd], axis=1))
        assert np.allclose(a["normal"][upd], b["normal"][upd] * sgn[:, None], atol=1e-9) and np.allclose(a["center"][upd], b["center"][upd], atol=1e-10)
        assert np.allclose(a["radius"][upd], b["radius"][upd], rtol=1e-6) and rel(a["plane_var"][upd], b["plane_var"][upd]) < 1e-6
        assert mo.counts() == mr.counts()
        capped = max(capped, int((a["pcr_fix"][:, 9] >= kw["max_points"]).sum()))
    assert windows == S - win + 1 and subdivided > 50 and capped > 0


def test_plane_match_of_the_reference_agrees_with_the_oracle_odometry():
    """`match` (voxel_map.hpp:1335-1392, 1674-1698) walked on the reference's own tree vs the oracle's lio sweep on the flattened plane
    map of the same tree: same matched points, same leaves, same sigma."""
    from tests.test_gpu_local_mapping_cycle import lio_leaf_args
    S, win, pts, seed = 5, 5, 15000, 21
    xyz, fp, poses_gt, _ = synth.make_scans(win_size=S, pts_per_scan=pts, seed=synth.MASTER_SEED + 900 + seed)
    var = point_vars(xyz.shape[0], seed)
    mo, mr = O.LocalMapOracle(win_size=win, **PRM), R.LocalMapOracle(win_size=win, **PRM)
    fo, fr = O.Oracle(win), R.Oracle(win)
    xs = []
    for k in range(S):
        xs.append(poses_gt[k].copy())
        s = slice(fp[k], fp[k + 1])
        for m, f in ((mo, fo), (mr, fr)):
            f.clear()
            m.cut_voxel(k, xyz[s], var[s], to_world(xs[-1], xyz[s]))
            m.recut(k + 1, np.stack(xs), f)
    for m, f in ((mo, fo), (mr, fr)):
        f.evaluate_only_residual(np.stack(xs))
        m.margi(S, np.stack(xs), f)                 # writes the plane records (plane_update) the odometry matches against
    lv = mo.leaves()
    args = lio_leaf_args(lv)
    assert args[0].shape[0] > 300
    # probe points: the last scan re-expressed with a slightly different pose
    rng = np.random.default_rng(3)
    pose = poses_gt[S - 1].copy(); pose[9:12] += rng.normal(0, 0.02, 3)
    s = slice(fp[S - 1], fp[S])
    pw = to_world(pose, xyz[s])
    oe = O.LioOracle(PRM["voxel_size"], PRM["max_layer"])
    oe.map_update(*args)
    oe.set_points(xyz[s], var[s])
    state = np.concatenate([pose, np.zeros(9), [0, 0, -9.8]])
    sw = oe.sweep(state, np.zeros((15, 15)), reset_cache=True, want_points=True)      # zero state covariance: var_wld = R var R^T as given below
    Rm = pose[:9].reshape(3, 3).T
    var_w = Rm @ var[s] @ Rm.T
    n = pw.shape[0]
    flag = np.zeros(n, dtype=np.int32); sig = np.zeros(n); lid = np.zeros(n, dtype=np.uint64)
    R.lib().vxo_localmap_match(mr._h, n, np.ascontiguousarray(pw), np.ascontiguousarray(np.transpose(var_w, (0, 2, 1)).reshape(n, 9)), flag, sig, lid)
    matched_o = sw["plane_of_point"] >= 0
    # the oracle recomputes the world point / covariance itself; a handful of knife-edge points may differ by rounding of R var R^T
    assert matched_o.sum() > 0.3 * n and np.mean(matched_o != (flag != 0)) < 2e-3
    both = matched_o & (flag != 0)
    sel = np.nonzero(lv["is_plane"] & (lv["last_num"] > 0))[0]
    assert np.array_equal(lv["node_id"][sel][sw["plane_of_point"][both]], lid[both])
    assert np.allclose(sw["sigma_of_point"][both], sig[both], rtol=1e-6)


def test_lio_state_estimation_restatement_matches_the_reference_text():
    """VOXEL_SLAM::lio_state_estimation (voxelslam.cpp:856-958) lives in a ROS translation unit; its TEXT is extracted at build time
    (oracle/Makefile: _ref/extracted/lio_state_estimation.inc, never committed) and compiled inside a harness in libref.so that supplies
    x_curr and surf_map.  The restatement every GPU odometry test is checked against (oracle/vxo_lio.hpp) must reproduce it: same verdict,
    same state after the iterated update, same posterior covariance -- on the reference's own tree (its `match`, its OctoTree) for the
    reference and on the flattened plane map of the oracle's tree for the restatement."""
    if not hasattr(R.lib(), "vxo_ref_lio_state_estimation"):
        pytest.skip("libref.so was built without the extraction (no voxelslam.cpp at build time)")
    from tests.test_gpu_local_mapping_cycle import lio_leaf_args
    S, win, pts, seed = 5, 5, 15000, 23
    xyz, fp, poses_gt, _ = synth.make_scans(win_size=S, pts_per_scan=pts, seed=synth.MASTER_SEED + 900 + seed)
    var = point_vars(xyz.shape[0], seed)
    mo, mr = O.LocalMapOracle(win_size=win, **PRM), R.LocalMapOracle(win_size=win, **PRM)
    fo, fr = O.Oracle(win), R.Oracle(win)
    xs = []
    for k in range(S - 1):                                   # the map holds scans 0 .. S-2; the odometry aligns scan S-1 against it
        xs.append(poses_gt[k].copy())
        s = slice(fp[k], fp[k + 1])
        for m, f in ((mo, fo), (mr, fr)):
            f.clear()
            m.cut_voxel(k, xyz[s], var[s], to_world(xs[-1], xyz[s]))
            m.recut(k + 1, np.stack(xs), f)
    for m, f in ((mo, fo), (mr, fr)):
        f.evaluate_only_residual(np.stack(xs))
        m.margi(S - 1, np.stack(xs), f)                      # plane_update: the records `match` reads
    args = lio_leaf_args(mo.leaves())
    assert args[0].shape[0] > 200
    rng = np.random.default_rng(5)
    s = slice(fp[S - 1], fp[S])
    n = fp[S - 1 + 1] - fp[S - 1]
    cov = np.diag(np.concatenate([np.full(3, 1e-4), np.full(3, 1e-3), np.full(3, 1e-2), np.full(3, 1e-6), np.full(3, 1e-4)]))
    cov[0:3, 3:6] = cov[3:6, 0:3] = 2e-5 * np.eye(3)
    checked = 0
    for trial, (dr, dp) in enumerate(((0.003, 0.03), (0.001, 0.01), (0.02, 0.3))):      # the last start is far off: few matches, both must say so alike
        pose = poses_gt[S - 1].copy()
        pose[9:12] += rng.normal(0, dp, 3)
        Rm = pose[:9].reshape(3, 3).T @ synth.rodrigues(rng.normal(0, dr, 3))
        pose[:9] = Rm.T.reshape(9)
        state = np.concatenate([pose, rng.normal(0, 0.1, 3), rng.normal(0, 1e-3, 3), rng.normal(0, 1e-2, 3), [0, 0, -9.8]])
        oe = O.LioOracle(PRM["voxel_size"], PRM["max_layer"])
        oe.map_update(*args)
        oe.set_points(xyz[s], var[s])
        got = oe.lio_state_estimation(state, cov)
        st_r = state.copy(); cv_r = np.ascontiguousarray(cov.T.copy())
        import ctypes as C
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        pnt_c = np.ascontiguousarray(xyz[s], dtype=np.float64); var_c = np.ascontiguousarray(np.transpose(var[s], (0, 2, 1)).reshape(n, 9), dtype=np.float64)
        ok_r = R.lib().vxo_ref_lio_state_estimation(mr._h, vp(st_r), vp(cv_r), C.c_int64(n), vp(pnt_c), vp(var_c))
        assert bool(ok_r) == got["ok"], trial
        assert np.allclose(got["state"], st_r, rtol=0, atol=1e-9), (trial, np.abs(got["state"] - st_r).max())
        assert np.allclose(got["cov"], cv_r.T, rtol=1e-7, atol=1e-13), trial
        assert not np.allclose(st_r[:12], state[:12], atol=1e-6)       # the update moved the pose
        checked += 1
    assert checked == 3


# ---- OctreeGBA batch factor construction (loop_refine.hpp:273-537) and the voxel-grid filter (tools.hpp:201-238) -------------------------
def _content_order(v):
    """Canonical order of factor voxels by content (the reference's voxels carry no id): merged cluster N, then its sums."""
    m = v["merged"]
    return np.lexsort((m[:, 8], m[:, 7], m[:, 6], m[:, 9]))


@pytest.mark.parametrize("W,pts,seed", [(4, 6000, 1), (7, 8000, 2)])
def test_octree_gba_voxelisation_matches_the_reference(W, pts, seed):
    xyz, fp, poses, _ = synth.make_scans(win_size=W, pts_per_scan=pts, seed=synth.MASTER_SEED + 800 + seed)
    xyz = xyz.astype(np.float32).astype(np.float64)          # upstream's scan points are pcl floats
    params = np.array([1.0, 2, 10, 0.02, 1 / 16, 1 / 16, 1 / 9, 1 / 9, 0.12, 0, 0, 0, 0, 2], dtype=np.float64)
    o, r = O.voxelize(W, xyz, fp, poses, params), R.voxelize(W, xyz, fp, poses, params)
    assert o["node_id"].shape[0] == r["node_id"].shape[0] > 100
    io, ir = _content_order(o), _content_order(r)
    assert np.array_equal(o["clusters"][io], r["clusters"][ir])          # body-frame cell clusters: bit-identical sums
    assert np.array_equal(o["merged"][io], r["merged"][ir])              # world-frame node clusters
    assert np.array_equal(o["eig_val"][io], r["eig_val"][ir]) and np.array_equal(o["eig_vec"][io], r["eig_vec"][ir])


def test_down_sampling_voxel_matches_the_reference():
    rng = np.random.default_rng(12)
    xyz = (rng.uniform(-30, 30, (40000, 3)) * np.array([1, 1, 0.2])).astype(np.float32)
    xyz[:50] = np.round(xyz[:50])                           # exact voxel faces incl. negative integers (the float-index quirk)
    for vs in (0.5, 0.1):
        o, r = O.down_sampling_voxel(xyz, vs), R.down_sampling_voxel(xyz, vs)
        assert o.shape == r.shape and o.shape[0] < xyz.shape[0]
        so = o[np.lexsort((o[:, 2], o[:, 1], o[:, 0]))]; sr = r[np.lexsort((r[:, 2], r[:, 1], r[:, 0]))]
        assert np.array_equal(so, sr)
