"""Full-size parity (BASELINE.json configs[1] / configs[2] and SURVEY 8d's secondary runs): the LM loop of the HIP library against the
CPU checker on the same window AT THE BENCHMARKED SIZE -- cfg2 sparse incidence, cfg2 with fix clusters, cfg3 in mixed precision, and
a cfg2-sized window whose first steps are rejected.  The checker is the oracle restatement and, when oracle/_ref/libref.so travelled
with the snapshot, the reference's own Lidar_BA_Optimizer on top of it (tests/_ref.py) -- both must agree with the GPU.
Tolerance: identical accept/reject + Hessian-recompute schedule, poses within 1e-7 m / 1e-7 rad in fp64 (contract 1e-4), 1e-6 mixed."""
import numpy as np
import pytest

from tests import _oracle as O
from tests import _ref
from voxel_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vx():
    from voxel_slam_amd import vxba
    vxba.load_library()
    return vxba


def checkers():
    out = [("oracle", O)]
    R = _ref.backend()
    if R is not None:
        out.append(("reference", R))
    return out


def run_case(vx, sc, iters, precision="f64", tol=1e-7, need_reject=False, thd=8, options=None):
    fg = vx.LidarFactor(sc.win_size)
    for name, value in (options or {}).items():
        fg.set_option(name, value)
    fg.push_points(sc.n_voxels, sc.points_body, sc.cell_ptr, sc.fix, sc.coe)        # K1 on the device, as the bench does
    clusters = fg.read_clusters()
    fg.evaluate_only_residual(sc.poses_init)
    fg.set_precision(precision)
    got = vx.Lidar_BA_Optimizer().damping_iter(sc.poses_init, fg, max_iter=iters)
    fg.set_precision("f64")
    if need_reject:
        assert (got["trace"][:, 6] == 0).any() and (got["trace"][:, 6] == 1).any(), "the case must reject and accept steps"
    ev_g, U_g, mg_g = fg.read_cache()
    for name, B in checkers():
        fo = B.Oracle(sc.win_size)
        fo.push_voxels(clusters, sc.fix, sc.coe)
        fo.evaluate_only_residual(sc.poses_init)
        ref = fo.damping_iter(sc.poses_init, max_iter=iters, thd_num=thd)
        assert got["trace"].shape == ref["trace"].shape, name
        assert np.array_equal(got["trace"][:, 6:], ref["trace"][:, 6:]), name                # accepted / recomputed_hess per iteration
        assert np.allclose(got["trace"][:, :2], ref["trace"][:, :2], rtol=1e-9 if precision == "f64" else 1e-6), name
        if precision == "f64":
            assert np.allclose(got["trace"][:, 2], ref["trace"][:, 2], rtol=1e-6), name      # damping schedule u
        et, er = synth.pose_errors(got["poses"], ref["poses"])
        assert et < tol and er < tol, (name, et, er)
        # the cache the map adopts afterwards (state of the LAST residual sweep, accepted or not)
        ev, U, mg = fo.read_cache()
        assert np.allclose(mg_g, mg, rtol=1e-9, atol=1e-7), name
        vb2 = np.sum((mg[:, 6:9] / mg[:, 9:10]) ** 2, axis=1, keepdims=True)
        assert np.all(np.abs(ev_g - ev) <= (1e-12 if precision == "f64" else 1e-9) * (vb2 + 1.0)), name
    return got


@pytest.mark.parametrize("fused", [1, 0])
def test_cfg2_dense_lm_at_the_bench_perturbation_matches_the_checkers(vx, fused):
    """BASELINE configs[1] exactly as bench.py runs it (dense incidence, the bench's 0.05 deg / 0.02 m perturbation, three iterations) against the
    restatement AND the reference's own Lidar_BA_Optimizer (libref.so) inside the suite -- the round-5 review found that comparison only in
    bench.py's cpu_baseline leg.  Both forms of the device loop: the fused residual + Hessian launch (default) and the three-launch iteration."""
    sc = synth.make_config("cfg2")
    assert sc.n_voxels == 50_000 and sc.win_size == 10
    got = run_case(vx, sc, iters=3, options={"fused_sweeps": fused})
    assert np.all(got["trace"][:, 6] == 1) and got["resis"][1] < got["resis"][0]


def test_cfg2_sparse_lm_matches_the_checkers(vx):
    sc = synth.make_config("cfg2_sparse")
    got = run_case(vx, sc, iters=3)
    assert got["resis"][1] < got["resis"][0]


def test_cfg2_fix_lm_matches_the_checkers(vx):
    sc = synth.make_config("cfg2_fix")
    assert (sc.fix[:, 9] > 0).mean() > 0.2
    run_case(vx, sc, iters=3)


def test_cfg3_mixed_precision_lm_matches_the_checkers(vx):
    sc = synth.make_config("cfg3")
    assert sc.n_voxels == 100_000 and sc.points_body.shape[0] == 2_000_000
    run_case(vx, sc, iters=3, precision="mixed", tol=1e-6)


def test_cfg3_fp64_lm_matches_the_checkers(vx):
    run_case(vx, synth.make_config("cfg3"), iters=3)


def test_cfg2_window_with_rejected_steps_matches_the_checkers(vx):
    """A cfg2 window started 0.2 deg / 0.03 m off (4x the bench's perturbation): the first four trial steps overshoot and are rejected
    (u *= v, v *= 2, no Hessian recompute, cache left at the rejected trial state), the next four are accepted.  (SURVEY 8d's 0.5 deg
    is outside the basin at this size -- voxels up to 100 m from the sensor: the reference's own LM rejects every step until the
    relative-change test stops it; checked on the oracle, nine iterations.)"""
    sc = synth.make_config("cfg2", rot_sigma_deg=0.2, trans_sigma=0.03)
    run_case(vx, sc, iters=8, need_reject=True)


def test_cfg4_single_gpu_lm_matches_the_oracle(vx):
    """BASELINE configs[3] is this 400k-voxel window sharded over eight GPUs; on one GPU it is the largest window the bench runs
    (13 workgroup steps per CU in the Hessian sweep instead of four) -- two LM iterations against the oracle restatement."""
    sc = synth.make_config("cfg4")
    assert sc.n_voxels == 400_000
    fg = vx.LidarFactor(sc.win_size)
    fg.push_points(sc.n_voxels, sc.points_body, sc.cell_ptr, sc.fix, sc.coe)
    clusters = fg.read_clusters()
    fg.evaluate_only_residual(sc.poses_init)
    got = vx.Lidar_BA_Optimizer().damping_iter(sc.poses_init, fg, max_iter=2)
    for name, B in checkers():      # the restatement and, when oracle/_ref/libref.so travelled, the reference's own Lidar_BA_Optimizer
        fo = B.Oracle(sc.win_size)
        fo.push_voxels(clusters, sc.fix, sc.coe)
        fo.evaluate_only_residual(sc.poses_init)
        ref = fo.damping_iter(sc.poses_init, max_iter=2, thd_num=16)
        assert np.array_equal(got["trace"][:, 6:], ref["trace"][:, 6:]), name
        assert np.allclose(got["trace"][:, :2], ref["trace"][:, :2], rtol=1e-9), name
        et, er = synth.pose_errors(got["poses"], ref["poses"])
        assert et < 1e-7 and er < 1e-7, (name, et, er)
        Hg = got["hess"]; Ho = ref["hess"]
        assert np.allclose(Hg, Ho, rtol=0, atol=1e-9 * np.abs(Ho).max()), name
        del fo
