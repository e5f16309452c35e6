"""GPU parity of the LiDAR-inertial BA (LI_BA_Optimizer, voxel_map.hpp:446-655): voxel sweeps on the GPU + the host-side
inertial half, against the CPU oracle's LI_BA_Optimizer on the same window, IMU stream and initial states.
Tolerances as in test_gpu_parity.py; the IMU information matrices (condition ~1e9) bound the joint system at ~1e-6."""
import os

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


def build(vx, W, V, pts, seed, p_obs=1.0):
    sc = synth.make_scene(win_size=W, pts_per_scan=pts, n_voxels=V, p_obs=p_obs, seed=seed)
    iw = synth.make_imu(sc, seed=seed + 1)
    bg, ba = iw.states_init[0, 15:18], iw.states_init[0, 18:21]
    blobs = O.imu_preintegrate(iw.samples, iw.noise_meas, iw.noise_walk, bg, ba)
    facs = []
    for gyr, acc, dts in iw.samples:
        f = vx.IMU_PRE(bg, ba)
        for g, a, dt in zip(gyr, acc, dts):
            f.add_imu(g, a, dt, iw.noise_meas, iw.noise_walk)
        facs.append(f)
    fo = O.Oracle(W); fo.push_voxels(sc.clusters, sc.fix, sc.coe); fo.evaluate_only_residual(sc.poses_init)
    fg = vx.LidarFactor(W); fg.push_voxels(sc.clusters, sc.fix, sc.coe); fg.evaluate_only_residual(sc.poses_init)
    return sc, iw, blobs, facs, fo, fg


@pytest.mark.parametrize("W,V,pts", [(5, 500, 6000), (10, 3000, 40000), (2, 200, 3000)])
def test_joint_system_matches_oracle(vx, W, V, pts):
    sc, iw, blobs, facs, fo, fg = build(vx, W, V, pts, seed=500 + W)
    opt = vx.LI_BA_Optimizer(imu_coef=1e-4)
    H, J, r = opt.divide_thread(iw.states_init, fg, facs)
    Hr, Jr, rr = O.li_divide_thread(fo, iw.states_init, blobs, thd_num=5, imu_coef=1e-4)
    assert H.shape == (15 * W, 15 * W)
    assert np.allclose(H, H.T, rtol=1e-9, atol=1e-9 * np.abs(H).max())
    assert np.allclose(H, Hr, rtol=1e-6, atol=1e-8 * np.abs(Hr).max())
    assert np.allclose(J, Jr, rtol=1e-6, atol=1e-8 * np.abs(Jr).max())
    assert np.isclose(r, rr, rtol=1e-9)
    # the LiDAR blocks alone are reproduced to sweep precision
    lid = np.array([15 * i + k for i in range(W) for k in range(6)])
    H6, J6, r6 = fo.divide_thread(sc.poses_init, thd_num=5)
    Himu = np.zeros_like(H)
    for i in range(W - 1):
        _, jtj, _ = O.imu_evaluate(blobs[i], iw.states_init[i], iw.states_init[i + 1])
        Himu[15 * i:15 * i + 30, 15 * i:15 * i + 30] += 1e-4 * jtj
    assert np.allclose((H - Himu)[np.ix_(lid, lid)], H6, rtol=1e-7, atol=1e-7 * np.abs(H6).max())
    r2 = opt.only_residual(iw.states_init, fg, facs)
    assert np.isclose(r2, O.li_only_residual(fo, iw.states_init, blobs, 5, 1e-4), rtol=1e-10)


@pytest.mark.parametrize("mode", ["host_shell_queued_sweeps", "host_shell_queued_host_pose_solve", "host_shell_plain", "host_shell_dense_solve"])
@pytest.mark.parametrize("W,V,pts,iters", [(5, 800, 8000, 3), (10, 3000, 40000, 6), (2, 300, 4000, 4)])
def test_li_damping_iter_matches_oracle(vx, W, V, pts, iters, mode):
    """The four ways the library runs LI_BA_Optimizer::damping_iter, same contract each: the host shell with its sweeps queued and the reduced
    pose system solved inside the residual-sweep launch (default: velocities / biases eliminated on the host under the Hessian sweep, no kernel
    waits for the host), the same with the pose system solved on the host (trial poses fed to the waiting residual sweep through mapped host
    memory -- also what the default does after a rejected step), the host shell launching every sweep when its poses exist, and the same with
    the dense 15W LDL^T instead of the band / Schur solve.  (A fifth, the whole loop enqueued on the GPU, was removed in round 4: 4x slower.)"""
    sc, iw, blobs, facs, fo, fg = build(vx, W, V, pts, seed=600 + W)
    fg.set_option("li_queued_sweeps", 1 if mode.startswith("host_shell_queued") else 0)
    fg.set_option("li_device_pose_solve", 0 if mode == "host_shell_queued_host_pose_solve" else 1)   # default: the reduced pose system is solved inside the residual-sweep launch
    fg.set_option("li_structured_solve", 0 if mode == "host_shell_dense_solve" else 1)
    assert fg._L.vxba_set_option(fg._h, 3, 1) != 0 and fg._L.vxba_set_option(fg._h, 3, 0) == 0     # VXBA_OPT_LI_DEVICE_LOOP: reserved slot, only 0 is accepted
    ref = O.li_damping_iter(fo, iw.states_init, blobs, max_iter=iters, thd_num=5, imu_coef=1e-4)
    got = vx.LI_BA_Optimizer(imu_coef=1e-4).damping_iter(iw.states_init, fg, facs, max_iter=iters)
    assert got["trace"].shape == ref["trace"].shape
    assert np.array_equal(got["trace"][:, 6:], ref["trace"][:, 6:])                 # accept / reject / recompute sequence
    assert np.allclose(got["trace"][:, :2], ref["trace"][:, :2], rtol=1e-7)          # residual1 / residual2
    assert np.allclose(got["trace"][:, 2:4], ref["trace"][:, 2:4], rtol=1e-4)        # damping trajectory
    et, er = synth.pose_errors(got["states"][:, :12], ref["states"][:, :12])
    assert et < 1e-7 and er < 1e-7, (et, er)                                         # north_star: 1e-4 m / 1e-4 rad
    assert np.allclose(got["states"][:, 12:21], ref["states"][:, 12:21], atol=1e-6)  # v, bg, ba
    assert np.array_equal(got["states"][:, 21:], iw.states_init[:, 21:])             # gravity is not optimised
    gi = np.stack([f.blob for f in facs])
    assert np.allclose(gi[:, 67:79], ref["imus"][:, 67:79], atol=1e-7)               # dbg, dba and their _buf copies
    assert np.allclose(got["hess"], ref["hess"], rtol=1e-5, atol=1e-7 * np.abs(ref["hess"]).max())
    # the cache afterwards describes the last evaluated trial state, as upstream
    ev_g, _, m_g = fg.read_cache(); ev_o, _, m_o = fo.read_cache()
    assert np.allclose(m_g, m_o, rtol=1e-9, atol=1e-9) and np.allclose(ev_g, ev_o, rtol=1e-6, atol=1e-11)


@pytest.mark.parametrize("queued", [1, 0])
def test_li_shells_with_f32_cluster_rows(vx, queued):
    """VXBA_PRECISION_MIXED_F32_CLUSTERS under the LiDAR-inertial shell (the queued residual sweep with the host-fed first workgroup is the
    f32-row variant too): same accept / reject sequence as the fp64 oracle, states within the mixed-precision tolerance."""
    sc, iw, blobs, facs, fo, fg = build(vx, 10, 3000, 40000, seed=610)
    fg.set_option("li_queued_sweeps", queued)
    fg.set_precision("mixed_f32_clusters")
    fg.evaluate_only_residual(iw.states_init[:, :12])
    ref = O.li_damping_iter(fo, iw.states_init, blobs, max_iter=6, thd_num=5, imu_coef=1e-4)
    got = vx.LI_BA_Optimizer(imu_coef=1e-4).damping_iter(iw.states_init, fg, facs, max_iter=6)
    assert got["trace"].shape == ref["trace"].shape and np.array_equal(got["trace"][:, 6:], ref["trace"][:, 6:])
    assert np.allclose(got["trace"][:, :2], ref["trace"][:, :2], rtol=1e-4)
    assert not np.array_equal(got["trace"][:, 1], ref["trace"][:, 1])                 # really the f32 rows
    et, er = synth.pose_errors(got["states"][:, :12], ref["states"][:, :12])
    assert et < 1e-5 and er < 1e-5, (et, er)
    assert np.allclose(got["states"][:, 12:21], ref["states"][:, 12:21], atol=1e-4)


def test_li_ba_improves_on_lidar_only_velocity_and_bias(vx):
    """The inertial terms make velocity / bias observable: after LI-BA they are closer to the truth than the initial guess."""
    sc, iw, blobs, facs, fo, fg = build(vx, 10, 3000, 40000, seed=777)
    got = vx.LI_BA_Optimizer().damping_iter(iw.states_init, fg, facs, max_iter=8)
    e0 = synth.pose_errors(iw.states_init[:, :12], iw.states_gt[:, :12])
    e1 = synth.pose_errors(got["states"][:, :12], iw.states_gt[:, :12])
    assert e1[0] < e0[0] and e1[1] < e0[1]
    assert np.array_equal(got["states"][0], iw.states_init[0])


@pytest.mark.parametrize("queued", [1, 0])
@pytest.mark.parametrize("W,V,pts,iters", [(5, 800, 8000, 5), (10, 3000, 40000, 5)])
def test_gravity_variant_matches_oracle(vx, W, V, pts, iters, queued):
    """LI_BA_OptimizerGravity (voxel_map.hpp:658-864), max_iter = 5 as at its call site (voxelslam.cpp:1644)."""
    sc, iw, blobs, facs, fo, fg = build(vx, W, V, pts, seed=700 + W)
    st = iw.states_init.copy()
    st[:, 21:24] += [0.05, -0.03, 0.08]
    ref = O.li_damping_iter_gravity(fo, st, blobs, max_iter=iters, thd_num=5, imu_coef=1e-4)
    opt = vx.LI_BA_OptimizerGravity(imu_coef=1e-4)
    fg.set_option("li_queued_sweeps", queued)
    got = opt.damping_iter(st, fg, facs, max_iter=iters)
    assert got["hess"].shape == (15 * W + 3, 15 * W + 3)
    assert got["trace"].shape == ref["trace"].shape
    assert np.array_equal(got["trace"][:, 6:], ref["trace"][:, 6:])
    assert np.allclose(got["trace"][:, :2], ref["trace"][:, :2], rtol=1e-7)
    assert np.allclose(got["resis"], ref["resis"], rtol=1e-7)
    et, er = synth.pose_errors(got["states"][:, :12], ref["states"][:, :12])
    assert et < 1e-7 and er < 1e-7, (et, er)
    assert np.allclose(got["states"][:, 12:24], ref["states"][:, 12:24], atol=1e-6)   # v, bg, ba, g
    assert np.allclose(got["hess"], ref["hess"], rtol=1e-5, atol=1e-7 * np.abs(ref["hess"]).max())


def test_motion_init_round_from_raw_scans_matches_oracle(vx):
    """One round of motion_init (voxelslam.cpp:596-640) end to end: the window's raw scans -> OctoTree-criteria batch map build ->
    factor -> LI_BA_OptimizerGravity::damping_iter(3), on the GPU path and on the oracle; then the normals' spread test (:651-657)."""
    import types
    W = 6
    xyz, fp, poses, gt = synth.make_scans(win_size=W, pts_per_scan=25_000, extent=24.0, noise=0.005, seed=synth.MASTER_SEED + 880, rot_sigma_deg=0.05, trans_sigma=0.02)
    iw = synth.make_imu(types.SimpleNamespace(poses_gt=gt, poses_init=poses, win_size=W), seed=881)
    st = iw.states_init.copy()
    st[:, 21:24] += [0.05, -0.03, 0.08]
    bg, ba = st[0, 15:18], st[0, 18:21]
    blobs = O.imu_preintegrate(iw.samples, iw.noise_meas, iw.noise_walk, bg, ba)
    facs = []
    for gyr, acc, dts in iw.samples:
        f = vx.IMU_PRE(bg, ba)
        for g, a, dt in zip(gyr, acc, dts):
            f.add_imu(g, a, dt, iw.noise_meas, iw.noise_walk)
        facs.append(f)
    P = vx.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=20, min_eigen_value=0.02, eigen_ratio=(1 / 4, 1 / 4, 1 / 4, 1 / 4),
                          min_points_layer=(20, 20, 15, 10), min_frames=0)                     # motion_init's first-phase thresholds (:570-574)
    fg = vx.LidarFactor(W)
    ids = fg.voxelize_push(xyz, fp, st[:, :12].copy(), P)
    r = O.voxelize(W, xyz, fp, st[:, :12].copy(), P.as_array())
    order = np.lexsort((r["node_id"], (r["node_id"] & np.uint64(7)).astype(np.int64)))       # the GPU's push order
    assert np.array_equal(ids, r["node_id"][order]) and ids.size >= 10                          # `if(voxhess.plvec_voxels.size() < 10) break;`
    fo = O.Oracle(W)
    n = ids.size
    fo.push_voxels(r["clusters"][order], np.zeros((n, 10)), np.ones(n), r["eig_val"][order], r["eig_vec"][order], r["merged"][order])
    ref = O.li_damping_iter_gravity(fo, st, blobs, max_iter=3, thd_num=5, imu_coef=1e-4)
    got = vx.LI_BA_OptimizerGravity(imu_coef=1e-4).damping_iter(st, fg, facs, max_iter=3)
    assert np.array_equal(got["trace"][:, 6:], ref["trace"][:, 6:]) and np.allclose(got["resis"], ref["resis"], rtol=1e-7)
    et, er = synth.pose_errors(got["states"][:, :12], ref["states"][:, :12])
    assert et < 1e-7 and er < 1e-7, (et, er)
    assert np.allclose(got["states"][:, 12:24], ref["states"][:, 12:24], atol=1e-6)
    # degeneracy test on the factors' plane normals (voxhess.eig_vectors, :651-657)
    _, U_g, _ = fg.read_cache(); _, U_o, _ = fo.read_cache()
    n_g = U_g.reshape(n, 3, 3).transpose(0, 2, 1)[:, :, 0]          # eig_vectors[k].col(0): the plane normals (column-major 3x3 per voxel)
    n_o = U_o.reshape(n, 3, 3).transpose(0, 2, 1)[:, :, 0]
    nn_g = n_g.T @ n_g; nn_o = n_o.T @ n_o
    assert np.allclose(np.linalg.eigvalsh(nn_g), np.linalg.eigvalsh(nn_o), rtol=1e-6) and np.linalg.eigvalsh(nn_g)[0] > 15


@pytest.mark.parametrize("n_voxels", [300, 20000])
def test_queued_sweeps_back_to_back_solves_are_reproducible(vx, n_voxels):
    """Thousands of back-to-back LiDAR-inertial solves (both optimisers, 2-5 iterations) with the sweeps queued ahead of the host solve:
    every one must equal the first of its kind bit for bit.  The completion sentinels in mapped host memory are a timing matter -- a
    fill that can land after the reduction it guards shows up here within a few hundred calls on small windows (scripts/dbg_li_stress.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "dbg_li_stress.py"), "2500", str(n_voxels)], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
    assert "all identical" in r.stdout


@pytest.mark.parametrize("queued", [1, 0])
def test_li_edge_cases_one_iteration_and_rejected_steps(vx, queued):
    """max_iter = 1 (no speculative sweep is ever queued), a start so far off that steps are rejected (the wasted speculative reduction,
    the roll-back of the bias deltas, the re-solve with a larger damping), and an early convergence exit that leaves a speculative
    reduction behind for the next call to find -- all against the oracle, in both shell modes."""
    sc, iw, blobs, facs, fo, fg = build(vx, 6, 900, 9000, seed=811)
    fg.set_option("li_queued_sweeps", queued)
    ref = O.li_damping_iter(fo, iw.states_init, blobs, max_iter=1, thd_num=5, imu_coef=1e-4)
    got = vx.LI_BA_Optimizer(imu_coef=1e-4).damping_iter(iw.states_init, fg, facs, max_iter=1)
    assert got["trace"].shape == ref["trace"].shape == (1, 8)
    et, er = synth.pose_errors(got["states"][:, :12], ref["states"][:, :12])
    assert et < 1e-7 and er < 1e-7 and np.allclose(got["trace"][:, :2], ref["trace"][:, :2], rtol=1e-7)
    # a bad start: velocities and biases far off -> rejected steps on the way
    sc, iw, blobs, facs, fo, fg = build(vx, 5, 700, 7000, seed=812)
    fg.set_option("li_queued_sweeps", queued)
    st = iw.states_init.copy()
    rng = np.random.default_rng(5)
    st[:, 12:15] += rng.normal(0, 2.0, (5, 3)); st[:, 15:21] += rng.normal(0, 0.2, (5, 6)); st[1:, 9:12] += rng.normal(0, 0.3, (4, 3))
    ref = O.li_damping_iter(fo, st, blobs, max_iter=8, thd_num=5, imu_coef=1e-4)
    got = vx.LI_BA_Optimizer(imu_coef=1e-4).damping_iter(st, fg, facs, max_iter=8)
    assert got["trace"].shape == ref["trace"].shape
    assert np.array_equal(got["trace"][:, 6:], ref["trace"][:, 6:])
    assert np.allclose(got["trace"][:, :2], ref["trace"][:, :2], rtol=1e-6)
    et, er = synth.pose_errors(got["states"][:, :12], ref["states"][:, :12])
    assert et < 1e-6 and er < 1e-6, (et, er)
    # converge early (tiny steps from the optimum), then call again at once: the second call must not see the first one's leftovers
    sc, iw, blobs, facs, fo, fg = build(vx, 5, 700, 7000, seed=813)
    fg.set_option("li_queued_sweeps", queued)
    first = vx.LI_BA_Optimizer(imu_coef=1e-4).damping_iter(iw.states_init, fg, facs, max_iter=30)
    assert first["trace"].shape[0] < 30                                   # stopped on the convergence test
    blobs2 = np.stack([f.blob for f in facs])
    fo2 = O.Oracle(5); fo2.push_voxels(sc.clusters, sc.fix, sc.coe); fo2.evaluate_only_residual(first["states"][:, :12])
    fg.evaluate_only_residual(first["states"][:, :12])
    ref2 = O.li_damping_iter(fo2, first["states"], blobs2, max_iter=2, thd_num=5, imu_coef=1e-4)
    again = vx.LI_BA_Optimizer(imu_coef=1e-4).damping_iter(first["states"], fg, facs, max_iter=2)
    assert np.array_equal(again["trace"][:, 6:], ref2["trace"][:, 6:]) and np.allclose(again["trace"][:, :2], ref2["trace"][:, :2], rtol=1e-6)
    et, er = synth.pose_errors(again["states"][:, :12], ref2["states"][:, :12])
    assert et < 1e-7 and er < 1e-7


def test_information_matrices_follow_the_imu_factors_between_calls(vx):
    """The shells keep the 15 x 15 information matrices of the previous call and re-use one when a factor's covariance is bit-identical
    to one seen then (vxba_capi_li.hip, li_information_matrices).  Same LidarFactor, three IMU windows in turn -- A, B (other noise
    densities: other covariances), a window that mixes factors of both (the sliding-window case: known covariances in other slots), A
    again -- every call must equal the call of a factor that has never seen another window, bit for bit."""
    W, V, pts = 6, 1200, 12000
    sc = synth.make_scene(win_size=W, pts_per_scan=pts, n_voxels=V, seed=8800)

    def imu_set(cov_gyr, cov_acc, seed):
        iw = synth.make_imu(sc, seed=seed, cov_gyr=cov_gyr, cov_acc=cov_acc)
        facs = []
        for gyr, acc, dts in iw.samples:
            f = vx.IMU_PRE(iw.states_init[0, 15:18], iw.states_init[0, 18:21])
            for g, a, dt in zip(gyr, acc, dts):
                f.add_imu(g, a, dt, iw.noise_meas, iw.noise_walk)
            facs.append(f)
        return iw, facs

    def factor():
        f = vx.LidarFactor(W); f.push_voxels(sc.clusters, sc.fix, sc.coe); f.evaluate_only_residual(sc.poses_init); f.snapshot_cache()
        return f

    def run(f, iw, facs):
        blobs = [x.blob.copy() for x in facs]
        f.restore_cache()
        out = vx.LI_BA_Optimizer(imu_coef=1e-4).damping_iter(iw.states_init, f, facs, max_iter=3)
        for x, b in zip(facs, blobs): x.blob[:] = b          # the optimiser moves the factors' bias deltas: undo for the next use
        return out

    A, B = imu_set(0.01, 1.0, 8801), imu_set(0.03, 2.5, 8801)
    assert not np.array_equal(A[1][0].blob, B[1][0].blob)
    mixed = (A[0], [B[1][1], A[1][0], A[1][2], B[1][3], A[1][4]])       # factors of both windows, some in slots they did not have before
    shared = factor()
    for iw, facs in (A, B, mixed, A):
        got = run(shared, iw, facs)
        ref = run(factor(), iw, facs)
        assert np.array_equal(got["states"], ref["states"]) and np.array_equal(got["trace"], ref["trace"])


def test_undelivered_device_step_falls_back_to_the_host_solve(vx):
    """The default shell lets the device solve the reduced pose system inside the residual-sweep launch and recognises its step by NaN
    sentinels -- so a non-finite step (an ill-conditioned reduced system) looks like one that never arrived.  When the launch has ended
    and slots still hold sentinels, the shell must discard that launch's residual sweep, queue the sweeps again and take the host's
    (pivoted) solve, like the host-solve modes do -- not fail the call.  The test hook VXBA_OPT_DEBUG_SOLVE_TIMEOUT = 2 declares the first
    device step of a call undelivered: the result must equal the host-solve mode's, and the fallback must be counted."""
    sc, iw, blobs, facs, fo, fg = build(vx, 8, 3000, 24000, seed=821)
    blobs0 = [f.blob.copy() for f in facs]
    fg.snapshot_cache()
    fg.set_option("li_device_pose_solve", 0)
    host = vx.LI_BA_Optimizer(imu_coef=1e-4).damping_iter(iw.states_init, fg, facs, max_iter=4)
    for f, b in zip(facs, blobs0):
        f.blob[:] = b
    fg.restore_cache()
    fg.set_option("li_device_pose_solve", 1)
    n0 = fg.get_option("stat_li_device_fallbacks")
    fg.set_option("debug_solve_timeout", 2)
    got = vx.LI_BA_Optimizer(imu_coef=1e-4).damping_iter(iw.states_init, fg, facs, max_iter=4)
    fg.set_option("debug_solve_timeout", 0)
    assert fg.get_option("stat_li_device_fallbacks") == n0 + 1
    assert got["trace"].shape == host["trace"].shape and np.array_equal(got["trace"][:, 6:], host["trace"][:, 6:])
    assert np.allclose(got["trace"][:, :2], host["trace"][:, :2], rtol=1e-9)
    et, er = synth.pose_errors(got["states"][:, :12], host["states"][:, :12])
    assert et < 1e-9 and er < 1e-9, (et, er)
    ref = O.li_damping_iter(fo, iw.states_init, blobs, max_iter=4, thd_num=5, imu_coef=1e-4)
    et, er = synth.pose_errors(got["states"][:, :12], ref["states"][:, :12])
    assert et < 1e-7 and er < 1e-7, (et, er)


def test_gravity_variant_joint_system_and_residual_match_the_checkers(vx):
    """LI_BA_OptimizerGravity::divide_thread / only_residual / hess_plus under their own names (voxel_map.hpp:663-773): the (15W+3)^2 system
    with the gravity unknowns at the tail against the oracle restatement and, where libref.so travelled, the reference's own members."""
    from tests import _ref
    sc, iw, blobs, facs, fo, fg = build(vx, 6, 1500, 12000, seed=831)
    opt = vx.LI_BA_OptimizerGravity(imu_coef=1e-4)
    H, J, r = opt.divide_thread(iw.states_init, fg, facs)
    W = sc.win_size
    assert H.shape == (15 * W + 3, 15 * W + 3) and J.shape == (15 * W + 3,)
    checkers = [("oracle", O, fo)]
    R = _ref.backend()
    if R is not None and hasattr(R.lib(), "vxo_li_divide_thread_gravity"):
        fr = R.Oracle(W); fr.push_voxels(sc.clusters, sc.fix, sc.coe); fr.evaluate_only_residual(sc.poses_init)
        checkers.append(("reference", R, fr))
    for name, B, fb in checkers:
        Hr, Jr, rr = B.li_divide_thread_gravity(fb, iw.states_init, blobs, thd_num=5, imu_coef=1e-4)
        scale = np.abs(Hr).max()
        assert np.allclose(H, Hr, rtol=0, atol=1e-10 * scale), (name, np.abs(H - Hr).max() / scale)
        assert np.allclose(J, Jr, rtol=0, atol=1e-10 * np.abs(Jr).max()), name
        assert abs(r - rr) <= 1e-10 * abs(rr), name
        assert np.abs(Hr[-3:, :]).max() > 0                      # the gravity rows are really there
        r2 = opt.only_residual(iw.states_init, fg, facs)
        assert abs(r2 - B.li_only_residual_gravity(fb, iw.states_init, blobs, thd_num=5, imu_coef=1e-4)) <= 1e-10 * abs(rr), name
    # hess_plus into the (15W+3) system: the 6x6 blocks land at (15 i, 15 j), the tail stays untouched
    rng = np.random.default_rng(3)
    n, m = 15 * W + 3, 6 * W
    H15 = np.zeros((n, n)); J15 = np.zeros(n); hs = rng.normal(size=(m, m)); js = rng.normal(size=m)
    vx.load_library().vxba_hess_plus_gravity(W, H15, J15, np.ascontiguousarray(hs.T), js)       # column-major on the wire
    H15 = H15.T
    for i in range(W):
        assert np.array_equal(J15[15 * i:15 * i + 6], js[6 * i:6 * i + 6])
        for j in range(W):
            assert np.array_equal(H15[15 * i:15 * i + 6, 15 * j:15 * j + 6], hs[6 * i:6 * i + 6, 6 * j:6 * j + 6])
    assert np.count_nonzero(H15) == m * m and not H15[-3:, :].any() and not J15[-3:].any()
