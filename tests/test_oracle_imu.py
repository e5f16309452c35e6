"""Pins the inertial half of the CPU oracle (oracle/vxo_imu.hpp) through mathematics -- the reference ships no
vectors for it: SO(3) Jacobians against finite differences, the LU inverse against numpy, the
preintegrated measurement against a direct integration and its bias Jacobians / the factor Jacobian against finite
differences, the joint LiDAR-inertial gradient against finite differences of the joint cost."""
import numpy as np
import pytest

from voxel_slam_amd import synth
from tests import _oracle as O


def rod(v):
    return synth.rodrigues(np.asarray(v, dtype=np.float64))


def perturb_state(st, d):
    """st (+) d in the reference's tangent order [dphi dp dv dbg dba] (tools.hpp:154-162)."""
    out = st.copy()
    R = st[:9].reshape(3, 3).T
    out[:9] = (R @ rod(d[:3])).T.reshape(9)
    out[9:21] = st[9:21] + d[3:15]
    return out


@pytest.fixture(scope="module")
def window():
    sc = synth.make_scene(win_size=5, pts_per_scan=3000, n_voxels=400, seed=synth.MASTER_SEED + 31)
    iw = synth.make_imu(sc)
    blobs = O.imu_preintegrate(iw.samples, iw.noise_meas, iw.noise_walk, iw.states_init[0, 15:18], iw.states_init[0, 18:21])
    return sc, iw, blobs


def test_so3_right_jacobians():
    rng = np.random.default_rng(5)
    assert np.array_equal(O.jr(np.array([1e-12, -2e-12, 5e-13])), np.eye(3))       # below the 1e-9 cut-off
    for scale in (1e-4, 0.3, 2.5):
        v = rng.normal(size=3); v *= scale / np.linalg.norm(v)
        J = O.jr(v)
        # Exp(v + e) = Exp(v) Exp(Jr(v) e)
        num = np.zeros((3, 3)); h = 1e-7
        for k in range(3):
            e = np.zeros(3); e[k] = h
            num[:, k] = synth.so3_log(rod(v).T @ rod(v + e)) / h
        assert np.allclose(J, num, atol=2e-6)
        Ji = O.jr_inv(rod(v))
        assert np.allclose(Ji @ J, np.eye(3), atol=1e-9)
    assert np.array_equal(O.jr_inv(np.eye(3)), np.eye(3))


def test_lu_inverse_matches_numpy():
    rng = np.random.default_rng(6)
    A = rng.normal(size=(15, 15)); A = A @ A.T + 1e-3 * np.eye(15)
    A *= np.outer(10.0 ** rng.uniform(-4, 0, 15), np.ones(15)); A = 0.5 * (A + A.T) + np.diag(10.0 ** rng.uniform(-6, -2, 15))
    inv = O.mat_inverse(A)
    assert np.allclose(inv @ A, np.eye(15), atol=1e-8)
    assert np.allclose(inv, np.linalg.inv(A), rtol=1e-7, atol=1e-9 * np.abs(np.linalg.inv(A)).max())


def test_preintegration_against_direct_integration(window):
    sc, iw, blobs = window
    gyr, acc, dts = iw.samples[1]
    R = np.eye(3); v = np.zeros(3); p = np.zeros(3)
    for g, a, dt in zip(gyr, acc, dts):
        p = p + v * dt + 0.5 * dt * dt * (R @ a)
        v = v + dt * (R @ a)
        R = R @ rod(g * dt)
    b = blobs[1]
    assert np.allclose(b[:9].reshape(3, 3).T, R, atol=1e-13)
    assert np.allclose(b[9:12], p, atol=1e-14) and np.allclose(b[12:15], v, atol=1e-13)
    assert abs(b[66] - dts.sum()) < 1e-15
    cov = b[79:].reshape(15, 15).T
    assert np.allclose(cov, cov.T, atol=1e-18) and np.all(np.linalg.eigvalsh(cov) > 0)
    # preintegration really reproduces the relative motion of the ground truth (up to sensor noise / bias-estimate error)
    Rg = iw.states_gt[:, :9].reshape(-1, 3, 3).transpose(0, 2, 1)
    assert np.linalg.norm(synth.so3_log(R.T @ (Rg[1].T @ Rg[2]))) < 2e-3


def test_bias_jacobians_against_finite_differences(window):
    sc, iw, blobs = window
    gyr, acc, dts = iw.samples[0]
    b0 = blobs[0]
    R0 = b0[:9].reshape(3, 3).T
    h = 1e-6
    for which, off_r, off_p, off_v in (("bg", 21, 30, 48), ("ba", None, 39, 57)):
        for k in range(3):
            e = np.zeros(3); e[k] = h
            b = O.imu_init()
            for g, a, dt in zip(gyr, acc, dts):
                O.imu_add(b, g - (e if which == "bg" else 0), a - (e if which == "ba" else 0), dt, iw.noise_meas, iw.noise_walk)
            dp = (b[9:12] - b0[9:12]) / h
            dv = (b[12:15] - b0[12:15]) / h
            assert np.allclose(dp, b0[off_p:off_p + 9].reshape(3, 3).T[:, k], atol=2e-6)
            assert np.allclose(dv, b0[off_v:off_v + 9].reshape(3, 3).T[:, k], atol=2e-6)
            if off_r is not None:
                dr = synth.so3_log(R0.T @ b[:9].reshape(3, 3).T) / h
                assert np.allclose(dr, b0[off_r:off_r + 9].reshape(3, 3).T[:, k], atol=2e-6)


def _half_cost(blob, s1, s2):
    return 0.5 * O.imu_evaluate(blob, s1, s2, jac_enable=False)[0]


def test_factor_gradient_against_finite_differences(window):
    sc, iw, blobs = window
    blob = blobs[2].copy()
    blob[67:70] = [2e-4, -1e-4, 3e-4]     # non-zero dbg / dba so that the bias-correction blocks are exercised
    blob[70:73] = [3e-3, 1e-3, -2e-3]
    s1, s2 = iw.states_init[2], iw.states_init[3]
    _, jtj, gg = O.imu_evaluate(blob, s1, s2)
    assert np.allclose(jtj, jtj.T, rtol=1e-9, atol=1e-6)
    num = np.zeros(30)
    for k in range(30):
        h = 1e-6
        vals = []
        for sgn in (+1, -1):
            d = np.zeros(30); d[k] = sgn * h
            bb = blob.copy()
            bb[67:70] += d[9:12]          # update_state moves the factor's dbg / dba together with frame 1's biases
            bb[70:73] += d[12:15]
            vals.append(_half_cost(bb, perturb_state(s1, d[:15]), perturb_state(s2, d[15:])))
        num[k] = (vals[0] - vals[1]) / (2 * h)
    assert np.allclose(gg, num, rtol=2e-4, atol=1e-6 * np.abs(gg).max())


def test_factor_information_at_zero_residual(window):
    """Where the residual vanishes the Hessian of the half cost is exactly J^T cov^-1 J."""
    sc, iw, blobs = window
    blob = blobs[0]
    s1 = iw.states_init[0].copy()
    R1 = s1[:9].reshape(3, 3).T
    dt = blob[66]; g = s1[21:24]
    Rc = blob[:9].reshape(3, 3).T; tc = blob[9:12]; vc = blob[12:15]
    s2 = s1.copy()
    s2[:9] = (R1 @ Rc).T.reshape(9)
    s2[12:15] = s1[12:15] + g * dt + R1 @ vc
    s2[9:12] = s1[9:12] + s1[12:15] * dt + 0.5 * g * dt * dt + R1 @ tc
    r, jtj, gg = O.imu_evaluate(blob, s1, s2)
    assert r < 1e-18 and np.abs(gg).max() < 1e-6
    # second differences along the 12 non-bias directions of frame 2 (frame-1 bias moves would also need dbg)
    idx = [15 + k for k in range(9)]
    h = 1e-4
    for a in idx:
        d = np.zeros(30); d[a] = h
        fpp = _half_cost(blob, s1, perturb_state(s2, d[15:]))
        fmm = _half_cost(blob, s1, perturb_state(s2, -d[15:]))
        num = (fpp + fmm) / (h * h)       # f(0) = 0
        assert np.isclose(num, jtj[a, a], rtol=1e-4), (a, num, jtj[a, a])


def test_joint_system_assembly_and_gradient(window):
    sc, iw, blobs = window
    W = sc.win_size
    o = O.Oracle(W)
    o.push_voxels(sc.clusters, sc.fix, sc.coe)
    o.evaluate_only_residual(sc.poses_init)          # seeds the cache at the linearisation point
    coef = 1e-4
    H, J, r = O.li_divide_thread(o, iw.states_init, blobs, thd_num=5, imu_coef=coef)
    # manual assembly: imu_coef * IMU blocks + the 6x6 LiDAR blocks at (15 i, 15 j)
    H6, J6, r6 = o.divide_thread(sc.poses_init, thd_num=5)
    Hm = np.zeros_like(H); Jm = np.zeros_like(J); rm = 0.0
    for i in range(W - 1):
        ri, jtj, gg = O.imu_evaluate(blobs[i], iw.states_init[i], iw.states_init[i + 1])
        Hm[15 * i:15 * i + 30, 15 * i:15 * i + 30] += jtj; Jm[15 * i:15 * i + 30] += gg; rm += ri
    Hm *= coef; Jm *= coef; rm *= 0.5 * coef
    for i in range(W):
        Jm[15 * i:15 * i + 6] += J6[6 * i:6 * i + 6]
        for j in range(W):
            Hm[15 * i:15 * i + 6, 15 * j:15 * j + 6] += H6[6 * i:6 * i + 6, 6 * j:6 * j + 6]
    assert np.allclose(H, Hm, rtol=1e-12, atol=1e-9) and np.allclose(J, Jm, rtol=1e-12, atol=1e-12)
    assert np.isclose(r, rm + r6, rtol=1e-13)
    assert np.isclose(O.li_only_residual(o, iw.states_init, blobs, 5, coef), r, rtol=1e-12)
    # gradient of the joint cost: finite differences over pose / velocity directions of two frames
    for k in (15 + 0, 15 + 4, 15 + 7, 45 + 2, 45 + 5):
        h = 1e-6
        vals = []
        for sgn in (+1, -1):
            st = iw.states_init.copy()
            d = np.zeros(15); d[k % 15] = sgn * h
            st[k // 15] = perturb_state(st[k // 15], d)
            vals.append(O.li_only_residual(o, st, blobs, 5, coef))
        num = (vals[0] - vals[1]) / (2 * h)
        assert np.isclose(num, J[k], rtol=5e-4, atol=1e-7 * np.abs(J).max()), (k, num, J[k])


def test_li_damping_iter_reduces_cost_and_error(window):
    sc, iw, blobs = window
    o = O.Oracle(sc.win_size)
    o.push_voxels(sc.clusters, sc.fix, sc.coe)
    o.evaluate_only_residual(sc.poses_init)
    out = O.li_damping_iter(o, iw.states_init, blobs, max_iter=6)
    tr = out["trace"]
    assert tr.shape[0] >= 1 and tr[0, 6] == 1
    acc = tr[tr[:, 6] == 1]
    assert np.all(acc[:, 1] < acc[:, 0])
    e0 = synth.pose_errors(iw.states_init[:, :12], iw.states_gt[:, :12])
    e1 = synth.pose_errors(out["states"][:, :12], iw.states_gt[:, :12])
    assert e1[0] < e0[0] and e1[1] < e0[1]
    assert np.array_equal(out["states"][0], iw.states_init[0])        # gauge: frame 0 untouched
    # accepted steps leave the factors' bias deltas moved; dbg_buf holds the previous value
    assert np.any(out["imus"][:, 67:73] != 0)


def test_gravity_columns_against_finite_differences(window):
    """give_evaluate_g: the three extra Jacobian columns are the derivative of the residual w.r.t. frame 1's gravity."""
    sc, iw, blobs = window
    s1, s2 = iw.states_init[1], iw.states_init[2]
    r, jtj, gg = O.imu_evaluate_g(blobs[1], s1, s2)
    r0, jtj0, gg0 = O.imu_evaluate(blobs[1], s1, s2)
    assert r == r0 and np.array_equal(jtj[:30, :30], jtj0) and np.array_equal(gg[:30], gg0)
    num = np.zeros(3); h = 1e-6
    for k in range(3):
        v = []
        for sgn in (+1, -1):
            a = s1.copy(); a[21 + k] += sgn * h
            v.append(_half_cost(blobs[1], a, s2))
        num[k] = (v[0] - v[1]) / (2 * h)
    assert np.allclose(gg[30:], num, rtol=1e-5)


def test_gravity_optimizer_reduces_cost(window):
    sc, iw, blobs = window
    o = O.Oracle(sc.win_size)
    o.push_voxels(sc.clusters, sc.fix, sc.coe)
    o.evaluate_only_residual(sc.poses_init)
    st = iw.states_init.copy()
    st[:, 21:24] += [0.05, -0.03, 0.08]                  # a wrong gravity estimate shared by all frames
    out = O.li_damping_iter_gravity(o, st, blobs, max_iter=5)
    assert out["hess"].shape == (15 * sc.win_size + 3,) * 2
    assert out["resis"][1] < out["resis"][0]
    g = out["states"][:, 21:24]
    assert np.all(g == g[0])                             # one gravity vector, copied to every frame (:823)
    assert np.array_equal(out["states"][0, :12], st[0, :12])          # pose gauge on frame 0 ...
    assert not np.array_equal(out["states"][0, 12:21], st[0, 12:21])  # ... but its velocity / biases move (:800-803)
