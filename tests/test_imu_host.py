"""Host-side inertial half of the product (voxel-slam_amd/csrc/vxba_imu.hpp through the C ABI: vxba_imu_*,
vxba_hess_plus) against the CPU oracle on the same seeded IMU streams.  These entry points are host code by design
(O(W) work on 15x15 / 30x30 blocks that the reference also runs on the calling thread), so they run without a GPU;
everything that sweeps voxels is in the -m gpu tests."""
import numpy as np
import pytest

from voxel_slam_amd import synth
from tests import _oracle as O


@pytest.fixture(scope="module")
def vx():
    import __graft_entry__ as g
    from voxel_slam_amd import vxba
    try:
        vxba.load_library()
    except vxba.VxbaError:
        g.build()
        vxba.load_library()
    return vxba


@pytest.fixture(scope="module")
def window():
    sc = synth.make_scene(win_size=6, pts_per_scan=500, n_voxels=60, seed=synth.MASTER_SEED + 32)
    iw = synth.make_imu(sc, seed=99)
    return sc, iw


def preintegrate(vx, iw):
    out = []
    for gyr, acc, dts in iw.samples:
        f = vx.IMU_PRE(iw.states_init[0, 15:18], iw.states_init[0, 18:21])
        for g, a, dt in zip(gyr, acc, dts):
            f.add_imu(g, a, dt, iw.noise_meas, iw.noise_walk)
        out.append(f)
    return out


def test_preintegration_matches_oracle(vx, window):
    sc, iw = window
    ref = O.imu_preintegrate(iw.samples, iw.noise_meas, iw.noise_walk, iw.states_init[0, 15:18], iw.states_init[0, 18:21])
    got = np.stack([f.blob for f in preintegrate(vx, iw)])
    assert got.shape == ref.shape == (sc.win_size - 1, 304)
    # same recurrences, different operation order inside the 3x3 / 9x9 products: agreement to round-off
    scale = np.maximum(np.abs(ref), 1e-300)
    assert np.all(np.abs(got - ref) <= 1e-12 * scale + 1e-18), np.abs((got - ref) / scale).max()
    f = preintegrate(vx, iw)[0]
    assert np.allclose(f.field("R_delta") @ f.field("R_delta").T, np.eye(3), atol=1e-13)
    assert f.field("cov").shape == (15, 15) and abs(f.field("dtime") - iw.dt_frame) < 1e-12


def test_factor_evaluation_matches_oracle(vx, window):
    sc, iw = window
    facs = preintegrate(vx, iw)
    rng = np.random.default_rng(3)
    for i, f in enumerate(facs):
        f.blob[67:73] = rng.normal(0, 1e-3, 6)          # non-zero dbg / dba
        s1, s2 = iw.states_init[i], iw.states_init[i + 1]
        r_ref, jtj_ref, gg_ref = O.imu_evaluate(f.blob, s1, s2)
        r, jtj, gg = f.give_evaluate(s1, s2)
        # cov has a condition number of ~1e9: its inverse is only determined to ~1e-7 relative, whatever the algorithm
        assert np.isclose(r, r_ref, rtol=1e-6)
        assert np.allclose(jtj, jtj_ref, rtol=1e-6, atol=1e-6 * np.abs(jtj_ref).max())
        assert np.allclose(gg, gg_ref, rtol=1e-6, atol=1e-6 * np.abs(gg_ref).max())
        r0, j0, g0 = f.give_evaluate(s1, s2, jac_enable=False)
        assert r0 == r and j0 is None and g0 is None
        rg, jg, gg_g = f.give_evaluate_g(s1, s2)                     # gravity columns (preintegration.hpp:214-294)
        rg_ref, jg_ref, gg_ref2 = O.imu_evaluate_g(f.blob, s1, s2)
        assert rg == r and jg.shape == (33, 33)
        assert np.allclose(jg, jg_ref, rtol=1e-6, atol=1e-6 * np.abs(jg_ref).max())
        assert np.allclose(gg_g, gg_ref2, rtol=1e-6, atol=1e-6 * np.abs(gg_ref2).max())


def test_update_state_and_error_paths(vx, window):
    sc, iw = window
    f = preintegrate(vx, iw)[0]
    d = np.arange(15, dtype=np.float64) * 1e-3
    f.update_state(d)
    assert np.array_equal(f.field("dbg"), d[9:12]) and np.array_equal(f.field("dba"), d[12:15])
    assert np.array_equal(f.field("dbg_buf"), np.zeros(3))
    f.update_state(d)
    assert np.array_equal(f.field("dbg_buf"), d[9:12]) and np.allclose(f.field("dbg"), 2 * d[9:12])
    empty = vx.IMU_PRE()                                  # no samples: singular covariance -> loud error, not NaNs
    with pytest.raises(vx.VxbaError):
        empty.give_evaluate(iw.states_init[0], iw.states_init[1])


def test_hess_plus_scatter(vx):
    W = 4
    rng = np.random.default_rng(8)
    H15 = rng.normal(size=(15 * W, 15 * W)); J15 = rng.normal(size=15 * W)
    H6 = rng.normal(size=(6 * W, 6 * W)); J6 = rng.normal(size=6 * W)
    H, J = vx.hess_plus(W, H15, J15, H6, J6)
    He, Je = H15.copy(), J15.copy()
    for i in range(W):
        Je[15 * i:15 * i + 6] += J6[6 * i:6 * i + 6]
        for j in range(W):
            He[15 * i:15 * i + 6, 15 * j:15 * j + 6] += H6[6 * i:6 * i + 6, 6 * j:6 * j + 6]
    assert np.array_equal(H, He) and np.array_equal(J, Je)
