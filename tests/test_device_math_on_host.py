"""The per-lane arithmetic of the HIP kernels (voxel-slam_amd/csrc/vxba_math.hpp), compiled for the host and
checked against the oracle term by term: Jacobi eigensolver, K2 merge + cache, K3 rank-3 rows / gradient /
block-diagonal correction.  (The kernels themselves are exercised by the -m gpu tests.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import _oracle as O
from voxel_slam_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module", params=["exact", "hw_estimates"])
def hm(request):
    """Two builds of the lane arithmetic for the host: plain divisions / square roots, and (-DVXM_EMULATE_HW_ESTIMATES) the GPU's
    reciprocal / reciprocal-square-root ESTIMATES emulated at their 22-bit accuracy, so that every Newton refinement on the device
    path (eigen-solver rotations, the warm start's cheap tangents) is run and checked on the CPU."""
    emu = request.param == "hw_estimates"
    src = os.path.join(HERE, "hostmath", "vxm_hostcheck.cpp")
    so = os.path.join(HERE, "hostmath", "libvxm_hostcheck_emu.so" if emu else "libvxm_hostcheck.so")
    hdr = os.path.join(HERE, "..", "voxel-slam_amd", "csrc", "vxba_math.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"] + (["-DVXM_EMULATE_HW_ESTIMATES"] if emu else []) + ["-o", so, src])
    L = C.CDLL(so)
    L.vxmh_eig_sym3.argtypes = [f64p, f64p, f64p]
    L.vxmh_eig_sym3_warm.argtypes = [f64p, f64p, f64p, f64p]
    L.vxmh_k2.argtypes = [C.c_int, C.c_int, f64p, f64p, f64p, f64p, f64p, f64p, f64p, C.POINTER(C.c_double)]
    L.vxmh_k3.argtypes = [C.c_int, C.c_int, f64p, f64p, f64p, f64p, f64p, f64p, f64p, f64p, C.POINTER(C.c_double)]
    L.vxmh_k3_spare.argtypes = L.vxmh_k3.argtypes
    L.vxmh_k2_f32.argtypes = [C.c_int, C.c_int, f64p, f64p, f64p, f64p, np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS"), f64p, f64p,
                              C.POINTER(C.c_double)]
    return L


def test_jacobi_eigensolver(hm):
    rng = np.random.default_rng(5)
    mats = []
    for s in (1e-8, 1.0, 1e5):
        for _ in range(100):
            A = rng.normal(size=(3, 3)) * s
            mats.append(0.5 * (A + A.T))
    for _ in range(200):   # planar covariances incl. nearly equal in-plane eigenvalues
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        lam = np.array([4e-4 * rng.uniform(0.2, 3), 0.07, 0.07 * (1 + rng.choice([0, 1e-12, 1e-6, 0.3]))])
        mats.append(Q @ np.diag(lam) @ Q.T)
    mats += [np.zeros((3, 3)), np.eye(3), np.diag([3.0, 1.0, 2.0])]
    for M in mats:
        M = 0.5 * (M + M.T)
        c6 = np.array([M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2]])
        lam = np.zeros(3); U = np.zeros(9)
        hm.vxmh_eig_sym3(c6, lam, U)
        U = U.reshape(3, 3)
        ref = np.linalg.eigvalsh(M)
        nrm = max(np.abs(M).max(), 1e-300)
        assert np.all(np.diff(lam) >= 0)
        assert np.allclose(lam, ref, rtol=0, atol=1e-14 * nrm)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-14)
        assert np.allclose(M @ U, U * lam, atol=2e-14 * nrm)


def test_warm_started_eigensolver(hm):
    """Warm start from a nearby basis (what K2 does between LM iterations), from an exact basis, from garbage."""
    rng = np.random.default_rng(6)
    for trial in range(300):
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        lam_true = np.sort(np.array([4e-4 * rng.uniform(0.2, 3), 0.07 * rng.uniform(0.5, 1.5), 0.07 * rng.uniform(0.5, 1.5)]))
        M = Q @ np.diag(lam_true) @ Q.T; M = 0.5 * (M + M.T)
        c6 = np.array([M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2]])
        if trial % 3 == 0:      # nearby basis: previous eigenvectors rotated by a small angle
            from scipy.spatial.transform import Rotation
            Up = Q @ Rotation.from_rotvec(rng.normal(size=3) * 1e-3).as_matrix()
        elif trial % 3 == 1:    # exact previous basis, columns permuted / sign-flipped
            Up = Q[:, rng.permutation(3)] * rng.choice([-1.0, 1.0], size=3)
        else:                   # never-written cache (zeros) or non-orthogonal junk -> cold start
            Up = np.zeros((3, 3)) if trial % 2 else rng.normal(size=(3, 3))
        lam = np.zeros(3); U = np.zeros(9)
        hm.vxmh_eig_sym3_warm(c6, np.ascontiguousarray(Up).reshape(9), lam, U)
        U = U.reshape(3, 3)
        nrm = np.abs(M).max()
        assert np.allclose(lam, np.linalg.eigvalsh(M), rtol=0, atol=2e-15 * nrm)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-13)
        assert np.allclose(M @ U, U * lam, atol=4e-15 * nrm + 1e-13 * nrm * (trial % 3 == 0))


def test_warm_start_fixed_sweeps_and_fallback(hm):
    """The warm start runs three fixed branch-free sweeps and only then looks at convergence: small basis errors (the LM case) must be
    done by then, large ones and (nearly) degenerate in-plane eigenvalues must be finished by the generic loop -- same accuracy either way."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(60)
    for angle in (1e-7, 1e-4, 3e-3, 5e-2, 0.6, 2.0):
        for gap in (0.3, 1e-6, 1e-12, 0.0):
            for _ in range(20):
                Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
                lam_true = np.sort(np.array([4e-4 * rng.uniform(0.2, 3), 0.07, 0.07 * (1 + gap)]))
                M = Q @ np.diag(lam_true) @ Q.T; M = 0.5 * (M + M.T)
                c6 = np.array([M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2]])
                Up = Q @ Rotation.from_rotvec(rng.normal(size=3) * angle).as_matrix()
                lam = np.zeros(3); U = np.zeros(9)
                hm.vxmh_eig_sym3_warm(c6, np.ascontiguousarray(Up).reshape(9), lam, U)
                U = U.reshape(3, 3)
                nrm = np.abs(M).max()
                assert np.allclose(lam, np.linalg.eigvalsh(M), rtol=0, atol=2e-15 * nrm), (angle, gap)
                assert np.allclose(U.T @ U, np.eye(3), atol=1e-13), (angle, gap)
                assert np.allclose(M @ U, U * lam, atol=1e-13 * nrm), (angle, gap)
                # the plane normal (eigenvector of the isolated smallest eigenvalue) is determined: compare it up to sign
                assert abs(abs(U[:, 0] @ Q[:, 0]) - 1.0) < 1e-12, (angle, gap)


@pytest.mark.parametrize("p_obs,fix_frac", [(1.0, 0.0), (0.6, 0.4)])
def test_k2_and_k3_lane_math_match_oracle(hm, p_obs, fix_frac):
    sc = synth.make_scene(win_size=5, pts_per_scan=3000, n_voxels=150, p_obs=p_obs, fix_frac=fix_frac, seed=21,
                          rot_sigma_deg=0.2, trans_sigma=0.03)
    V, W = sc.n_voxels, sc.win_size
    coe = np.linspace(0.5, 2.0, V)
    f = O.Oracle(W)
    f.push_voxels(sc.clusters, sc.fix, coe)
    r_ref = f.evaluate_only_residual(sc.poses_init)
    ev_ref, U_ref, m_ref = f.read_cache()

    ev = np.zeros((V, 3)); U = np.zeros((V, 9)); m = np.zeros((V, 10)); r = C.c_double(0)
    hm.vxmh_k2(V, W, sc.clusters, sc.fix, coe, sc.poses_init, ev, U, m, C.byref(r))
    assert np.allclose(m, m_ref, rtol=1e-13, atol=1e-9)
    assert np.array_equal(m[:, 9], m_ref[:, 9])
    assert np.allclose(ev, ev_ref, rtol=1e-9, atol=1e-12)
    assert np.isclose(r.value, r_ref, rtol=1e-10)
    # eigenvectors agree up to sign
    d = np.abs(np.einsum("nck,nck->nc", U.reshape(V, 3, 3), U_ref.reshape(V, 3, 3)))
    assert np.all(d[:, 0] > 1 - 1e-9)

    # K3 lane math against the oracle's acc_evaluate2 (same cache on both sides)
    H_ref, J_ref, res_ref = f.acc_evaluate2(sc.poses_init)
    n = 6 * W
    H = np.zeros((n, n)); J = np.zeros(n); rr = C.c_double(0)
    hm.vxmh_k3(V, W, sc.clusters, coe, ev_ref, U_ref, m_ref, sc.poses_init, H, J, C.byref(rr))
    H = H.T
    assert np.allclose(H, H_ref, rtol=1e-9, atol=1e-10 * np.abs(H_ref).max())
    assert np.allclose(J, J_ref, rtol=1e-9, atol=1e-11 * np.abs(J_ref).max())
    assert np.isclose(rr.value, res_ref, rtol=1e-13)
    # the narrow-window kernel's variant: Drt / Dtt as products of the z row with the spare columns (sqrt2 sqrt(coe) u)
    H2 = np.zeros((n, n)); J2 = np.zeros(n)
    hm.vxmh_k3_spare(V, W, sc.clusters, coe, ev_ref, U_ref, m_ref, sc.poses_init, H2, J2, C.byref(rr))
    H2 = H2.T
    assert np.allclose(H2, H_ref, rtol=1e-9, atol=1e-10 * np.abs(H_ref).max())
    assert np.allclose(H2, H2.T, rtol=0, atol=1e-12 * np.abs(H_ref).max())
    assert np.allclose(J2, J_ref, rtol=1e-9, atol=1e-11 * np.abs(J_ref).max())


@pytest.mark.parametrize("p_obs,fix_frac", [(1.0, 0.0), (0.6, 0.3)])
def test_f32_recentred_cluster_records(hm, p_obs, fix_frac):
    """VXBA_PRECISION_MIXED_F32_CLUSTERS: the residual sweep's arithmetic on [C | c | n] records rounded to f32 stays within ~1e-5 of the
    fp64 one on a window whose clusters sit tens of metres from the sensor -- and the raw moments rounded to f32 do not (why the records
    are re-centred)."""
    sc = synth.make_scene(win_size=6, pts_per_scan=6000, n_voxels=300, p_obs=p_obs, fix_frac=fix_frac, seed=77, rot_sigma_deg=0.2, trans_sigma=0.03)
    V, W = sc.n_voxels, sc.win_size
    # move the whole scene 40 m away from the sensor: body-frame second moments ~ n * 1600 m^2
    off = np.array([40.0, -25.0, 3.0])
    cl = sc.clusters.reshape(V, W, 10).copy()
    n = cl[..., 9]
    v = cl[..., 6:9].copy()
    I, J = [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]
    for k in range(6):
        cl[..., k] += v[..., I[k]] * off[J[k]] + off[I[k]] * v[..., J[k]] + n * off[I[k]] * off[J[k]]
    cl[..., 6:9] += n[..., None] * off
    poses = sc.poses_init.reshape(W, 12).copy()
    for i in range(W):
        R = poses[i, :9].reshape(3, 3).T        # column-major in the pose record
        poses[i, 9:12] -= R @ off                # world positions unchanged
    cl = np.ascontiguousarray(cl.reshape(V * W, 10)); poses = np.ascontiguousarray(poses.reshape(-1))
    coe = np.linspace(0.5, 2.0, V)
    ev = np.zeros((V, 3)); U = np.zeros((V, 9)); m = np.zeros((V, 10)); r = C.c_double(0)
    hm.vxmh_k2(V, W, cl, sc.fix, coe, poses, ev, U, m, C.byref(r))
    ev0 = np.zeros((V, 3)); r0 = C.c_double(0)
    hm.vxmh_k2(V, W, np.ascontiguousarray(sc.clusters), sc.fix, coe, sc.poses_init, ev0, U, np.zeros((V, 10)), C.byref(r0))
    assert np.isclose(r.value, r0.value, rtol=1e-9)            # the shifted scene is the same scene
    rec = np.zeros((V * W, 10), dtype=np.float32)
    ev32 = np.zeros((V, 3)); m32 = np.zeros((V, 10)); r32 = C.c_double(0)
    hm.vxmh_k2_f32(V, W, cl, sc.fix, coe, poses, rec, ev32, m32, C.byref(r32))
    assert np.array_equal(m32[:, 9], m[:, 9])                    # point counts exact
    assert np.all(rec[cl[:, 9] == 0] == 0)                       # unobserved slots: all-zero records
    assert abs(r32.value / r.value - 1) < 2e-5, r32.value / r.value - 1
    assert np.allclose(ev32[:, 0], ev[:, 0], rtol=2e-4, atol=0) and np.allclose(ev32[:, 1:], ev[:, 1:], rtol=1e-5)
    # merged mean: micrometres
    assert np.max(np.abs(m32[:, 6:9] / m32[:, 9:10] - m[:, 6:9] / m[:, 9:10])) < 2e-5
    # the same through naively rounded raw moments: the smallest eigenvalue (the residual) is destroyed
    ev_n = np.zeros((V, 3)); r_n = C.c_double(0)
    hm.vxmh_k2(V, W, cl.astype(np.float32).astype(np.float64), sc.fix, coe, poses, ev_n, U, np.zeros((V, 10)), C.byref(r_n))
    worst_naive, worst_centred = np.max(np.abs(ev_n[:, 0] / ev[:, 0] - 1)), np.max(np.abs(ev32[:, 0] / ev[:, 0] - 1))
    assert worst_naive > 0.05 and worst_naive > 300 * worst_centred, (worst_naive, worst_centred)
