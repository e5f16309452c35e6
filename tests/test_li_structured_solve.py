"""The structured solve of the damped LiDAR-inertial system (voxel-slam_amd/csrc/vxba_host.hpp: band Cholesky of the velocity / bias
unknowns + Schur complement onto the poses) against a dense solve, on random positive definite systems with exactly the sparsity the
LI_BA_Optimizer system has: dense pose-pose blocks (LiDAR), full 15 x 15 couplings between consecutive frames only (IMU factors),
and -- gravity variant -- loose velocity / bias unknowns of frame 0 in front and three gravity unknowns, coupled to everything, behind.
Host code: runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from voxel_slam_amd import vxba

f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def li_like_system(rng, nframes, lead_y, tail_x, cond_boost=1.0):
    m = lead_y + 15 * nframes + tail_x
    J = []
    # LiDAR: dense in the pose unknowns of all frames
    pose_idx = np.concatenate([lead_y + 15 * j + np.arange(6) for j in range(nframes)])
    for _ in range(80):
        r = np.zeros(m); r[pose_idx] = rng.normal(size=pose_idx.size); J.append(r)
    # IMU factor between consecutive frames: couples all 15 + 15 unknowns (and gravity); the lead block is frame -1's velocity / biases
    blocks = ([np.arange(lead_y)] if lead_y else []) + [lead_y + 15 * j + np.arange(15) for j in range(nframes)]
    # the factor towards the gauge-fixed frame in front leaves its mark on the first block alone
    for _ in range(40):
        r = np.zeros(m); idx = np.concatenate([blocks[0], m - tail_x + np.arange(tail_x)]); r[idx] = rng.normal(size=idx.size); J.append(r)
    for a, b in zip(blocks[:-1], blocks[1:]):
        idx = np.concatenate([a, b, m - tail_x + np.arange(tail_x)])
        for _ in range(40):
            r = np.zeros(m); r[idx] = rng.normal(size=idx.size) * rng.choice([1.0, cond_boost]); J.append(r)
    J = np.array(J)
    A = J.T @ J
    A += 1e-3 * np.diag(np.diag(A))          # the LM damping
    return np.ascontiguousarray(A), rng.normal(size=m)


@pytest.mark.parametrize("nframes,lead_y,tail_x", [(9, 0, 0), (9, 9, 3), (1, 0, 0), (2, 9, 3), (4, 0, 0)])
def test_band_schur_matches_dense_solve(nframes, lead_y, tail_x):
    L = vxba.load_library()
    L.vxba_debug_band_schur.argtypes = [C.c_int, f64p, f64p, C.c_int, C.c_int, C.c_int, f64p]
    rng = np.random.default_rng(7 * nframes + lead_y)
    for boost in (1.0, 1e3):
        A, b = li_like_system(rng, nframes, lead_y, tail_x, boost)
        x = np.zeros_like(b)
        assert L.vxba_debug_band_schur(A.shape[0], A, b, nframes, lead_y, tail_x, x) == 0
        ref = np.linalg.solve(A, b)
        assert np.allclose(A @ x, b, rtol=0, atol=1e-9 * np.abs(b).max() * max(1.0, boost))
        assert np.allclose(x, ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())


def test_band_schur_refuses_an_indefinite_band():
    L = vxba.load_library()
    L.vxba_debug_band_schur.argtypes = [C.c_int, f64p, f64p, C.c_int, C.c_int, C.c_int, f64p]
    rng = np.random.default_rng(3)
    A, b = li_like_system(rng, 3, 0, 0)
    A[8, 8] = -1.0                            # a velocity unknown with a negative pivot: the caller must take the dense pivoted path
    x = np.zeros_like(b)
    assert L.vxba_debug_band_schur(A.shape[0], A, b, 3, 0, 0, x) != 0
