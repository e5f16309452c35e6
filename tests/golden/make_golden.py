#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- small regression fixtures for the BA hot path.

PROVENANCE: the reference (hku-mars/Voxel-SLAM) holds no golden vectors and cannot be built or imported here
(Eigen/PCL/ROS absent), so these vectors are produced by the CPU oracle (oracle/, a restatement pinned through
mathematics in tests/test_oracle_math.py), NOT by the reference binary.  They freeze today's behaviour so that the
oracle, the device arithmetic and the HIP kernels can all be checked against one committed set of numbers on a box
that has neither the oracle sources rebuilt nor /root/reference.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _oracle as O          # noqa: E402
from voxel_slam_amd import synth        # noqa: E402

CASES = {
    # name: scene kwargs
    "w5_dense": dict(win_size=5, pts_per_scan=1500, n_voxels=96, p_obs=1.0, fix_frac=0.0, seed=9001, rot_sigma_deg=0.2, trans_sigma=0.03),
    "w10_sparse_fix": dict(win_size=10, pts_per_scan=2500, n_voxels=130, p_obs=0.7, fix_frac=0.3, seed=9002, rot_sigma_deg=0.1, trans_sigma=0.02),
    "w3_ragged": dict(win_size=3, pts_per_scan=400, n_voxels=41, p_obs=0.8, fix_frac=0.5, seed=9003, rot_sigma_deg=0.3, trans_sigma=0.05),
}


def build(name, kw):
    sc = synth.make_scene(**kw)
    coe = np.linspace(0.5, 1.5, sc.n_voxels)
    f = O.Oracle(sc.win_size)
    f.push_voxels(sc.clusters, sc.fix, coe)
    res0 = f.evaluate_only_residual(sc.poses_init)
    ev, U, merged = f.read_cache()
    H, J, r = f.acc_evaluate2(sc.poses_init)
    lm = f.damping_iter(sc.poses_init, max_iter=4, thd_num=2)
    return dict(win_size=sc.win_size, points_body=sc.points_body, cell_ptr=sc.cell_ptr, clusters=sc.clusters, fix=sc.fix, coe=coe,
                poses_init=sc.poses_init, residual0=res0, eig_val=ev, eig_vec=U, merged=merged, Hess=H, JacT=J, residual_k3=r,
                lm_poses=lm["poses"], lm_trace=lm["trace"], lm_resis=lm["resis"], lm_hess=lm["hess"])


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name, kw in CASES.items():
        d = build(name, kw)
        np.savez_compressed(os.path.join(here, name + ".npz"), **d)
        print(name, {k: getattr(v, "shape", v) for k, v in d.items() if k in ("clusters", "Hess", "lm_trace")})
