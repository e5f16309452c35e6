#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- small fixtures for the BA hot path, produced by THE REFERENCE'S OWN CODE.

PROVENANCE: every number below comes out of oracle/_ref/libref.so, i.e. /root/reference/VoxelSLAM/src/{tools,preintegration,
voxel_map}.hpp, unmodified, compiled where they lie by `make -C oracle ref` (oracle/ref_capi.cpp).  The image has no Eigen: the
headers are compiled against the Eigen API shim under oracle/shim/ (see oracle/shim/Eigen/Core for what that implies at the last-ulp
level; `backend` inside each file records which it was).  The reference itself ships no vectors, and /root/reference does not exist
on the GPU box -- which is why these are committed.  Inputs are synthetic (voxel-slam_amd/synth.py, seeds below).

Run from the repo root, in the container that has /root/reference:   python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _ref                  # noqa: E402
from voxel_slam_amd import synth        # noqa: E402

CASES = {
    # name: scene kwargs
    "w5_dense": dict(win_size=5, pts_per_scan=1500, n_voxels=96, p_obs=1.0, fix_frac=0.0, seed=9001, rot_sigma_deg=0.2, trans_sigma=0.03),
    "w10_sparse_fix": dict(win_size=10, pts_per_scan=2500, n_voxels=130, p_obs=0.7, fix_frac=0.3, seed=9002, rot_sigma_deg=0.1, trans_sigma=0.02),
    "w3_ragged": dict(win_size=3, pts_per_scan=400, n_voxels=41, p_obs=0.8, fix_frac=0.5, seed=9003, rot_sigma_deg=0.3, trans_sigma=0.05),
    # a window whose LM rejects steps (large initial error): reject branch, cache left at the rejected trial state
    "w6_reject": dict(win_size=6, pts_per_scan=3000, n_voxels=200, seed=4242, rot_sigma_deg=2.5, trans_sigma=0.25),
}
LM_ITERS = {"w6_reject": 8}


def build(R, name, kw):
    sc = synth.make_scene(**kw)
    coe = np.linspace(0.5, 1.5, sc.n_voxels)
    f = R.Oracle(sc.win_size)
    f.push_voxels(sc.clusters, sc.fix, coe)
    res0 = f.evaluate_only_residual(sc.poses_init)
    ev, U, merged = f.read_cache()
    H, J, r = f.acc_evaluate2(sc.poses_init)
    it = LM_ITERS.get(name, 4)
    lm = f.damping_iter(sc.poses_init, max_iter=it, thd_num=2)
    ev2, U2, merged2 = f.read_cache()
    d = dict(win_size=sc.win_size, points_body=sc.points_body, cell_ptr=sc.cell_ptr, clusters=sc.clusters, fix=sc.fix, coe=coe,
             poses_init=sc.poses_init, residual0=res0, eig_val=ev, eig_vec=U, merged=merged, Hess=H, JacT=J, residual_k3=r,
             lm_iters=it, lm_poses=lm["poses"], lm_trace=lm["trace"], lm_resis=lm["resis"], lm_hess=lm["hess"],
             lm_eig_val=ev2, lm_merged=merged2, backend=R.BACKEND_NAME)
    # the LiDAR-inertial optimizers on the same window (LI_BA_Optimizer 3 iterations, gravity variant 2)
    if sc.win_size >= 5 and name != "w6_reject":
        iw = synth.make_imu(sc, seed=kw["seed"] + 1)
        bg, ba = iw.states_init[0, 15:18], iw.states_init[0, 18:21]
        blobs = R.imu_preintegrate(iw.samples, iw.noise_meas, iw.noise_walk, bg, ba)
        gyr = np.stack([s[0] for s in iw.samples]); acc = np.stack([s[1] for s in iw.samples]); dts = np.stack([s[2] for s in iw.samples])
        f2 = R.Oracle(sc.win_size); f2.push_voxels(sc.clusters, sc.fix, sc.coe); f2.evaluate_only_residual(sc.poses_init)
        Hl, Jl, rl = R.li_divide_thread(f2, iw.states_init, blobs, 5, 1e-4)
        li = R.li_damping_iter(f2, iw.states_init, blobs, max_iter=3, imu_coef=1e-4)
        f2.evaluate_only_residual(sc.poses_init)
        lg = R.li_damping_iter_gravity(f2, iw.states_init, blobs, max_iter=2, imu_coef=1e-4)
        d.update(li_coe=sc.coe, imu_gyr=gyr, imu_acc=acc, imu_dt=dts, imu_noise_meas=iw.noise_meas, imu_noise_walk=iw.noise_walk,
                 li_states_init=iw.states_init, li_blobs=blobs, li_Hess=Hl, li_JacT=Jl, li_residual=rl,
                 li_states=li["states"], li_imus=li["imus"], li_hess=li["hess"],
                 lig_states=lg["states"], lig_imus=lg["imus"], lig_hess=lg["hess"], lig_resis=lg["resis"])
    return d


def build_localmap(R):
    """The reference's OctoTree driven through a sliding window (cut_voxel_multi -> recut + tras_opt -> damping_iter -> margi -> shift):
    the leaf table after every window, for the device-resident map and the restatement to be checked against."""
    from tests.test_oracle_octree import PRM, point_vars, to_world
    S, win, pts, seed, extent = 7, 3, 3000, 31, 6.0
    xyz, fp, poses_gt, _ = synth.make_scans(win_size=S, pts_per_scan=pts, extent=extent, seed=synth.MASTER_SEED + 900 + seed)
    rng = np.random.default_rng(seed)
    var = point_vars(xyz.shape[0], seed)
    kw = dict(PRM); kw["max_points"] = 60
    m = R.LocalMapOracle(win_size=win, **kw)
    f = R.Oracle(win)
    xb = []
    win_count = 0
    out = dict(S=S, win=win, xyz=xyz, fp=fp, var_seed=seed, max_points=60, backend=R.BACKEND_NAME,     # var = point_vars(n, var_seed)
               prm=np.array([kw["voxel_size"], kw["max_layer"], *kw["min_point"], kw["min_eigen_value"], *kw["plane_eigen_value_thre"]]))
    poses_in = []
    w = 0
    for k in range(S):
        pose = poses_gt[k].copy(); pose[9:12] += rng.normal(0, 0.01, 3)
        poses_in.append(pose)
        s = slice(fp[k], fp[k + 1])
        win_count += 1
        xb.append(pose.copy())
        f.clear()
        m.cut_voxel(win_count - 1, xyz[s], var[s], to_world(xb[-1], xyz[s]))
        m.recut(win_count, np.stack(xb), f)
        if win_count < win:
            continue
        lm = f.damping_iter(np.stack(xb), max_iter=3, thd_num=2)
        m.margi(win_count, lm["poses"], f)
        m.slide(1)
        xb = [p for p in lm["poses"][1:]]
        win_count -= 1
        lv = m.leaves()
        o = np.argsort(lv["node_id"], kind="stable")
        for key in ("node_id", "pcr_add", "pcr_fix", "eig_val", "center", "normal", "radius"):
            out[f"w{w}_{key}"] = lv[key][o]
        for key in ("layer", "isexist", "is_plane", "has_sw", "in_slide"):
            out[f"w{w}_{key}"] = lv[key][o].astype(np.uint8)
        for key in ("last_num", "n_point_fix"):
            out[f"w{w}_{key}"] = lv[key][o].astype(np.int32)
        out[f"w{w}_n_points"] = lv["n_points"][o].astype(np.int32)
        if k == S - 1:
            out["last_pcrs_local"] = lv["pcrs_local"][o]
        out[f"w{w}_poses"] = lm["poses"]; out[f"w{w}_trace"] = lm["trace"]
        w += 1
    out["poses_in"] = np.stack(poses_in); out["windows"] = w
    return out


if __name__ == "__main__":
    R = _ref.backend()
    if R is None:
        sys.exit("oracle/_ref/libref.so is not available: run in the container that has /root/reference (make -C oracle ref)")
    here = os.path.dirname(os.path.abspath(__file__))
    for name, kw in CASES.items():
        d = build(R, name, kw)
        np.savez_compressed(os.path.join(here, name + ".npz"), **d)
        print(name, {k: getattr(v, "shape", v) for k, v in d.items() if k in ("clusters", "Hess", "lm_trace", "li_hess")})
    d = build_localmap(R)
    np.savez_compressed(os.path.join(here, "localmap_cycle.npz"), **d)
    print("localmap_cycle", d["windows"], "windows;", d["w0_node_id"].shape[0], "->", d[f"w{d['windows'] - 1}_node_id"].shape[0], "leaves")
