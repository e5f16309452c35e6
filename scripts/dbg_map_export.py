import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from tests import _oracle as O
from tests.test_oracle_octree import PRM, point_vars, to_world
from tests.test_gpu_local_mapping_cycle import lio_leaf_args
from voxel_slam_amd import synth, vxba
S, win, pts, seed = 6, 4, 20000, 7
xyz, fp, poses_gt, _ = synth.make_scans(win_size=S, pts_per_scan=pts, seed=synth.MASTER_SEED + 900 + seed)
mg = vxba.LocalMap(win_size=win, **PRM); fg = vxba.LidarFactor(win)
ge = vxba.LioEstimator(PRM["voxel_size"], PRM["max_layer"])
xg = []; win_count = 0
cov = np.eye(15) * 1e-4
for k in range(S):
    s = slice(fp[k], fp[k + 1])
    scan32 = xyz[s].astype(np.float32)
    state = np.concatenate([poses_gt[k], np.zeros(9), [0, 0, -9.8]])
    ge.var_init(scan32)
    lv = mg.leaves() if k else None
    if lv is not None and (lv["is_plane"] & (lv["last_num"] > 0)).sum() > 200:
        g2 = vxba.LioEstimator(PRM["voxel_size"], PRM["max_layer"]); g2.map_update(*lio_leaf_args(lv)); g2.var_init(scan32)
        a = ge.sweep(state, cov, reset_cache=True, want_points=True); b = g2.sweep(state, cov, reset_cache=True, want_points=True)
        print(k, "incremental map matches", a["match_num"], "full rebuild", b["match_num"], "map sizes", ge.map_size(), g2.map_size())
        ma, mb = a["plane_of_point"] >= 0, b["plane_of_point"] >= 0
        only_b = np.nonzero(mb & ~ma)[0]
        print("  matched only in rebuild:", only_b.size, "only incremental:", (ma & ~mb).sum())
        if only_b.size:
            sel = np.nonzero(lv["is_plane"] & (lv["last_num"] > 0))[0]
            lid = sel[b["plane_of_point"][only_b]]
            u = np.unique(lid)
            print("  distinct missing leaves", u.size, "last_num", lv["last_num"][u][:8], "N", lv["pcr_add"][u, 9][:8], "radius", lv["radius"][u][:5], "ids", [hex(int(x)) for x in lv["node_id"][u][:4]])
            args = lio_leaf_args(lv)
            # position of u inside sel
            pos = np.searchsorted(sel, u)
            sub = [a[pos] for a in args]
            ge.map_update(*sub)
            a2 = ge.sweep(state, cov, reset_cache=True, want_points=True)
            print("  after re-sending the missing leaves through the host path:", a2["match_num"])
            print("  layers of missing leaves", np.bincount(lv["layer"][lid], minlength=3), "in_slide", np.bincount(lv["in_slide"][lid].astype(int), minlength=2), "isexist", np.bincount(lv["isexist"][lid].astype(int), minlength=2))
    ge.pvec_update(state, cov, resident=True)
    win_count += 1; xg.append(state[:12].copy()); fg.clear()
    mg.cut_voxel_lio(win_count - 1, ge)
    mg.recut(win_count, np.stack(xg), fg)
    if win_count >= win:
        gg = vxba.Lidar_BA_Optimizer().damping_iter(np.stack(xg), fg, max_iter=3)
        mg.margi(win_count, gg["poses"], fg); mg.slide(1)
        xg = [p for p in gg["poses"][1:]]; win_count -= 1
    n1 = mg.export_planes(ge); sz1 = ge.map_size()
    n2 = mg.export_planes(ge); sz2 = ge.map_size()
    lv2 = mg.leaves()
    print(k, "exported", n1, sz1, "again", n2, sz2, "plane leaves now", int((lv2["is_plane"] & (lv2["last_num"] > 0)).sum()), "leaves in slide", int(lv2["in_slide"].sum()), "leaves", lv2["layer"].size)
