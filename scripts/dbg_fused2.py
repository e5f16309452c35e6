"""Round 6 debugging: fused vs three-launch *hess at the sizes of the full-steps tests.  args: W batches_per_wg_x255 [mixed]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from voxel_slam_amd import synth, vxba
np.set_printoptions(linewidth=250, precision=3, suppress=False)
W = int(sys.argv[1]); mixed = "mixed" in sys.argv
nt = (6 * W + 15) // 16
nv = {1: 12, 2: 12, 3: 12, 4: 12, 5: 10, 6: 10, 7: 8, 8: 6, 9: 6, 10: 6}[W]
for nb in [int(x) for x in sys.argv[2].split(",")]:
    V = nv * nb + 5
    sc = synth.make_scene(win_size=W, pts_per_scan=10 * V, n_voxels=V, p_obs=0.8 if W > 2 else 1.0, fix_frac=0.2, seed=1300 + W, rot_sigma_deg=0.1, trans_sigma=0.03)
    outs = []
    for fused in (0, 1):
        f = vxba.LidarFactor(sc.win_size, device=0)
        f.push_voxels(sc.clusters, sc.fix, sc.coe)
        f.evaluate_only_residual(sc.poses_init)
        f.set_option("fused_sweeps", fused)
        if mixed: f.set_precision("mixed")
        r = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=2)
        outs.append(r)
        f.close()
    a, b = outs
    d = np.abs(a["hess"] - b["hess"]) / np.abs(a["hess"]).max()
    bad = np.argwhere(d > 1e-9)
    blocks = sorted({(int(r) // 4, int(c) // 4) for r, c in bad if r <= c})
    print(f"W={W} batches {nb} (per wg {nb / 255:.2f}) V={V}: max rel diff {d.max():.4f}; 4x4 blocks that differ: {blocks}", flush=True)
