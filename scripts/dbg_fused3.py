"""Round 6 debugging: element sums of the Hessian sweep's workgroup partials, fused launch vs stand-alone sweep, same linearisation point."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from voxel_slam_amd import synth, vxba
W = 2; nv = 12; nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2040
V = nv * nb + 5
sc = synth.make_scene(win_size=W, pts_per_scan=10 * V, n_voxels=V, p_obs=1.0, fix_frac=0.2, seed=1300 + W, rot_sigma_deg=0.1, trans_sigma=0.03)
L = vxba.load_library()
L.vxba_debug_partials.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
PLEN = 8 * 4 * 16 + W * 28
res = {}
for fused in (0, 1):
    f = vxba.LidarFactor(W, device=0)
    f.push_voxels(sc.clusters, sc.fix, sc.coe)
    f.evaluate_only_residual(sc.poses_init)
    f.set_option("fused_sweeps", fused)
    r = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=2)   # fused: K3, fin, F(0), fin, plain -- last Hessian sweep = F(0)'s half; else K3(it 1)
    buf = np.zeros(256 * PLEN)
    assert L.vxba_debug_partials(f._h, buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    n = 256 if (fused == 0 or "256" in __import__("os").environ.get("VXBA_LIB", "")) else 255
    res[fused] = (buf.reshape(256, PLEN)[:n].copy(), r)
    f.close()
a, b = res[0][0], res[1][0]
sa, sb = a.sum(axis=0), b.sum(axis=0)
d = np.abs(sa - sb) / np.abs(sa).max()
print("element sums over workgroups: max rel diff", d.max(), "elements differing > 1e-9:", np.argwhere(d > 1e-9).ravel().tolist()[:40])
print("pair (0,0) of wave 0, sums:", sa[:16].round(3).tolist()); print("fused:                     ", sb[:16].round(3).tolist())
if a.shape == b.shape:
    dd = np.abs(a[:, :16] - b[:, :16]).max(axis=1) / np.abs(a[:, :16]).max()
    bad = np.argwhere(dd > 1e-12).ravel()
    print("workgroups whose pair (0,0) differs:", len(bad), bad.tolist()[:64])
    if len(bad):
        g = bad[0]
        print("wg", g, "standalone", a[g, :4], "fused", b[g, :4], "ratio", b[g, 0] / a[g, 0])
        print("ratios of element 0 over differing wgs:", np.round(b[bad, 0] / a[bad, 0], 4).tolist()[:40])
    d2 = np.abs(a - b).max(axis=0) / np.abs(a).max()
    print("elements differing anywhere:", np.argwhere(d2 > 1e-12).ravel().tolist()[:60])
print("hess (0,0):", res[0][1]["hess"][0, 0], res[1][1]["hess"][0, 0])
