"""Round 6: the fused [solve | residual sweep | Hessian sweep] launch (VXBA_OPT_FUSED_SWEEPS) against the three-launch iteration on the same
windows -- LM trace, poses, *hess, cache; then the step rate of both through vxba_lm_steps.  Usage: dbg_fused.py [parity] [rate] [cfgs...]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from voxel_slam_amd import synth, vxba


def factor(sc, fused):
    f = vxba.LidarFactor(sc.win_size, device=0)
    f.push_voxels(sc.clusters, sc.fix, sc.coe)
    f.evaluate_only_residual(sc.poses_init)
    f.set_option("fused_sweeps", fused)
    return f


def parity(name, sc, max_iter):
    out = {}
    for fused in (0, 1):
        f = factor(sc, fused)
        r = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=max_iter)
        r["cache"] = f.read_cache()
        out[fused] = r
        f.close()
    a, b = out[0], out[1]
    same_flags = a["trace"].shape == b["trace"].shape and np.array_equal(a["trace"][:, 6:], b["trace"][:, 6:])
    et, er = synth.pose_errors(a["poses"], b["poses"])
    tr = np.abs(a["trace"][:, :6] - b["trace"][:, :6]).max() if same_flags else np.nan
    hs = np.abs(a["hess"] - b["hess"]).max() / max(1e-300, np.abs(a["hess"]).max())
    cm = max(np.abs(x - y).max() for x, y in zip(a["cache"], b["cache"]))
    print(f"{name}: iters {a['trace'].shape[0]} / {b['trace'].shape[0]} accept {a['trace'][:, 6].astype(int).tolist()} / {b['trace'][:, 6].astype(int).tolist()} flags_same {same_flags} "
          f"pose diff {et:.2e} m {er:.2e} rad, trace max abs diff {tr:.2e}, hess rel {hs:.2e}, cache max abs {cm:.2e}, bitwise poses {np.array_equal(a['poses'], b['poses'])}", flush=True)
    return same_flags and et < 1e-10 and er < 1e-10


def rate(name, sc, steps=150, sps=3):
    for fused in (0, 1, 0, 1):
        f = factor(sc, fused)
        f.snapshot_cache()
        f.lm_steps(sc.poses_init, 30, sps)
        t0 = time.perf_counter()
        p, r, st = f.lm_steps(sc.poses_init, steps, sps)
        dt = time.perf_counter() - t0
        f.set_profiling(1 | 2 | 4 | 32)
        f.lm_steps(sc.poses_init, steps, sps)
        kt = f.kernel_times(reset=True); ft = f.fused_time(reset=True)
        f.set_profiling(0)
        us = lambda d: 1e3 * d["ms_sum"] / max(1, d["calls"])
        print(f"{name} fused={fused}: {steps / dt:8.0f} it/s  {1e6 * dt / steps:6.2f} us/step  accepted {st['accepted']} rejected {st['rejected']}  | k3 {us(kt['k3_hessian']):.2f} us x{kt['k3_hessian']['calls']}  "
              f"k2-launch {us(kt['k2_residual']):.2f} x{kt['k2_residual']['calls']}  fin {us(kt['k3_finalize']):.2f} x{kt['k3_finalize']['calls']}  fused {us(ft):.2f} x{ft['calls']}", flush=True)
        f.close()


if __name__ == "__main__":
    args = sys.argv[1:]
    do_par = "parity" in args or not any(a in ("parity", "rate") for a in args)
    do_rate = "rate" in args or not any(a in ("parity", "rate") for a in args)
    cfgs = [a for a in args if a not in ("parity", "rate")] or ["cfg1", "cfg2"]
    ok = True
    if do_par:
        ok &= parity("w10 small", synth.make_scene(win_size=10, pts_per_scan=20000, n_voxels=2000, seed=81), 3)
        ok &= parity("w10 sparse fix rejections", synth.make_scene(win_size=10, pts_per_scan=30000, n_voxels=3000, p_obs=0.7, fix_frac=0.3, seed=71, rot_sigma_deg=0.1, trans_sigma=0.02), 8)
        for W in (2, 3, 5, 7, 9):
            ok &= parity(f"W={W}", synth.make_scene(win_size=W, pts_per_scan=12000, n_voxels=1200, p_obs=0.8 if W > 2 else 1.0, fix_frac=0.2, seed=900 + W, rot_sigma_deg=0.1, trans_sigma=0.03), 5)
        ok &= parity("tiny (7 voxels)", synth.make_scene(win_size=4, pts_per_scan=400, n_voxels=7, seed=5), 4)
    for c in cfgs:
        sc = synth.make_config(c)
        if do_par:
            ok &= parity(c, sc, 4)
        if do_rate:
            rate(c, sc)
    print("PARITY", "OK" if ok else "MISMATCH")
