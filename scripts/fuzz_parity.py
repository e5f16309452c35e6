"""Randomised GPU-vs-oracle sweep over the library's paths (run on the GPU box): random window sizes, voxel counts, incidences, fix
clusters, perturbations.  Prints one line per case and a summary; exits non-zero on the first mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from tests import _oracle as O
from voxel_slam_amd import synth, vxba

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed0)
fails = 0

def check(cond, msg):
    global fails
    if not cond:
        fails += 1
        print("  MISMATCH:", msg, flush=True)

def marginal_flip(got, ref):
    """The first accept / reject decision the two traces disagree on was taken AT a converged state: residual1 - residual2 within 1e-9 of the residual on both
    sides (the two sums of V eigenvalues agree to ~1e-12 relative; a difference of that size has no sign).  Everything behind it may differ."""
    n = min(len(got), len(ref))
    d = np.nonzero(got[:n, 6] != ref[:n, 6])[0]
    if d.size == 0:
        return len(got) != len(ref) and n > 0 and abs(ref[n - 1, 0] - ref[n - 1, 1]) < 1e-9 * abs(ref[n - 1, 0])   # one side stopped at convergence, the other went on
    k = int(d[0])
    return abs(ref[k, 0] - ref[k, 1]) < 1e-9 * abs(ref[k, 0]) and abs(got[k, 0] - got[k, 1]) < 1e-9 * abs(got[k, 0])

t0 = time.time()
for case in range(n_cases):
    kinds = os.environ.get("FUZZ_KINDS", "lm,mixed,wide,li,gravity,lio,vox,vox_octo,vox_shard,ds,planes").split(",")   # FUZZ_KINDS=mixed,li: only those
    kind = kinds[case % len(kinds)]
    s = int(rng.integers(1, 1 << 30))
    big = kind.endswith("_big")     # lm_big / li_big / mixed_big: windows large enough for the Hessian sweep's steady-state loop (>= 8 batches per workgroup)
    if big: kind = kind[:-4]
    if kind in ("lm", "mixed", "wide", "li", "gravity"):
        W = int(rng.integers(11, 40)) if kind == "wide" else int(rng.integers(2, 11))
        V = int(rng.integers(150, 3000)); pts = int(V * rng.uniform(8, 20))   # >= 8 points per (voxel, frame): fewer make rank-deficient voxels no map would hand over
        if big:
            from tests.test_k3_mapping_model import k3_nv
            nv = k3_nv(W)        # voxels per wave-batch of the Hessian sweep (round 5: the rule of csrc/vxba_kernels.h)
            V = nv * (2048 * int(rng.integers(1, 4)) + int(rng.integers(0, 2048))) + int(rng.integers(0, nv)); pts = V * 9
        p_obs = float(rng.choice([1.0, 0.8, 0.4])) if kind != "wide" else float(rng.uniform(0.1, 0.4))
        sc = synth.make_scene(win_size=W, pts_per_scan=pts, n_voxels=V, p_obs=p_obs, fix_frac=float(rng.choice([0.0, 0.3])), seed=s,
                              rot_sigma_deg=float(rng.choice([0.05, 0.2, 0.5])), trans_sigma=float(rng.choice([0.02, 0.08])))
        fo = O.Oracle(W); fo.push_voxels(sc.clusters, sc.fix, sc.coe); fo.evaluate_only_residual(sc.poses_init)
        fg = vxba.LidarFactor(W); fg.push_voxels(sc.clusters, sc.fix, sc.coe); fg.evaluate_only_residual(sc.poses_init)
        fused_sw = int(rng.integers(0, 4) != 0)      # round 6: three in four cases through the fused residual + Hessian launch (the default), one through the three-launch iteration
        if W <= vxba.MAX_WIN: fg.set_option("fused_sweeps", fused_sw)
        iters = int(rng.integers(2, 5 if big else 8))
        if kind == "mixed":
            # f32 products on the matrix cores, f64 accumulation: same schedule, poses within 1e-5 of the fp64 oracle (contract 1e-4)
            # every other case also with the residual sweep on f32 re-centred cluster rows: the data moves by micrometres, tolerance 5e-5
            f32rows = bool(rng.integers(0, 2))
            fg.set_precision("mixed_f32_clusters" if f32rows else "mixed")
            if f32rows: fg.evaluate_only_residual(sc.poses_init)
            tol = 5e-5 if f32rows else 1e-5
            ref = fo.damping_iter(sc.poses_init, max_iter=iters, thd_num=3)
            got = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, fg, max_iter=iters)
            et, er = synth.pose_errors(got["poses"], ref["poses"])
            same = got["trace"].shape == ref["trace"].shape and np.array_equal(got["trace"][:, 6], ref["trace"][:, 6])
            if f32rows and not same and got["trace"].shape == ref["trace"].shape:
                # a decision may flip only where it was marginal: |residual1 - residual2| within the rows' rounding of the residual
                k = int(np.argmax(got["trace"][:, 6] != ref["trace"][:, 6]))
                same = abs(ref["trace"][k, 4]) < 1e-4 * abs(ref["trace"][k, 0])
            check(same, "mixed%s trace W=%d V=%d seed=%d" % ("+f32rows" if f32rows else "", W, V, s))
            check((et < tol and er < tol) or not np.array_equal(got["trace"][:, 6], ref["trace"][:, 6]), "mixed%s poses %.2e %.2e W=%d V=%d seed=%d" % ("+f32rows" if f32rows else "", et, er, W, V, s))
            desc = "W=%d V=%d iters=%d %s acc=%s pose diff %.1e/%.1e" % (W, V, iters, "f32 rows" if f32rows else "f64 rows", got["trace"][:, 6].astype(int), et, er)
        elif kind in ("lm", "wide"):
            ref = fo.damping_iter(sc.poses_init, max_iter=iters, thd_num=3)
            got = vxba.Lidar_BA_Optimizer().damping_iter(sc.poses_init, fg, max_iter=iters)
            et, er = synth.pose_errors(got["poses"], ref["poses"])
            same_tr = got["trace"].shape == ref["trace"].shape and np.array_equal(got["trace"][:, 6:], ref["trace"][:, 6:])
            marg = (not same_tr) and marginal_flip(got["trace"], ref["trace"])
            check(same_tr or marg, "%s trace W=%d V=%d seed=%d" % (kind, W, V, s))
            if marg: print("  (a decision at a converged state fell the other way: |residual1 - residual2| < 1e-9 residual; poses compared to 1e-6)", flush=True)
            if not (et < 1e-7 and er < 1e-7) and not marg:   # how much of it is conditioning?
                Hf = ref["hess"][6:, 6:]
                ev = np.linalg.eigvalsh(Hf)
                fo2 = O.Oracle(W); fo2.push_voxels(sc.clusters, sc.fix, sc.coe); fo2.evaluate_only_residual(sc.poses_init)
                fg2 = vxba.LidarFactor(W); fg2.push_voxels(sc.clusters, sc.fix, sc.coe); fg2.evaluate_only_residual(sc.poses_init)
                evo = fo2.read_cache()[0]
                gap = (evo[:, 1] - evo[:, 0]) / evo[:, 1]
                worst = np.argsort(gap)[:3]
                Ho, Jo, ro = fo2.acc_evaluate2(sc.poses_init); Hg, Jg, rg = fg2.acc_evaluate2(sc.poses_init)
                a = int(worst[0])
                Hox = fo2.acc_evaluate2(sc.poses_init, 0, a)[0] + fo2.acc_evaluate2(sc.poses_init, a + 1, V)[0]
                Hgx = fg2.acc_evaluate2(sc.poses_init, 0, a)[0] + fg2.acc_evaluate2(sc.poses_init, a + 1, V)[0]
                print("  diag: smallest eigen-gaps (l1-l0)/l1 %s at voxels %s (eigvals %s); H rel diff at the initial poses %.1e, without voxel %d: %.1e" % (
                      gap[worst], worst, evo[a], np.abs(Hg - Ho).max() / np.abs(Ho).max(), a, np.abs(Hgx - Hox).max() / np.abs(Hox).max()), flush=True)
                print("  diag: cond(H_free) %.2e, eig min %.3e max %.3e, r2 rel diff per iter %s, hess rel diff %.1e" % (ev[-1] / ev[0], ev[0], ev[-1],
                      np.abs(got["trace"][:min(len(got["trace"]), len(ref["trace"])), 1] / ref["trace"][:min(len(got["trace"]), len(ref["trace"])), 1] - 1),
                      np.abs(got["hess"] - ref["hess"]).max() / np.abs(ref["hess"]).max()), flush=True)
            check((et < 1e-7 and er < 1e-7) or (marg and et < 1e-6 and er < 1e-6), "%s poses %.2e %.2e W=%d V=%d seed=%d" % (kind, et, er, W, V, s))
            # a window whose steps are all rejected ends at its start: the Hessian and the residuals are what was computed
            hd = np.abs(got["hess"] - ref["hess"]).max() / np.abs(ref["hess"]).max()
            nt = min(len(got["trace"]), len(ref["trace"]))
            check(hd < max(1e-8, 200 * max(et, er)), "%s hess rel diff %.2e W=%d V=%d seed=%d" % (kind, hd, W, V, s))   # exported at the last accepted state: a pose difference of d metres moves it by ~ d / (plane thickness)
            # residual1 is the accepted state's; residual2 of a REJECTED step belongs to a trial state far outside the linearisation's reach,
            # where the round-off of dx is amplified by the residual's curvature: a looser bound there
            rd = np.abs(got["trace"][:nt, :2] / ref["trace"][:nt, :2] - 1)
            acc_k = ref["trace"][:nt, 6] != 0
            check(marg or (rd[:, 0].max() < 1e-8 and (rd[acc_k, 1].max() if acc_k.any() else 0) < 1e-8 and rd[:, 1].max() < 1e-5), "%s residuals rel diff %s W=%d V=%d seed=%d" % (kind, rd.max(axis=0), W, V, s))
            desc = "W=%d V=%d p_obs=%.1f iters=%d fused=%d acc=%s pose diff %.1e/%.1e hess %.1e" % (W, V, p_obs, iters, fused_sw, got["trace"][:, 6].astype(int), et, er, hd)
        else:
            iw = synth.make_imu(sc, seed=s + 1)
            bg, ba = iw.states_init[0, 15:18], iw.states_init[0, 18:21]
            blobs = O.imu_preintegrate(iw.samples, iw.noise_meas, iw.noise_walk, bg, ba)
            facs = []
            for gyr, acc, dts in iw.samples:
                fac = vxba.IMU_PRE(bg, ba)
                for g, a, dt in zip(gyr, acc, dts):
                    fac.add_imu(g, a, dt, iw.noise_meas, iw.noise_walk)
                facs.append(fac)
            iters = min(iters, 5)
            if kind == "gravity":
                st0 = iw.states_init.copy(); st0[:, 21:24] += rng.normal(0, 0.05, 3)
                ref = O.li_damping_iter_gravity(fo, st0, blobs, max_iter=iters, thd_num=5, imu_coef=1e-4)
                got = vxba.LI_BA_OptimizerGravity(imu_coef=1e-4).damping_iter(st0, fg, facs, max_iter=iters)
                check(np.allclose(got["states"][:, 21:24], ref["states"][:, 21:24], atol=1e-6), "gravity vector W=%d V=%d seed=%d" % (W, V, s))
            else:
                ref = O.li_damping_iter(fo, iw.states_init, blobs, max_iter=iters, thd_num=5, imu_coef=1e-4)
                got = vxba.LI_BA_Optimizer(imu_coef=1e-4).damping_iter(iw.states_init, fg, facs, max_iter=iters)
            et, er = synth.pose_errors(got["states"][:, :12], ref["states"][:, :12])
            same_tr = got["trace"].shape == ref["trace"].shape and np.array_equal(got["trace"][:, 6:], ref["trace"][:, 6:])
            if not same_tr:
                nt0 = min(len(got["trace"]), len(ref["trace"]))
                print("  diag: device (residual1, residual2, accept) %s" % [(float(a), float(b), int(c)) for a, b, c in got["trace"][:nt0, [0, 1, 6]]], flush=True)
                print("  diag: oracle (residual1, residual2, accept) %s" % [(float(a), float(b), int(c)) for a, b, c in ref["trace"][:nt0, [0, 1, 6]]], flush=True)
            marg = (not same_tr) and marginal_flip(got["trace"], ref["trace"])
            check(same_tr or marg, "%s trace W=%d V=%d seed=%d" % (kind, W, V, s))
            if marg: print("  (a decision at a converged state fell the other way: |residual1 - residual2| < 1e-9 residual; states compared to 1e-6)", flush=True)
            tol_s = 1e-6 if marg else 1e-7
            check(et < tol_s and er < tol_s and np.allclose(got["states"][:, 12:21], ref["states"][:, 12:21], atol=1e-6 if not marg else 1e-5), "%s states %.2e %.2e W=%d V=%d seed=%d" % (kind, et, er, W, V, s))
            hd = np.abs(got["hess"] - ref["hess"]).max() / np.abs(ref["hess"]).max()
            nt = min(len(got["trace"]), len(ref["trace"]))
            check(hd < max(1e-8, 200 * max(et, er)), "%s hess rel diff %.2e W=%d V=%d seed=%d" % (kind, hd, W, V, s))   # exported at the last accepted state: a pose difference of d metres moves it by ~ d / (plane thickness)
            # residual1 is the accepted state's; residual2 of a REJECTED step belongs to a trial state far outside the linearisation's reach,
            # where the round-off of dx is amplified by the residual's curvature: a looser bound there
            rd = np.abs(got["trace"][:nt, :2] / ref["trace"][:nt, :2] - 1)
            acc_k = ref["trace"][:nt, 6] != 0
            check(marg or (rd[:, 0].max() < 1e-8 and (rd[acc_k, 1].max() if acc_k.any() else 0) < 1e-8 and rd[:, 1].max() < 1e-5), "%s residuals rel diff %s W=%d V=%d seed=%d" % (kind, rd.max(axis=0), W, V, s))
            desc = "W=%d V=%d iters=%d acc=%s pose diff %.1e/%.1e hess %.1e" % (W, V, iters, got["trace"][:, 6].astype(int), et, er, hd)
        fg.close()
    elif kind == "lio":
        ml = int(rng.integers(0, 4)); vs = float(rng.choice([0.5, 1.0, 2.0]))
        ext = int(rng.integers(6, 10))
        pm = synth.make_plane_map(n_roots=int(rng.integers(200, int(0.5 * (2 * ext) ** 3))), extent=ext, voxel_size=vs, max_layer=ml, seed=s)
        sc = synth.make_lio_scan(pm, n_points=int(rng.integers(2000, 40000)), seed=s + 1, rot_sigma_deg=float(rng.choice([0.1, 0.5])), trans_sigma=float(rng.choice([0.02, 0.06])))
        o = O.LioOracle(vs, ml); o.map_update(*pm.args()); o.var_init(sc.xyz)
        g = vxba.LioEstimator(vs, ml); g.map_update(*pm.args()); g.var_init(sc.xyz)
        ref = o.lio_state_estimation(sc.state_init, sc.cov); got = g.lio_state_estimation(sc.state_init, sc.cov)
        et, er = synth.pose_errors(got["state"][None, :12], ref["state"][None, :12])
        check(got["iterations"] == ref["iterations"] and got["match_num"] == ref["match_num"] and got["ok"] == ref["ok"], "lio counts seed=%d" % s)
        check(et < 1e-8 and er < 1e-8, "lio pose %.2e %.2e seed=%d" % (et, er, s))
        desc = "max_layer=%d vs=%.1f planes=%d pts=%d matched=%d it=%d pose diff %.1e/%.1e" % (ml, vs, g.map_size()[1], sc.xyz.shape[0], got["match_num"], got["iterations"], et, er)
        g.close()
    elif kind in ("vox", "vox_octo"):
        W = int(rng.integers(2, 12)); pts = int(rng.integers(4000, 40000)); ml = int(rng.integers(0, 4)); vs = float(rng.choice([0.5, 1.0, 2.0]))
        xyz, fp, poses, _ = synth.make_scans(win_size=W, pts_per_scan=pts, seed=s)
        if kind == "vox":
            P = vxba.VoxelizeParams(voxel_size=vs, max_layer=ml, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
        else:
            P = vxba.VoxelizeParams(voxel_size=vs, max_layer=ml, min_points=20, min_eigen_value=0.02, eigen_ratio=(1 / 4,) * 4, min_points_layer=(20, 20, 15, 10), min_frames=0)
        ref = O.voxelize(W, xyz, fp, poses, P.as_array())
        f = vxba.LidarFactor(W); ids = f.voxelize_push(xyz, fp, poses, P)
        check(np.array_equal(np.sort(ids), ref["node_id"]), "%s factor set W=%d seed=%d (%d vs %d)" % (kind, W, s, ids.size, ref["node_id"].size))
        if ids.size and ids.size == ref["node_id"].size:
            order = np.argsort(ids)
            cl = f.read_clusters()[order]
            short = ref["clusters"][:, :, 9] <= 2048
            check(np.array_equal(cl[short], ref["clusters"][short]) and np.allclose(cl, ref["clusters"], rtol=1e-12, atol=0), "%s clusters W=%d seed=%d" % (kind, W, s))
        desc = "W=%d pts=%d max_layer=%d vs=%.1f factors=%d" % (W, pts, ml, vs, ids.size)
        f.close()
    elif kind == "vox_shard":
        # voxel-sharded voxelisation (vxba_voxelize_params.shard_*): the shards partition the unsharded factor set, every voxel bit for bit,
        # and every voxel sits on the shard dist.root_shard names; narrow and wide (compressed-row) factors
        from voxel_slam_amd import dist as vdist
        W = int(rng.integers(2, 11)) if rng.integers(0, 2) else int(rng.integers(11, 30))
        pts = int(rng.integers(3000, 25000)); ml = int(rng.integers(0, 4)); vs = float(rng.choice([0.5, 1.0, 2.0])); N = int(rng.choice([2, 3, 5, 8]))
        xyz, fp, poses, _ = synth.make_scans(win_size=W, pts_per_scan=pts, seed=s)
        P = vxba.VoxelizeParams(voxel_size=vs, max_layer=ml, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
        f = vxba.LidarFactor(W); ids = f.voxelize_push(xyz, fp, poses, P)
        full = {int(i): k for k, i in enumerate(ids)}
        cl_full = f.read_clusters() if W <= 10 and ids.size else None
        ev_full = f.read_cache()[0] if ids.size else None
        seen = 0
        for r in range(N):
            g = vxba.LidarFactor(W); ids_r = g.voxelize_push(xyz, fp, poses, P.sharded(r, N))
            check(np.all(vdist.root_shard(ids_r >> np.uint64(16), N) == r), "vox_shard: a voxel on the wrong shard W=%d N=%d seed=%d" % (W, N, s))
            check(all(int(i) in full for i in ids_r), "vox_shard: a voxel the unsharded run does not have W=%d N=%d seed=%d" % (W, N, s))
            if ids_r.size and all(int(i) in full for i in ids_r):
                sel = np.array([full[int(i)] for i in ids_r])
                check(np.array_equal(g.read_cache()[0], ev_full[sel]), "vox_shard: eigenvalues differ W=%d N=%d seed=%d" % (W, N, s))
                if cl_full is not None:
                    check(np.array_equal(g.read_clusters(), cl_full[sel]), "vox_shard: clusters differ W=%d N=%d seed=%d" % (W, N, s))
            seen += ids_r.size
            g.close()
        check(seen == ids.size, "vox_shard: shards hold %d voxels, the unsharded run %d (W=%d N=%d seed=%d)" % (seen, ids.size, W, N, s))
        desc = "W=%d pts=%d max_layer=%d vs=%.1f shards=%d factors=%d" % (W, pts, ml, vs, N, ids.size)
        f.close()
    elif kind == "map_release":
        # vxba_map_release on a random drive: a released map against an unreleased twin -- leaves under the kept roots bit for bit, after every release
        from tests.test_gpu_map import _corridor_scan
        from tests.test_oracle_octree import PRM, point_vars, to_world
        win = int(rng.integers(3, 7)); ptsn = int(rng.integers(1500, 5000)); S = int(rng.integers(80, 260)); every = int(rng.integers(7, 40)); age = int(rng.integers(26, 60))   # older than anything the corridor lets a scan touch again (cross walls are seen from up to 11 m, sparsely: gaps of 17 journeys between two hits occur) -- a released root that IS touched again starts empty, by design, and differs from the twin
        kw = dict(PRM); kw["max_points"] = int(rng.choice([40, 60, 100]))
        ma, mb = vxba.LocalMap(win_size=win, **kw), vxba.LocalMap(win_size=win, **kw)
        fa, fb = vxba.LidarFactor(win), vxba.LidarFactor(win)
        r2 = np.random.default_rng(s)
        xs, wc, jour, gone, ok = [], 0, 0.0, 0, True
        for k in range(S):
            body, pose = _corridor_scan(k, ptsn, r2)
            var = point_vars(body.shape[0], k)
            xs.append(pose); wc += 1
            wld = to_world(pose, body)
            for m, f in ((ma, fa), (mb, fb)):
                f.clear(); m.cut_voxel(wc - 1, body, var, wld); m.recut(wc, np.stack(xs), f)
            if wc >= win:
                ok = ok and fa.size() == fb.size()
                for m, f in ((ma, fa), (mb, fb)):
                    if f.size():
                        f.evaluate_only_residual(np.stack(xs))
                    m.set_journey(jour); m.margi(wc, np.stack(xs), f); m.slide(1)
                xs = xs[1:]; wc -= 1; jour += 0.5
            if k % every == every - 1:
                gone += ma.release(jour, age)["roots"]
                la, lb = ma.leaves(), mb.leaves()
                keep = np.isin(lb["node_id"] >> np.uint64(16), np.unique(la["node_id"] >> np.uint64(16)))
                ok = ok and int(keep.sum()) == la["node_id"].size and all(np.array_equal(v, lb[key][keep]) for key, v in la.items() if isinstance(v, np.ndarray) and v.shape[:1] == la["node_id"].shape)
        check(ok, "map_release: the released map differs from its twin under the kept roots (win=%d pts=%d S=%d every=%d age=%d seed=%d)" % (win, ptsn, S, every, age, s))
        desc = "win=%d pts=%d scans=%d release every %d at age %d: %d roots released, %d left" % (win, ptsn, S, every, age, gone, ma.counts()["roots"])
        for h in (ma, mb, fa, fb):
            h.close()
    elif kind == "hba":
        # the bottom-up pass of the hierarchical BA below the C ABI (vxba_hba_pass: resident keyframes, two streams, submaps on the device) against the
        # window-by-window orchestration of the same calls from Python (hba.hierarchical_ba, itself checked against the oracle in tests/test_gpu_hba.py)
        from voxel_slam_amd import hba
        wd = int(rng.integers(4, 9)); mg = int(rng.integers(2, wd)); S_ = int(rng.integers(3, 14)); K = wd + mg * (S_ - 1) + int(rng.integers(0, mg)); ptsn = int(rng.integers(2500, 6000))
        xyz, fp, poses, _ = synth.make_scans(win_size=K, pts_per_scan=ptsn, extent=24.0, noise=0.005, seed=s, rot_sigma_deg=float(rng.choice([0.05, 0.15])), trans_sigma=0.02)
        clouds = [xyz[fp[i]:fp[i + 1]].astype(np.float32) for i in range(K)]
        coarse = vxba.VoxelizeParams(voxel_size=2.0, max_layer=2, min_points=10, min_eigen_value=0.02, eigen_ratio=(1 / 9, 1 / 9, 1 / 9, 1 / 9))
        fine = vxba.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
        tmi = int(rng.integers(1, 4)); nth = int(rng.integers(1, 9))
        ref = hba.hierarchical_ba(clouds, poses, coarse, fine, wdsize=wd, mgsize=mg, top_max_iter=tmi)
        ses = vxba.HbaSession(); ses.add_keyframes(clouds)
        got = ses.run_pass(poses, coarse, fine, wdsize=wd, mgsize=mg, top_max_iter=tmi, n_threads=nth)   # (both run the closing window: tail defaults to True)
        ses.close()
        ds_ = np.abs(np.asarray(got["submap_sizes"]) - np.asarray(ref["submap_sizes"]))
        et, er = synth.pose_errors(got["submap_poses"], ref["submap_poses"])
        check(got["submap_ids"] == ref["submap_ids"] and ds_.max() <= 2, "hba: submaps differ (max %d points) K=%d wd=%d mg=%d seed=%d" % (int(ds_.max()), K, wd, mg, s))
        check(len(got["top_rounds"]) == len(ref["top_rounds"]) and all(abs(a["n_voxels"] - b["n_voxels"]) <= 2 + 0.002 * b["n_voxels"] for a, b in zip(got["top_rounds"], ref["top_rounds"])),
              "hba: top-level rounds %s vs %s K=%d seed=%d" % ([r["n_voxels"] for r in got["top_rounds"]], [r["n_voxels"] for r in ref["top_rounds"]], K, s))
        check(et < 1e-6 and er < 1e-6, "hba: submap poses %.2e %.2e K=%d wd=%d mg=%d seed=%d" % (et, er, K, wd, mg, s))
        check(len(got["edges1"]) == len(ref["edges1"]) and all((a["i"], a["j"]) == (b["i"], b["j"]) and np.allclose(a["v6"], b["v6"], rtol=1e-9) for a, b in zip(got["edges1"], ref["edges1"])),
              "hba: bottom-level edges differ K=%d seed=%d" % (K, s))
        desc = "K=%d wd=%d mg=%d pts=%d top rounds %d threads %d: %d submaps, %d + %d edges, pose diff %.1e/%.1e" % (K, wd, mg, ptsn, tmi, nth, len(got["submap_ids"]), len(got["edges1"]), len(got["edges2"]), et, er)
    elif kind == "planes":
        # per-leaf producers of the plane map: clusters, eigen-decomposition, cov_add, plane_update
        n_leaf = int(rng.integers(50, 3000))
        counts = rng.integers(8, 80, size=n_leaf)
        cell_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        nrm = rng.normal(size=(n_leaf, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        a = np.cross(nrm, [0.3, 0.5, 0.8]); a /= np.linalg.norm(a, axis=1, keepdims=True); b = np.cross(nrm, a)
        cen = rng.uniform(-30, 30, size=(n_leaf, 3))
        k = np.repeat(np.arange(n_leaf), counts)
        xyz = cen[k] + rng.uniform(-0.4, 0.4, (k.size, 1)) * a[k] + rng.uniform(-0.4, 0.4, (k.size, 1)) * b[k] + rng.normal(0, 0.01, (k.size, 1)) * nrm[k]
        M = rng.normal(size=(k.size, 3, 3)) * 0.01; var = M @ np.transpose(M, (0, 2, 1)) + np.eye(3) * 1e-5
        cl_g = vxba.build_clusters(xyz, cell_ptr); cl_o = O.build_clusters(xyz, cell_ptr)
        ev_g, U_g = vxba.plane_fit(cl_g); ev_o, U_o = O.plane_fit(cl_o)
        ca_g = vxba.cov_add_build(xyz, var, cell_ptr); ca_o = O.cov_add_build(xyz, var, cell_ptr)
        pl_g = vxba.plane_update(cl_g, ev_g, U_g, ca_g); pl_o = O.plane_update(cl_o, ev_o, U_o, ca_o)
        sgn = np.sign(np.sum(pl_g["normal"] * pl_o["normal"], axis=1))
        S = np.ones((n_leaf, 6)); S[:, :3] = sgn[:, None]
        scale_pv = np.abs(pl_o["plane_var"]).max(axis=(1, 2), keepdims=True)
        check(np.array_equal(cl_g, cl_o), "planes clusters n=%d" % n_leaf)
        ca_scale = np.abs(ca_o).max(axis=(1, 2), keepdims=True)
        ca_err = float((np.abs(ca_g - ca_o) / ca_scale).max())
        check(ca_err < 1e-11, "planes cov_add n=%d rel %.2e" % (n_leaf, ca_err))          # per-cell scale: single entries cancel
        pv_dev = np.abs(pl_g["plane_var"] * S[:, :, None] * S[:, None, :] - pl_o["plane_var"]) / scale_pv
        worst = int(np.argmax(pv_dev.max(axis=(1, 2))))
        # the radius is the largest eigenvalue ROUNDED TO FLOAT (`float radius`, voxel_map.hpp:1139): two eigensolvers (Jacobi / QL) that agree to
        # an ulp of the double may still straddle a float rounding boundary -- one float ulp (1.2e-7) apart, once in ~10^5 leaves; never more
        rad_rel = np.abs(pl_g["radius"] / pl_o["radius"] - 1)
        rad_dev = float(rad_rel.max())
        rad_ok = rad_dev < 1.3e-7 and int(np.count_nonzero(rad_rel > 1e-12)) <= 2
        check(np.all(pv_dev <= 1e-6) and rad_ok,
              "planes plane_update n=%d seed=%d: plane_var rel dev %.2e at leaf %d (eigenvalues %s, N=%d), radius rel dev %.2e" %
              (n_leaf, s, pv_dev.max(), worst, ev_o[worst], int(cl_o[worst, 9]), rad_dev))
        desc = "leaves=%d points=%d cov_add rel %.1e" % (n_leaf, k.size, ca_err)
    else:
        n = int(rng.integers(1, 300000)); size = float(rng.choice([0.05, 0.1, 0.25, 1.0])); scale = float(rng.choice([2.0, 30.0]))
        xyz = (rng.normal(size=(n, 3)) * scale).astype(np.float32)
        got = vxba.down_sampling_voxel(xyz, size); ref = O.down_sampling_voxel(xyz, size)
        check(got.shape == ref.shape and np.array_equal(got, ref), "downsample n=%d size=%.2f" % (n, size))
        desc = "n=%d size=%.2f -> %d" % (n, size, got.shape[0])
    print("case %3d %-8s %s" % (case, kind, desc), flush=True)
print("%d cases, %d mismatches, %.1f s" % (n_cases, fails, time.time() - t0))
sys.exit(1 if fails else 0)
