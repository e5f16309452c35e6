"""Development: per-wave timelines of K2 / K3 from the instrumented kernels (run on the GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1]
if which == "k23":
    os.environ["VXBA_DBG"] = "2"
elif which == "k2":
    os.environ["VXBA_DBG"] = "1"
else:
    os.environ["VXBA_DBG"] = "1"
from voxel_slam_amd import synth, vxba
sc = synth.make_config("cfg2")
f = vxba.LidarFactor(sc.win_size)
f.push_points(sc.n_voxels, sc.points_body, sc.cell_ptr)
f.evaluate_only_residual(sc.poses_init)
for _ in range(3):
    f.acc_evaluate2(sc.poses_init); f.evaluate_only_residual(sc.poses_init)
vxba.debug_stamps(1, clear=True)
if which == "solve":
    from voxel_slam_amd.vxba import Lidar_BA_Optimizer
    Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=1)
    st = vxba.debug_stamps(4001).astype(np.int64)[4000, :6]
    print("solve kernel stamps (cycles since start):", st - st[0])
    sys.exit(0)
if which in ("fused", "fused_li"):
    from voxel_slam_amd.vxba import Lidar_BA_Optimizer
    if which == "fused_li":   # the residual-sweep launch of the LiDAR-inertial shell (pose system solved in the launch from the host's record)
        f.snapshot_cache()
        iw = synth.make_imu(sc)
        facs = []
        for gyr, acc, dts in iw.samples:
            fac = vxba.IMU_PRE(iw.states_init[0, 15:18], iw.states_init[0, 18:21])
            for g, a, dt in zip(gyr, acc, dts):
                fac.add_imu(g, a, dt, iw.noise_meas, iw.noise_walk)
            facs.append(fac)
        vxba.LI_BA_Optimizer().damping_iter(iw.states_init, f, facs, max_iter=1)
    else:
        Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=1)
    full = vxba.debug_stamps(4096).astype(np.int64)
    sol = full[4000, :6]
    n = (sc.n_voxels + 63) // 64
    k2 = full[:n, :6]
    k2 = k2[k2[:, 0] > 0]
    t0 = min(sol[0], k2[:, 0].min())
    print("solver stamps (cycles since kernel start): start %d, done-checked %d, loaded %d, factored %d, back-substituted %d, end %d" % tuple(sol - t0))
    ent = full[:n, 6]; ent = ent[ent > 0]
    if ent.size:   # kernel entry stamps (instrumented build): everything below relative to the first wave to enter
        t0 = min(t0, ent.min(), full[4000, 30] if full[4000, 30] > 0 else t0)
        fin = full[:n, 7]; fin = fin[fin > 0]
        print("kernel entry: voxel waves min %d median %d; solve workgroup %d" % (ent.min() - t0, np.median(ent) - t0, full[4000, 30] - t0))
        if fin.size:
            print("in-launch Hessian reduction: share written + counted, waves min %d median %d max %d; solve workgroup released at %d" % (
                fin.min() - t0, np.median(fin) - t0, fin.max() - t0, full[4000, 31] - t0))
        print("solver stamps again, since kernel entry: start %d, loaded %d, factored %d, back-substituted %d, end %d" % tuple((sol - t0)[[0, 2, 3, 4, 5]]))
    names = ["start", "loads landed", "cov done", "eig done", "end", "flag seen"]
    for k in (0, 5, 1, 2, 3, 4):
        print("voxel waves %-13s min %7d  median %7d  max %7d" % (names[k], k2[:, k].min() - t0, np.median(k2[:, k]) - t0, k2[:, k].max() - t0))
    # four-wave solve (vxba_solve4.hpp): per wave, arrival at / release from the barrier of every block step; per step, the owner's chain
    wv = full[4000:4004, 6:24]
    if (wv > 0).any():
        for w in range(4):
            print("wave %d  barrier arrive/leave: " % w + "  ".join("%d/%d" % (wv[w, 2 * s_] - t0, wv[w, 2 * s_ + 1] - t0) for s_ in range(9) if wv[w, 2 * s_] > 0))
        for s_ in range(9):
            ch = full[4010 + s_, :5]
            if ch[0] > 0:
                print("step %d owner chain: start %d  applied +%d  diagonal read +%d  factored +%d  panel stored +%d" % ((s_, ch[0] - t0) + tuple(np.diff(ch))))
    sys.exit(0)
if which == "k23":
    # the fused launch (vxba_k23.hpp): second iteration of a 3-iteration solve.  Sweep waves: 0 entry, 30 rows requested, 5 poses in LDS + barrier,
    # 15 transform done, 18 eigen done, 21 cache stores issued, 24 stores acknowledged, then the Hessian half's stamps (2 first requests, 1 barrier, 8+s ..)
    from voxel_slam_amd.vxba import Lidar_BA_Optimizer
    Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=2)
    full = vxba.debug_stamps(4096).astype(np.int64)
    sol = full[4000, :6]
    live = (full[:2048, 0] > 0) & (full[:2048, 6] > 0)
    fl = full[:2048][live]
    wgs = fl[:, 0].reshape(-1, 8).min(axis=1).repeat(8) if fl.shape[0] % 8 == 0 else fl[:, 0]
    print("fused launch: %d sweep waves stamped; solver (cycles since its start): loaded %d, factored %d, back-substituted %d, end %d" % ((fl.shape[0],) + tuple((sol - sol[0])[[2, 3, 4, 5]])))
    wv = np.arange(fl.shape[0]) % 8
    for nm, slot in (("rows requested", 30), ("poses in LDS, barrier passed", 5), ("rows landed (pair mode)", 4), ("transform done", 15), ("eigen done", 18), ("cache stores issued", 21), ("residual half done", 24),
                     ("first batch requested", 2), ("Hessian-half barrier passed", 1), ("barrier 0 (phase A of step 0)", 8), ("barrier 1", 9), ("barrier 2", 10), ("barrier 3", 11),
                     ("barrier 4", 12), ("step loop left", 3), ("tiles-done barrier", 27), ("accumulators parked", 28), ("partial stores issued", 6), ("acknowledged", 31)):
        ok = fl[:, slot] > 0
        if ok.any():
            col = (fl[:, slot] - wgs)[ok]
            print("since the workgroup's first wave entered: %-30s median %6.0f  p10 %6.0f  p90 %6.0f  max %6.0f   (%d waves; waves 0-3 median %6.0f)" % (
                nm, np.median(col), np.percentile(col, 10), np.percentile(col, 90), col.max(), int(ok.sum()), np.median((fl[:, slot] - wgs)[ok & (wv < 4)]) if (ok & (wv < 4)).any() else -1))
    # relative to the moment the poses arrived (slot 5): what the launch costs BEHIND the solve
    ok = (fl[:, 5] > 0) & (fl[:, 31] > 0)
    if ok.any():
        for nm, slot in (("rows landed (pair mode)", 4), ("transform done", 15), ("eigen done", 18), ("cache stores issued", 21), ("residual half done", 24), ("Hessian-half barrier passed", 1), ("barrier 0", 8), ("barrier 1", 9), ("barrier 2", 10), ("barrier 3", 11), ("step loop left", 3), ("acknowledged (end)", 31)):
            k = ok & (fl[:, slot] > 0)
            if k.any():
                col = fl[k, slot] - fl[k, 5]
                print("since the poses arrived: %-30s median %6.0f  p10 %6.0f  p90 %6.0f  max %6.0f" % (nm, np.median(col), np.percentile(col, 10), np.percentile(col, 90), col.max()))
    sys.exit(0)
if which == "fin":
    # cross-workgroup reduction of the Hessian sweep (k3_finalize_kernel): wave 0 of every workgroup, rows 3000.. of the stamp table:
    # 0 entry, 1 loads landed and summed, 2 reduced through LDS and outputs issued, 3 outputs acknowledged
    f.acc_evaluate2(sc.poses_init)
    full = vxba.debug_stamps(3400).astype(np.int64)[3000:3400, :4]
    full = full[full[:, 0] > 0]
    d = np.diff(full, axis=1)
    print("finalize: %d workgroups; per workgroup, cycles: entry -> loads summed  median %d p10 %d p90 %d | -> reduced, outputs issued  median %d | -> acknowledged  median %d | total median %d max %d" % (
        full.shape[0], np.median(d[:, 0]), np.percentile(d[:, 0], 10), np.percentile(d[:, 0], 90), np.median(d[:, 1]), np.median(d[:, 2]), np.median(full[:, 3] - full[:, 0]), (full[:, 3] - full[:, 0]).max()))
    for x in range(8):   # same-XCD workgroups share a clock: spread of the entry and of the end inside one XCD
        g = full[x::8]
        print("  XCD %d: %d workgroups, entry spread %d cycles, first entry -> last end %d cycles" % (x, g.shape[0], g[:, 0].max() - g[:, 0].min(), g[:, 3].max() - g[:, 0].min()))
    sys.exit(0)
if which == "k2":
    f.evaluate_only_residual(sc.poses_init); n = (sc.n_voxels + 63) // 64; ns = 5
else:
    # Hessian sweep (vxba_k3.hpp): 8 waves per workgroup; stamps 0 start, 1 poses decided, 8+s after the barrier of step s (s < 6),
    # 3 step loop left, 6 partial written
    if which == "k3lm":   # the sweep as the LM loop runs it: second iteration, accept/reject decision of the first in its prologue
        from voxel_slam_amd.vxba import Lidar_BA_Optimizer
        Lidar_BA_Optimizer().damping_iter(sc.poses_init, f, max_iter=2)
    else:
        f.acc_evaluate2(sc.poses_init)
    full = vxba.debug_stamps(2048).astype(np.int64)
    order = [0, 1, 8, 9, 10, 11, 12, 13, 3, 6]
    names = ["start", "poses decided", "barrier 0 (phase A of step 0 done)", "barrier 1", "barrier 2", "barrier 3", "barrier 4", "barrier 5", "loop left", "end"]
    live = full[:, 6] > 0
    st = full[live][:, order]
    t0 = st[:, 0].min()
    print("k3: waves stamped", int(live.sum()), " kernel span %.2f us (s_memtime at 100 MHz)" % ((st[:, -1].max() - t0) / 100.0))
    for k, nm in enumerate(names):
        col = st[:, k]; ok = col > 0
        if ok.any():
            print("%-36s min %7.2f  median %7.2f  max %7.2f us   (%d waves)" % (nm, (col[ok].min() - t0) / 100.0, (np.median(col[ok]) - t0) / 100.0, (col[ok].max() - t0) / 100.0, int(ok.sum())))
    # prologue, per wave and relative to the wave's own start (the cycle counters of the eight XCDs have different origins, so only
    # differences inside a workgroup mean anything): 2 first requests issued, 5 poses in LDS (wave 0), 4 tiles cleared, 1 barrier passed
    fl = full[live]
    wv = np.arange(fl.shape[0]) % 8
    wg_start = fl[:, 0].reshape(-1, 8).min(axis=1).repeat(8)
    for nm, slot in (("wave start", 0), ("first requests issued", 2), ("poses in LDS (wave 0)", 5), ("tiles cleared", 4), ("prologue barrier passed", 1), ("barrier 0 passed", 8)):
        col = fl[:, slot] - wg_start
        ok = fl[:, slot] > 0
        if ok.any():
            print("since the workgroup's first wave started: %-26s median %6.0f  p10 %6.0f  p90 %6.0f   waves 0-3 median %6.0f  waves 4-7 median %6.0f" % (
                nm, np.median(col[ok]), np.percentile(col[ok], 10), np.percentile(col[ok], 90),
                np.median(col[ok & (wv < 4)]) if (ok & (wv < 4)).any() else -1, np.median(col[ok & (wv >= 4)]) if (ok & (wv >= 4)).any() else -1))
    # epilogue, per wave since it left the step loop (slot 3): 27 tiles-done barrier passed, 28 accumulators parked, 29 second barrier passed,
    # 30 linear sums written, 6 partial stores issued, 31 ... and acknowledged
    for nm, slot in (("tiles-done barrier passed", 27), ("accumulators parked", 28), ("second barrier passed", 29), ("linear sums written", 30), ("partial stores issued", 6), ("stores acknowledged", 31)):
        ok = (fl[:, slot] > 0) & (fl[:, 3] > 0)
        if ok.any():
            col = fl[ok, slot] - fl[ok, 3]
            print("epilogue, since the wave left the step loop: %-28s median %6.0f  p10 %6.0f  p90 %6.0f" % (nm, np.median(col), np.percentile(col, 10), np.percentile(col, 90)))
    per = np.diff(st[:, 2:7], axis=1) / 100.0
    okp = (st[:, 2:7] > 0).all(axis=1)
    if okp.any():
        print("step period (barrier to barrier) us: median per step", np.round(np.median(per[okp], axis=0), 2))
    # inside steps 1..3: barrier s-1 passed -> phase M done -> phase A done -> barrier s passed
    for st_ in (1, 2, 3):
        cols = full[live][:, [8 + st_ - 1, 13 + 3 * st_, 14 + 3 * st_, 8 + st_]]
        okc = (cols > 0).all(axis=1)
        if okc.any():
            d = np.diff(cols[okc], axis=1)
            print("step %d (cycles, median | max over waves): phase M %6.0f | %6.0f   phase A %6.0f | %6.0f   wait at barrier %6.0f | %6.0f" % (
                st_, np.median(d[:, 0]), d[:, 0].max(), np.median(d[:, 1]), d[:, 1].max(), np.median(d[:, 2]), d[:, 2].max()))
    # the eight waves of three workgroups, step 2: cycles since barrier 1 was passed by the first of them (waves w and w + 4 share a SIMD)
    for wg in (0, 100, 200):
        blk = full[8 * wg:8 * wg + 8]
        t0w = blk[:, 9].min()
        print("workgroup %d, step 2:  wave: M done / A done / barrier 2 passed (cycles since the first wave left barrier 1)" % wg)
        print("   " + "  ".join("w%d: %5d/%5d/%5d" % (w_, blk[w_, 19] - t0w, blk[w_, 20] - t0w, blk[w_, 10] - t0w) for w_ in range(8)))
    # inside phase A of step 2 (round 4): M done (19) -> parameters + pose back in registers (7) -> rows computed (14) -> rows stored (15) ->
    # block-diagonal accumulators done (18) -> next batch requested = A done (20)
    cols = full[live][:, [19, 7, 14, 15, 18, 20]]
    okc = (cols > 0).all(axis=1)
    if okc.any():
        d = np.diff(cols[okc], axis=1)
        wvo = wv[okc]
        for nm, sel in (("all waves", wvo >= 0), ("waves 0-3", wvo < 4), ("waves 4-7", wvo >= 4)):
            print("phase A of step 2, %-9s (cycles, median): unstage + pose %5.0f | rows computed %5.0f | rows stored %5.0f | rest of the accumulators %5.0f | requests issued %5.0f" % (
                (nm,) + tuple(np.median(d[sel], axis=0))))
    sys.exit(0)
full = vxba.debug_stamps(n).astype(np.int64)
st = full[:, :ns]
t0 = st[:, 0].min()
rel = (st - t0) / 100.0     # s_memtime ticks at 100 MHz -> us
print(which, "waves", n, "kernel span us:", (st[:, -1].max() - t0) / 100.0)
print("per-slot (us since first wave start): min / median / max")
for k in range(ns):
    print(k, "%.2f %.2f %.2f" % (rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max()))
d = np.diff(st, axis=1) / 100.0
print("per-phase durations (us): median", np.median(d, axis=0), "max", d.max(axis=0))
