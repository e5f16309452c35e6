"""Development: one random-drive case of the map_release fuzz kind, verbose (which field of which leaf differs after which release)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from voxel_slam_amd import vxba
from tests.test_gpu_map import _corridor_scan
from tests.test_oracle_octree import PRM, point_vars, to_world
win, ptsn, S, every, age, s = [int(x) for x in sys.argv[1:7]]
for mp in (40, 60, 100):
    kw = dict(PRM); kw["max_points"] = mp
    ma, mb = vxba.LocalMap(win_size=win, **kw), vxba.LocalMap(win_size=win, **kw)
    fa, fb = vxba.LidarFactor(win), vxba.LidarFactor(win)
    r2 = np.random.default_rng(s)
    xs, wc, jour = [], 0, 0.0
    bad = None
    for k in range(S):
        body, pose = _corridor_scan(k, ptsn, r2)
        var = point_vars(body.shape[0], k)
        xs.append(pose); wc += 1
        wld = to_world(pose, body)
        for m, f in ((ma, fa), (mb, fb)):
            f.clear(); m.cut_voxel(wc - 1, body, var, wld); m.recut(wc, np.stack(xs), f)
        if wc >= win:
            for m, f in ((ma, fa), (mb, fb)):
                if f.size():
                    f.evaluate_only_residual(np.stack(xs))
                m.set_journey(jour); m.margi(wc, np.stack(xs), f); m.slide(1)
            xs = xs[1:]; wc -= 1; jour += 0.5
        if k % every == every - 1:
            rel = ma.release(jour, age)
            la, lb = ma.leaves(), mb.leaves()
            ra = np.unique(la["node_id"] >> np.uint64(16))
            keep = np.isin(lb["node_id"] >> np.uint64(16), ra)
            if int(keep.sum()) != la["node_id"].size:
                ida = set(int(i) for i in la["node_id"]); idb = set(int(i) for i in lb["node_id"][keep])
                extra_a = sorted(ida - idb)[:5]; extra_b = sorted(idb - ida)[:5]
                def xyz(i): return ((i >> 48) & 0xffff) - 32768, ((i >> 32) & 0xffff) - 32768, ((i >> 16) & 0xffff) - 32768, i & 0xffff
                bad = "scan %d (x0 = %.1f, jour %.1f): leaves %d vs %d under kept roots; only in released map %s; only in twin %s" % (
                    k, 0.5 * k, jour, la["node_id"].size, int(keep.sum()), [xyz(i) for i in extra_a], [xyz(i) for i in extra_b])
                break
            for key, v in la.items():
                if isinstance(v, np.ndarray) and v.shape[:1] == la["node_id"].shape and not np.array_equal(v, lb[key][keep]):
                    d = np.nonzero(np.any((v != lb[key][keep]).reshape(v.shape[0], -1), axis=1))[0]
                    i = int(la["node_id"][d[0]])
                    bad = "scan %d (x0 = %.1f, jour %.1f): field %s differs on %d leaves, first id root (%d, %d, %d) layer bits %d: %s vs %s" % (
                        k, 0.5 * k, jour, key, d.size, ((i >> 48) & 0xffff) - 32768, ((i >> 32) & 0xffff) - 32768, ((i >> 16) & 0xffff) - 32768, i & 0xffff,
                        np.asarray(v[d[0]]).ravel()[:6], np.asarray(lb[key][keep][d[0]]).ravel()[:6])
                    break
            if bad:
                break
    print("max_points %d: %s" % (mp, bad or "identical"))
    for h in (ma, mb, fa, fb):
        h.close()
