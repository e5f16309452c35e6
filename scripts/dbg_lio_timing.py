"""Timing of the odometry sweep / state estimation on the GPU box (run under rocprofv3 --kernel-trace --stats for kernel times)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import torch  # noqa: F401  (first: one HIP runtime in the process)
from voxel_slam_amd import synth, vxba

n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
pm = synth.make_plane_map(n_roots=20_000, extent=20, seed=synth.MASTER_SEED + 910)
mode = sys.argv[2] if len(sys.argv) > 2 else "shuffled"
kw = {"shuffled": {}, "coherent": dict(coherent=True, planes_hit=8000), "few_planes": dict(planes_hit=8000)}[mode]
sc = synth.make_lio_scan(pm, n_points=n_points, seed=synth.MASTER_SEED + 911, **kw)
g = vxba.LioEstimator(pm.voxel_size, pm.max_layer)
g.map_update(*pm.args()); g.var_init(sc.xyz)
print("scan order:", mode)
for name, fn in (("sweep", lambda: g.sweep(sc.state_init, sc.cov)), ("state_estimation", lambda: g.lio_state_estimation(sc.state_init, sc.cov))):
    ts = []
    for k in range(40):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    print("%s: median %.1f us, min %.1f us" % (name, 1e6 * np.median(ts[5:]), 1e6 * np.min(ts)))
