import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from voxel_slam_amd import synth, vxba
pm = synth.make_plane_map(n_roots=6000, extent=14, seed=5)
sc = synth.make_lio_scan(pm, n_points=40000, seed=6)
args = pm.args()
n = len(args[1])
g1 = vxba.LioEstimator(pm.voxel_size, pm.max_layer); g1.map_update(*args); g1.var_init(sc.xyz)
r1 = g1.sweep(sc.state_init, sc.cov, want_points=True)
for nb in (2, 7, 40):
    g2 = vxba.LioEstimator(pm.voxel_size, pm.max_layer)
    order = np.random.default_rng(nb).permutation(n)
    for part in np.array_split(order, nb):
        g2.map_update(*[a[part] for a in args])
    # send everything again (updates in place)
    for part in np.array_split(order[::-1], 3):
        g2.map_update(*[a[part] for a in args])
    g2.var_init(sc.xyz)
    r2 = g2.sweep(sc.state_init, sc.cov, want_points=True)
    print(nb, "batches:", r2["match_num"], "vs single", r1["match_num"], "sizes", g2.map_size(), g1.map_size(), "points differing", int(((r1["plane_of_point"] >= 0) != (r2["plane_of_point"] >= 0)).sum()))
