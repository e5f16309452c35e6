"""The cold-L3 rotation of bench.py's roofline.cold_l3 leg on its own (for rocprofv3: every K2 / K3 launch in the trace is a cold one):
six cfg2 factors visited round-robin, K2 then K3 on each -- 500 MB of other factors' planes pass between two visits of a factor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bench
from voxel_slam_amd import synth, vxba
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
sc = synth.make_config(cfg)
f = vxba.LidarFactor(sc.win_size)
f.push_points(sc.n_voxels, sc.points_body, sc.cell_ptr)
f.evaluate_only_residual(sc.poses_init)
f.snapshot_cache()
print(bench.cold_l3_leg(sc, f, 0, "f64", rounds=int(os.environ.get("ROUNDS", "12"))))
