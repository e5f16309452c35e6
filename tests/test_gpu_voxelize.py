"""GPU parity of the batch factor construction (vxba_voxelize_push: OctreeGBA::cut_voxel + subdivide + recut,
loop_refine.hpp:273-476) against the CPU oracle's hash-map / octree restatement on the same raw scans.

Integer work is exact: the set of voxels that become factors (root cell, octant path, layer) and every point count must be
identical, and the cluster sums are bit-identical (same summation order as PointCluster::push).  Eigen-decompositions
differ by solver round-off only (Jacobi on the device, Eigen-style QL in the oracle)."""
import numpy as np
import pytest

from tests import _oracle as O
from voxel_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vx():
    from voxel_slam_amd import vxba
    vxba.load_library()
    return vxba


def run_both(vx, W, pts, params, **kw):
    xyz, fp, poses, gt = synth.make_scans(win_size=W, pts_per_scan=pts, **kw)
    ref = O.voxelize(W, xyz, fp, poses, params.as_array())
    f = vx.LidarFactor(W)
    ids = f.voxelize_push(xyz, fp, poses, params)
    return xyz, fp, poses, gt, ref, f, ids


@pytest.mark.parametrize("W,pts,max_layer,vs,seed", [(5, 20_000, 2, 1.0, 1), (10, 60_000, 3, 2.0, 2), (3, 5_000, 0, 1.0, 3), (4, 150_000, 2, 0.5, 4)])
def test_factor_set_and_clusters_match_oracle(vx, W, pts, max_layer, vs, seed):
    P = vx.VoxelizeParams(voxel_size=vs, max_layer=max_layer, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
    xyz, fp, poses, gt, ref, f, ids = run_both(vx, W, pts, P, seed=synth.MASTER_SEED + 600 + seed)
    assert f.size() == ids.shape[0]
    assert ids.shape[0] > 50
    layers = (ids & np.uint64(7)).astype(int)
    assert np.all(np.diff(layers) >= 0) and layers.max() <= max_layer          # pushed layer by layer
    if max_layer >= 2:
        assert (layers >= 1).any()                                              # the subdivision path is exercised
    order = np.argsort(ids, kind="stable")
    assert np.array_equal(ids[order], ref["node_id"]), (ids.shape, ref["node_id"].shape)   # the same voxels become factors
    cl = f.read_clusters()[order]
    ev, U, m = f.read_cache()
    ev, U, m = ev[order], U[order], m[order]
    # bit-exact clusters (body-frame per frame, zeros where unobserved; world per node) wherever a cell holds <= 2048 points -- the
    # reference's sequential sum; longer cells are folded by a whole workgroup in another fixed order
    short_c = ref["clusters"][:, :, 9] <= 2048; short_m = ref["merged"][:, 9] <= 2048
    assert np.array_equal(cl[short_c], ref["clusters"][short_c]) and np.array_equal(m[short_m], ref["merged"][short_m])
    assert short_m.mean() > 0.5 and np.array_equal(cl[:, :, 9], ref["clusters"][:, :, 9]) and np.array_equal(m[:, 9], ref["merged"][:, 9])
    assert np.allclose(cl, ref["clusters"], rtol=1e-12, atol=0) and np.allclose(m, ref["merged"], rtol=1e-12, atol=0)
    # every factor satisfies recut's criteria
    assert np.all(m[:, 9] > 10) and np.all(ev[:, 0] < 0.01) and np.all(ev[:, 0] / ev[:, 1] <= 0.12 + 1e-12)
    assert np.all((cl[:, :, 9] > 0).sum(axis=1) >= 2)
    scale = np.abs(m[:, :6]).max(axis=1) / m[:, 9] + 1.0                        # |vbar|^2-sized cancellation in cov()
    assert np.all(np.abs(ev - ref["eig_val"]) <= 1e-13 * scale[:, None])
    n_g = U.reshape(-1, 3, 3)[:, 0, :]; n_o = ref["eig_vec"].reshape(-1, 3, 3)[:, 0, :]   # plane normals (column 0), up to sign
    gap = (ev[:, 1] - ev[:, 0]) > 1e-6
    assert np.all(np.abs(np.abs(np.sum(n_g * n_o, axis=1)) - 1)[gap] < 1e-8)


def test_negative_and_boundary_coordinates_follow_the_reference_rule(vx):
    """float quotient, '-1 if negative', truncation: -0.5 -> cell -1, but also -2.0 -> cell -3 (upstream quirk, kept)."""
    W = 2
    P = vx.VoxelizeParams(voxel_size=1.0, max_layer=0, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16,) * 4)
    rng = np.random.default_rng(5)
    def patch(cx):       # a small horizontal plane patch around x = cx, 40 points per frame
        return np.stack([cx + rng.uniform(-0.2, 0.2, 40), 3.3 + rng.uniform(-0.2, 0.2, 40), 0.5 + rng.normal(0, 0.003, 40)], axis=1)
    cloud = np.concatenate([patch(-0.5), patch(-2.2), np.tile([[-2.0, 7.3, 0.5]], (12, 1)) + rng.normal(0, 1e-3, (12, 3)) * [0, 1, 0.05]])
    xyz = np.ascontiguousarray(np.concatenate([cloud, cloud + 1e-3]))
    fp = np.array([0, cloud.shape[0], 2 * cloud.shape[0]], dtype=np.int64)
    poses = synth.pack_poses(np.stack([np.eye(3)] * 2), np.zeros((2, 3)))
    ref = O.voxelize(W, xyz, fp, poses, P.as_array())
    f = vx.LidarFactor(W)
    ids = f.voxelize_push(xyz, fp, poses, P)
    assert np.array_equal(np.sort(ids), ref["node_id"])
    xs = ((ids >> np.uint64(48)) & np.uint64(0xffff)).astype(np.int64) - 32768
    assert set(xs.tolist()) >= {-1, -3}


def test_out_of_range_and_argument_errors(vx):
    W = 2
    P = vx.VoxelizeParams(voxel_size=1.0, max_layer=1)
    xyz = np.array([[1e6, 0, 0]] * 30, dtype=np.float64)
    fp = np.array([0, 15, 30], dtype=np.int64)
    poses = synth.pack_poses(np.stack([np.eye(3)] * 2), np.zeros((2, 3)))
    f = vx.LidarFactor(W)
    with pytest.raises(vx.VxbaError, match="range"):
        f.voxelize_push(xyz, fp, poses, P)
    with pytest.raises(vx.VxbaError):
        f.voxelize_push(xyz, np.array([0, 15, 29], dtype=np.int64), poses, P)
    assert f.size() == 0
    assert f.voxelize_push(np.zeros((0, 3)), np.zeros(3, dtype=np.int64), poses, P).shape[0] == 0


def test_ba_on_voxelized_window_matches_oracle_and_recovers_poses(vx):
    """End to end, as one round of the hierarchical BA does (voxelslam.cpp:2374-2384): voxelise at the current poses, then
    Lidar_BA_Optimizer::damping_iter -- on the GPU against the oracle fed with the oracle's own factor list."""
    W = 6
    P = vx.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
    xyz, fp, poses, gt, ref, f, ids = run_both(vx, W, 40_000, P, seed=synth.MASTER_SEED + 650, rot_sigma_deg=0.05, trans_sigma=0.01, noise=0.005)
    order = np.argsort(ids, kind="stable")
    inv = np.empty_like(order); inv[order] = np.arange(order.size)
    fo = O.Oracle(W)
    # same voxel order as the GPU factor so that both LM runs see the same problem; cache seeded by recut's eig, as upstream
    fo.push_voxels(ref["clusters"][inv], np.zeros((ids.size, 10)), np.ones(ids.size), ref["eig_val"][inv], ref["eig_vec"][inv], ref["merged"][inv])
    r = fo.damping_iter(poses, max_iter=5, thd_num=4)
    g = vx.Lidar_BA_Optimizer().damping_iter(poses, f, max_iter=5)
    assert np.array_equal(g["trace"][:, 6:], r["trace"][:, 6:])
    et, er = synth.pose_errors(g["poses"], r["poses"])
    assert et < 1e-7 and er < 1e-7
    e0 = synth.pose_errors(poses, gt); e1 = synth.pose_errors(g["poses"], gt)
    assert e1[0] < 0.5 * e0[0] and e1[1] < 0.5 * e0[1]


@pytest.mark.parametrize("W,pts,seed", [(5, 30_000, 11), (10, 40_000, 12)])
def test_octotree_batch_build_matches_oracle(vx, W, pts, seed):
    """OctoTree's criteria instead of OctreeGBA's: the map build of motion_init (cut_voxel for every scan, one recut + tras_opt,
    voxelslam.cpp:606-625) -- point-count floor per layer (min_point[layer]), a single observing frame is enough."""
    P = vx.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=20, min_eigen_value=0.02, eigen_ratio=(1 / 4, 1 / 4, 1 / 4, 1 / 4),
                          min_points_layer=(20, 20, 15, 10), min_frames=0)
    xyz, fp, poses, gt = synth.make_scans(win_size=W, pts_per_scan=pts, seed=synth.MASTER_SEED + 700 + seed)
    # a wall only the last frame sees: single-frame voxels, factors for OctoTree, not for the GBA
    rng = np.random.default_rng(seed)
    wall = np.stack([rng.uniform(40, 46, 4000), rng.uniform(-3, 3, 4000), np.full(4000, 30.0) + rng.normal(0, 0.005, 4000)], axis=1)
    R = poses[W - 1, :9].reshape(3, 3).T; t = poses[W - 1, 9:12]
    xyz = np.ascontiguousarray(np.concatenate([xyz, (wall - t) @ R]))
    fp = fp.copy(); fp[W] += 4000
    ref = O.voxelize(W, xyz, fp, poses, P.as_array())
    f = vx.LidarFactor(W); ids = f.voxelize_push(xyz, fp, poses, P)
    assert ids.shape[0] > 100 and np.array_equal(np.sort(ids), ref["node_id"])
    order = np.argsort(ids)
    assert np.array_equal(f.read_clusters()[order], ref["clusters"])
    _, _, merged = f.read_cache()
    assert np.array_equal(merged[order], ref["merged"])
    n_obs = (ref["clusters"][:, :, 9] > 0).sum(axis=1)
    layers = (ref["node_id"] & np.uint64(7)).astype(int)
    assert (n_obs == 1).sum() >= 10                                      # the single-frame wall became factors ...
    assert ((ref["merged"][:, 9] <= 20) & (layers == 2)).any()           # ... and layer 2 accepts leaves under the layer-0 floor
    G = vx.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=20, min_eigen_value=0.02, eigen_ratio=(1 / 4, 1 / 4, 1 / 4, 1 / 4))
    f2 = vx.LidarFactor(W); ids2 = f2.voxelize_push(xyz, fp, poses, G)
    refg = O.voxelize(W, xyz, fp, poses, G.as_array())
    assert np.array_equal(np.sort(ids2), refg["node_id"]) and not set(ids[n_obs[np.argsort(np.argsort(ids))] == 1].tolist()) & set(ids2.tolist())


def test_cluster_build_with_very_long_cells(vx):
    """A coarse top-level voxelisation puts 10^5 points into one node: such cells are folded by a whole workgroup (fixed order,
    ~1e-16 from the sequential sum) instead of one lane; cells up to 2048 points keep the reference's sequential sum bit for bit."""
    rng = np.random.default_rng(77)
    # around the sixteen lanes a packed cell is folded by (15 / 16 / 17 / 31 / 32 / 33 / 48), empty cells beside full ones in one wave's four rows,
    # and both sides of the long-cell threshold
    lens = np.array([5, 300_000, 17, 2048, 2049, 1, 0, 70_001, 33, 16, 15, 0, 32, 31, 48, 0, 0, 2047, 64, 2])
    cell_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    xyz = rng.normal(size=(int(lens.sum()), 3)) * 20 + 5
    got = vx.build_clusters(xyz, cell_ptr); ref = O.build_clusters(xyz, cell_ptr)
    short = lens <= 2048
    assert np.array_equal(got[short], ref[short])
    assert np.array_equal(got[:, 9], ref[:, 9])
    assert np.allclose(got[~short], ref[~short], rtol=1e-12, atol=0)
    assert np.array_equal(vx.build_clusters(xyz, cell_ptr), got)            # deterministic


def test_voxelize_from_device_memory_and_repeated_calls(vx):
    """vxba_voxelize_push_device on a torch tensor gives the same factor as the host-pointer call; repeated calls on one factor
    (clear in between) reuse the scratch and stay exact."""
    import torch
    W = 6
    P = vx.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
    f = vx.LidarFactor(W); f2 = vx.LidarFactor(W)
    for rep, pts in enumerate((20_000, 45_000, 8_000)):
        xyz, fp, poses, _ = synth.make_scans(win_size=W, pts_per_scan=pts, seed=synth.MASTER_SEED + 780 + rep)
        f.clear(); f2.clear()
        ids = f.voxelize_push(xyz, fp, poses, P)
        t = torch.from_numpy(xyz).cuda()
        ids2 = f2.voxelize_push_device(t.data_ptr(), xyz.shape[0], fp, poses, P)
        ref = O.voxelize(W, xyz, fp, poses, P.as_array())
        assert np.array_equal(ids, ids2) and np.array_equal(np.sort(ids), ref["node_id"])
        assert np.array_equal(f.read_clusters(), f2.read_clusters())
        m1 = f.read_cache()[2]; m2 = f2.read_cache()[2]
        assert np.array_equal(m1, m2) and np.array_equal(m1[np.argsort(ids)], ref["merged"])
