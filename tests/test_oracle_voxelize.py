"""Pins the oracle's batch factor construction (oracle/vxo_voxelize.hpp) against an independent numpy formulation:
layer-0 voxelisation by the reference's key rule + numpy.linalg.eigvalsh of the raw world points, and structural
invariants of the octree output (disjoint point ownership, children inside their parent cell)."""
import numpy as np

from voxel_slam_amd import synth, vxba
from tests import _oracle as O


def ref_keys(world, vs):
    loc = (world / vs).astype(np.float32)                 # float quotient (loop_refine.hpp:452-457)
    loc = np.where(loc < 0, loc - np.float32(1), loc)     # "-1 if negative"
    return loc.astype(np.int64)                           # truncation


def test_layer0_matches_numpy():
    W = 4
    xyz, fp, poses, _ = synth.make_scans(win_size=W, pts_per_scan=15_000, seed=synth.MASTER_SEED + 700)
    P = vxba.VoxelizeParams(voxel_size=1.0, max_layer=0, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16,) * 4)
    out = O.voxelize(W, xyz, fp, poses, P.as_array())
    Rs, ps = synth.unpack_poses(poses)
    frame = np.repeat(np.arange(W), np.diff(fp))
    world = np.einsum("nij,nj->ni", Rs[frame], xyz) + ps[frame]
    keys = ref_keys(world, 1.0)
    packed = ((keys[:, 0] + 32768) << 32) | ((keys[:, 1] + 32768) << 16) | (keys[:, 2] + 32768)
    expect = {}
    for k in np.unique(packed):
        sel = packed == k
        pts = world[sel]
        if pts.shape[0] <= 10:
            continue
        lam = np.linalg.eigvalsh(np.cov(pts.T, bias=True))
        margin = min(abs(lam[0] - 0.01), abs(lam[0] / lam[2] - 1 / 16), abs(lam[0] / lam[1] - 0.12))
        ok = lam[0] < 0.01 and lam[0] / lam[2] < 1 / 16 and np.unique(frame[sel]).size > 1 and lam[0] / lam[1] <= 0.12
        expect[int(k)] = (ok, margin, pts.shape[0], lam)
    got = {int(i >> np.uint64(16)): j for j, i in enumerate(out["node_id"])}
    assert np.all((out["node_id"] & np.uint64(7)) == 0)
    for k, (ok, margin, npts, lam) in expect.items():
        if margin < 1e-9:
            continue
        assert (k in got) == ok, (k, ok, lam)
        if ok:
            j = got[k]
            assert out["merged"][j, 9] == npts
            assert np.allclose(out["eig_val"][j], lam, rtol=0, atol=1e-12 * (np.abs(world).max() ** 2 + 1))
            assert out["clusters"][j, :, 9].sum() == npts
    assert set(got) <= {k for k, v in expect.items() if v[0] or v[1] < 1e-9}


def test_octree_output_structure():
    W = 5
    xyz, fp, poses, _ = synth.make_scans(win_size=W, pts_per_scan=30_000, seed=synth.MASTER_SEED + 701)
    P = vxba.VoxelizeParams(voxel_size=2.0, max_layer=3, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
    out = O.voxelize(W, xyz, fp, poses, P.as_array())
    ids = out["node_id"]
    layer = (ids & np.uint64(7)).astype(int)
    assert layer.max() >= 1 and np.all(np.diff(ids.astype(np.uint64)) > 0)          # canonical order, no duplicates
    root = ids >> np.uint64(16); path = ((ids >> np.uint64(7)) & np.uint64(511)).astype(int)
    # no factor is an ancestor of another one (a node is either a factor or subdivided, never both)
    seen = set()
    for r, p, l in zip(root.tolist(), path.tolist(), layer.tolist()):
        for la in range(l):
            anc = (r, p & ~((1 << (3 * (3 - la))) - 1) & 511, la)
            assert anc not in seen
        seen.add((r, p, l))
    # the world cluster of a factor lies inside its octree cell
    c = out["merged"][:, 6:9] / out["merged"][:, 9:10]
    size = 2.0 / (2.0 ** layer)
    xyz0 = np.stack([((root >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64) - 32768, ((root >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64) - 32768,
                     (root & np.uint64(0xffff)).astype(np.int64) - 32768], axis=1) * 2.0
    off = np.zeros((ids.size, 3))
    for la in range(1, 4):
        o = (path >> (3 * (3 - la))) & 7
        bits = np.stack([(o >> 2) & 1, (o >> 1) & 1, o & 1], axis=1)
        off += np.where((layer >= la)[:, None], bits * (2.0 / 2 ** la), 0.0)
    lo = xyz0 + off
    assert np.all(c >= lo - 1e-9) and np.all(c <= lo + size[:, None] + 1e-9)
    assert out["merged"][:, 9].sum() <= xyz.shape[0]                                  # every point owned by at most one factor
