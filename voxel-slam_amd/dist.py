"""Voxel-sharded BA across the GPUs of one node: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPUs for tests).

The window's voxels are independent given the W poses (the reference already shards them over std::threads with private
accumulators, voxel_map.hpp:298-335), so every rank owns a contiguous voxel shard that never leaves its HBM; the only
exchange is one all-reduce(sum) of the packed ``[Hess (6W)^2 | JacT 6W | residual]`` buffer per Hessian sweep (29.3 KB for
W = 10) and of one scalar per residual sweep.  After the all-reduce every rank holds the same system and takes the same
LM decision, so no further synchronisation is needed.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_voxels: int, world: int, rank: int):
    """Contiguous voxel range of ``rank``: the reference's own truncation rule ``int(part*i) .. int(part*(i+1))``
    with ``part = n_voxels / world`` (voxel_map.hpp:318-321)."""
    part = float(n_voxels) / world
    return int(part * rank), int(part * (rank + 1))


def rank_scene(make_kwargs: dict, scaling: str, rank: int, world: int):
    """The voxel shard of ``rank`` for the two ways a window is spread over ``world`` GPUs (bench.py ``--scaling``):

    * ``"weak"``   -- per-GPU work fixed: every rank generates its OWN ``make_kwargs``-sized set of voxels (seed offset by the rank)
      of one shared window (same trajectory and initial guess: ``pose_seed``); the global window has ``world`` times the voxels
      (BASELINE configs[3] = 8 x configs[1]).
    * ``"strong"`` -- total work fixed: all ranks generate the SAME window and rank r keeps the contiguous voxel range the reference's
      own shard rule gives it (``shard_bounds``, voxel_map.hpp:318-321).

    Returns a ``synth.Scene`` holding only this rank's voxels (points re-bucketed per frame, clusters, fix clusters, weights)."""
    from . import synth
    kw = dict(make_kwargs)
    seed = kw.pop("seed")
    if scaling == "weak":
        return synth.make_scene(seed=seed + 1000 * rank, pose_seed=seed, **kw)
    if scaling != "strong":
        raise ValueError("scaling must be 'weak' or 'strong'")
    sc = synth.make_scene(seed=seed, pose_seed=seed, **kw)
    if world == 1:
        return sc
    lo, hi = shard_bounds(sc.n_voxels, world, rank)
    W, V = sc.win_size, sc.n_voxels
    pts, ptr = [], [0]
    for i in range(W):                               # cells are ordered frame-major: cell i * V + a
        a, b = int(sc.cell_ptr[i * V + lo]), int(sc.cell_ptr[i * V + hi])
        pts.append(sc.points_body[a:b])
        ptr.extend((sc.cell_ptr[i * V + lo + 1: i * V + hi + 1] - a + ptr[-1]).tolist())
    return synth.Scene(win_size=W, n_voxels=hi - lo, points_body=np.ascontiguousarray(np.concatenate(pts)), cell_ptr=np.asarray(ptr, dtype=np.int64),
                       clusters=sc.clusters[lo:hi], fix=sc.fix[lo:hi], coe=sc.coe[lo:hi], poses_gt=sc.poses_gt, poses_init=sc.poses_init,
                       normals=sc.normals[lo:hi])


def attach_allreduce(factor, group=None):
    """GPU path: make the factor's device-resident LM loop all-reduce its exchange buffers over ``group``.

    The buffers become torch tensors (so RCCL can reduce them in place) and the factor runs on torch's current stream,
    which orders the collective between the reduction kernel that fills the buffer and the solve kernel that reads it.
    Returns the tensors (keep them alive as long as the factor is used)."""
    import torch
    import torch.distributed as dist

    n = factor.packed_len()
    xbuf = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
    packed_t, scalar_t = xbuf[:n], xbuf[n:]
    # A dedicated stream shared by the factor's kernels and the collectives.  torch's default stream will not do: its handle is 0,
    # which vxba_set_stream reads as "back to the factor's own (non-blocking) stream" -- the collective would then run unordered
    # against the kernels that fill and read the exchange buffer (invisible with one rank, garbage with two).
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    factor.set_stream(stream.cuda_stream)
    factor.use_external_buffers(packed_t.data_ptr(), scalar_t.data_ptr())
    staged = dist.get_backend(group) == "gloo"   # plumbing runs of the multi-process path without RCCL: reduce through host memory

    def hook(_ptr, count, _stream):
        # count == 1: the residual scalar; count == n: the packed system; count == n + 1: both in one collective (the scalar
        # sits directly behind the packed buffer, which is what lets the device-resident loop merge them)
        t = scalar_t if count == 1 else xbuf[:count]
        with torch.cuda.stream(stream):
            if staged:
                h = t.cpu()                          # synchronises with the stream that filled the buffer
                dist.all_reduce(h, group=group)
                t.copy_(h)
            else:
                dist.all_reduce(t, group=group)

    factor.set_allreduce(hook)
    return xbuf, packed_t, scalar_t, stream


def attach_rccl(factor, group=None):
    """GPU path, no per-sweep host callback: give the factor its own RCCL communicator (created from the SAME librccl.so
    torch already loaded, so the process has one RCCL instance) and let the C++ loop call ncclAllReduce directly.
    The ncclUniqueId travels over the existing torch.distributed group.  Collective."""
    import os

    import torch
    import torch.distributed as dist

    from . import vxba

    # inside a torch process the communicator comes from the librccl.so torch already loaded (one RCCL instance per process); a
    # caller without torch passes NULL / "/opt/rocm/lib/librccl.so" to vxba_rccl_attach[_bcast] and exchanges the id itself
    lib = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    obj = [vxba.rccl_unique_id(lib) if rank == 0 else None]
    dist.broadcast_object_list(obj, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    factor.rccl_attach(lib, world, rank, obj[0])   # collectives are issued on the factor's own stream, in order with its kernels


def attach_peer(factor, group=None):
    """One-shot all-reduce over the ranks' mailboxes (vxba_peer_*): point-to-point xGMI reads instead of a ring collective for the 29 KB
    exchange buffer.  The 64-byte IPC handles travel over the existing torch.distributed group (any backend).  Collective.  Raises
    VxbaError where IPC / peer access is not available -- fall back to attach_rccl then."""
    import torch.distributed as dist

    import torch

    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def all_ok(flag: bool) -> bool:
        # every rank learns whether ANY rank failed, so that all of them raise (or fall back) together instead of one raising while
        # the others block in the next collective
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(int(t.item()))

    err = None
    try:
        mine = factor.peer_export()
    except Exception as e:          # no fine-grained memory / IPC disabled
        err, mine = e, None
    handles = [None] * world
    dist.all_gather_object(handles, mine, group=group)
    if err is None and any(h is None for h in handles):
        err = RuntimeError("a peer could not export its mailbox")
    if err is None:
        try:
            factor.peer_attach(world, rank, handles)
        except Exception as e:
            err = e
    if not all_ok(err is None):       # also the barrier: nobody starts a collective before every rank has mapped every mailbox
        if err is None:
            factor.peer_detach()
        raise RuntimeError("peer all-reduce: attach failed on some rank (rank %d: %s)" % (rank, err or "ok here"))
    ok = False
    try:
        ok = factor.peer_selftest()   # one all-reduce of a known pattern: a link that does not sum exactly is not used
    except Exception as e:
        err = e
    if not all_ok(ok):
        factor.peer_detach()
        raise RuntimeError("peer all-reduce self-test failed on some rank (rank %d: %s)" % (rank, err or ("ok here" if ok else "wrong sum / peer absent")))


def damping_iter_sharded(win_size: int, x_stats, local_hess, local_resid, max_iter: int = 3, group=None):
    """``Lidar_BA_Optimizer::damping_iter`` over a voxel shard per rank, host-driven.

    ``local_hess(xs) -> (Hess, JacT, residual)`` and ``local_resid(xs) -> residual`` evaluate THIS rank's shard; their
    results are summed across ``group`` with ``all_reduce`` before the (replicated) LM step.  Backend-agnostic: works
    with gloo on CPU tensors, which is how the N > 1 logic is tested without GPUs."""
    import torch
    import torch.distributed as dist

    from . import vxba

    n = 6 * win_size

    def hess_fn(xs):
        H, J, r = local_hess(xs)
        buf = torch.from_numpy(np.concatenate([np.asarray(H, dtype=np.float64).reshape(-1), np.asarray(J, dtype=np.float64), [r]]))
        dist.all_reduce(buf, group=group)
        out = buf.numpy()
        return out[: n * n].reshape(n, n), out[n * n: n * n + n], float(out[n * n + n])

    def resid_fn(xs):
        buf = torch.tensor([float(local_resid(xs))], dtype=torch.float64)
        dist.all_reduce(buf, group=group)
        return float(buf[0])

    return vxba.damping_iter_generic(win_size, x_stats, hess_fn, resid_fn, max_iter=max_iter)


# ---------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: the hierarchical global BA over the GPUs of one node
# ---------------------------------------------------------------------------------------------------------------------------------
def root_shard(root48, count: int):
    """Shard of a root voxel ``[x:16 | y:16 | z:16]`` (coordinates offset by 32768): the numpy twin of ``vxv::root_shard``
    (csrc/vxba_voxelize.h) -- Fibonacci hash of the 48-bit root modulo ``count``.  Accepts node ids' upper 48 bits (``id >> 16``)."""
    r = np.asarray(root48, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = (r * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)
    return (h % np.uint64(count)).astype(np.int64)


def _gather_arrays(arrs, group=None):
    """all_gather of a list of variable-length float32 (n_i, 3) arrays per rank -> list over ranks of lists; any backend (tensors on the
    GPU for RCCL, on the host for gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    sizes = [None] * world
    dist.all_gather_object(sizes, [int(a.shape[0]) for a in arrs], group=group)
    mine = np.concatenate(arrs).astype(np.float32) if arrs else np.zeros((0, 3), np.float32)
    cap = max(1, max(sum(s) for s in sizes))
    buf = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
    buf[: mine.shape[0]] = torch.from_numpy(np.ascontiguousarray(mine)).to(dev)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    res = []
    for r in range(world):
        flat = outs[r].cpu().numpy()
        o, parts = 0, []
        for n in sizes[r]:
            parts.append(flat[o:o + n].copy()); o += n
        res.append(parts)
    return res


def hba_ctx_close(ctx: dict):
    """Release what ``hierarchical_ba_sharded`` keeps in ``ctx`` (factors first, then the tensors / stream their collective hook uses)."""
    for k in [k for k in ctx if k[0] in ("bottom", "top")]:
        ctx.pop(k).close()
    ctx.clear()


def hierarchical_ba_sharded(clouds, poses, coarse, fine, wdsize: int = 10, mgsize: int = 5, top_max_iter: int = 1, device: int = 0, group=None,
                            bottom_refine=None, top_refine=None, downsample=None, ctx: dict | None = None, tail: bool = True):
    """``hba.hierarchical_ba`` (thd_globalmapping's bottom-up pass, voxelslam.cpp:2485-2595) over the ranks of ``group``, one process per GPU:

    * **bottom level** -- the windows of ``wdsize`` keyframes (stride ``mgsize``; 99 of them at BASELINE configs[4]) are independent
      ``HBA_add_edge`` problems: window w runs on rank ``w % world`` as a replica of the single-GPU path, no exchange inside it;
    * the merged, voxel-filtered submaps and the windows' pose-graph edges are all-gathered (every rank needs every submap next);
    * **top level** -- ONE window over all submap poses (W ~ 99: the wide-window path), voxel-sharded: every rank voxelises the same points
      but keeps the root voxels that hash to it (``VoxelizeParams.sharded``; filtered on the device before the sorts), holds that shard of
      factor voxels in its HBM, and the packed ``[Hess | JacT | residual]`` buffer ((6W)^2 + 6W + 1 doubles = 2.88 MB at W = 99) is summed
      by ONE all-reduce per sweep -- the "multi-level Hessian reduction": wave -> workgroup -> grid -> RCCL.  At this size RCCL picks
      reduce-scatter + all-gather over all seven xGMI links by itself; the factor calls ``ncclAllReduce`` directly (``attach_rccl``) or goes
      through ``attach_allreduce``'s hook.  After the all-reduce every rank holds the same system and takes the same LM decisions, so the
      refined submap poses are identical on all ranks, bit for bit.

    ``bottom_refine(xyz, fp, poses) -> dict(poses, hess, rounds)`` and ``top_refine(xyz, fp, poses, shard_index, shard_count) -> the same``
    replace the GPU path in the CPU tests (gloo): the oracle stands in for a rank's GPU.  ``ctx``: a dict that keeps the two factors (and the
    top-level factor's communicator) alive between calls -- a mapper that runs pass after pass creates them once; release with
    ``hba_ctx_close(ctx)``.  Returns what ``hba.hierarchical_ba`` returns, plus ``windows_of_rank``."""
    import torch.distributed as dist

    from . import hba, vxba

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    K = poses.shape[0]
    wins = hba.windows(K, wdsize, mgsize, tail)
    bases = [b for b, _ in wins]
    mine = [w for w in range(len(bases)) if w % world == rank]
    bottom = None
    own = ctx is None
    ctx = {} if ctx is None else ctx
    if bottom_refine is None:
        def bottom_refine(xyz, fp, xs):
            cnt = xs.shape[0]
            bottom = ctx.get(("bottom", cnt))     # `is None`, not truthiness: LidarFactor.__len__ is its voxel count, an empty cached factor is falsy
            if bottom is None:
                bottom = ctx[("bottom", cnt)] = vxba.LidarFactor(cnt, device=device)
            return hba.window_refine(xyz, fp, xs, coarse, fine, max_iter=1, device=device, factor=bottom)
    my_sub, my_edges = [], []
    for w in mine:
        ids = list(range(wins[w][0], wins[w][0] + wins[w][1]))
        if len(ids) >= 2:
            xyz = np.ascontiguousarray(np.concatenate([np.asarray(clouds[i], dtype=np.float64) for i in ids]))
            fp = np.concatenate([[0], np.cumsum([len(clouds[i]) for i in ids])]).astype(np.int64)
            r = bottom_refine(xyz, fp, poses[ids])
            my_edges.append([dict(e, i=ids[e["i"]], j=ids[e["j"]]) for e in hba.edges_from_hessian(r["poses"], r["hess"])])
            refined = r["poses"]
        else:                                     # a closing window of one keyframe: nothing to refine, no pair for an edge
            my_edges.append([])
            refined = poses[ids]
        my_sub.append(hba.merge_submap([clouds[i] for i in ids], refined, fine.voxel_size, downsample=downsample, device=device))
    # every rank needs every submap for the top level; the edges are small
    all_sub = _gather_arrays(my_sub, group)
    all_edges = [None] * world
    dist.all_gather_object(all_edges, my_edges, group=group)
    sub_clouds, edges1 = [], []
    for w in range(len(bases)):
        r, k = w % world, w // world
        sub_clouds.append(all_sub[r][k])
        edges1.extend(all_edges[r][k])
    S = len(bases)
    top_xyz = np.ascontiguousarray(np.concatenate(sub_clouds).astype(np.float64))
    top_fp = np.concatenate([[0], np.cumsum([len(c) for c in sub_clouds])]).astype(np.int64)
    sub_ids = bases
    # A SHORT session's top level is a narrow window (S <= VXBA_MAX_WIN): the device-resident narrow loop refuses an empty factor and skips the
    # collective with it, so a rank whose hashed shard came out empty would fail while the others wait in the all-reduce (round-4 advisor).
    # There is nothing to gain from sharding ten submap poses anyway: every rank runs that top level whole -- same inputs, same arithmetic,
    # same poses on all ranks, no collective.
    shard_top = world > 1 and S > vxba.MAX_WIN
    if top_refine is None:
        topf = ctx.get(("top", S))
        if topf is None:
            topf = ctx[("top", S)] = vxba.LidarFactor(S, device=device)
        if shard_top and ("keep", S) not in ctx:
            keep = None
            try:
                attach_rccl(topf, group) if dist.get_backend(group) == "nccl" else None
                ok = dist.get_backend(group) == "nccl"
            except Exception:      # noqa: BLE001 -- fall back together below
                ok = False
            import torch
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            if not int(t.item()):
                if ok:
                    topf.rccl_detach()
                keep = attach_allreduce(topf, group)
            ctx[("keep", S)] = keep

        def top_refine(xyz, fp, xs, si, sc):
            return hba.window_refine(xyz, fp, xs, coarse.sharded(si, sc), fine.sharded(si, sc), max_iter=top_max_iter, device=device, factor=topf)
    top = top_refine(top_xyz, top_fp, poses[sub_ids], rank if shard_top else 0, world if shard_top else 1)
    if own:
        hba_ctx_close(ctx)
    edges2 = [dict(e, i=sub_ids[e["i"]], j=sub_ids[e["j"]]) for e in hba.edges_from_hessian(top["poses"], top["hess"])]
    return dict(edges1=edges1, edges2=edges2, submap_ids=sub_ids, submap_poses=top["poses"], submap_sizes=[len(c) for c in sub_clouds], top_rounds=top["rounds"],
                windows_of_rank=mine)


def hba_pass(ses: "vxba.HbaSession", poses, coarse, fine, wdsize: int = 10, mgsize: int = 5, tail: bool = True, top_max_iter: int = 1, group=None,
             n_threads: int = 0, ctx: dict | None = None):
    """ONE code path for a bottom-up pass on any number of ranks (BASELINE configs[4]; round 6): the session's keyframes are resident on every
    rank's GPU (``HbaSession.add_keyframes``), and a pass is

    * ``vxba_hba_bottom`` over the windows ``rank, rank + world, ..`` -- below the C ABI, four polled streams per rank, submaps left in HBM;
    * the submaps exchanged DEVICE TO DEVICE: each rank packs its windows' submaps (``vxba_hba_export_submaps``) into a tensor, one
      ``all_gather`` (RCCL over xGMI under the ``nccl`` backend; host tensors under ``gloo``), the peers' submaps unpacked into place
      (``vxba_hba_import_submaps``); the sizes and the windows' edges travel as Python objects (a few KB);
    * ``vxba_hba_top`` on every rank: a wide top level (S > VXBA_MAX_WIN) voxel-sharded -- every rank voxelises the same submaps and keeps the
      root voxels that hash to it, the packed [Hess | JacT | residual] summed by the collective attached to the session's top-level factor
      (``ncclAllReduce`` inside the factor, else the ``torch.distributed`` hook) -- a narrow one whole on every rank (same inputs, same bits).

    ``world == 1`` (``group`` None and no process group initialised, or a group of one) runs the same three calls without the exchange.
    ``ctx``: keeps the collective's attachment alive between passes.  Returns what ``HbaSession.run_pass`` returns, plus ``windows_of_rank``."""
    import torch
    import torch.distributed as dist

    from . import vxba

    have_pg = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if have_pg else (0, 1)
    import time
    ctx = {} if ctx is None else ctx
    t_0 = time.perf_counter()
    bot = ses.bottom(poses, coarse, fine, wdsize, mgsize, tail, w_first=rank, w_stride=world, n_threads=n_threads)
    t_1 = time.perf_counter()
    wins, sizes = bot["windows"], bot["sizes"]
    S = len(wins)
    edges1 = bot["edges"]
    if world > 1:
        on_gpu = dist.get_backend(group) == "nccl"
        all_sizes = [None] * world
        dist.all_gather_object(all_sizes, [int(x) for x in sizes], group=group)
        merged = np.array([max(col) for col in zip(*all_sizes)], dtype=np.int64)            # every window was run by exactly one rank (-1 elsewhere)
        per_rank = [int(sum(merged[r::world])) for r in range(world)]
        cap = max(1, max(per_rank))
        key = ("xchg", cap, world, on_gpu)
        if key not in ctx:
            ctx[key] = (torch.zeros((cap, 3), dtype=torch.float32, device="cuda"), [torch.zeros((cap, 3), dtype=torch.float32, device="cuda" if on_gpu else "cpu") for _ in range(world)])
        mine_t, outs = ctx[key]
        n = ses.export_submaps(rank, world, mine_t.data_ptr(), cap)
        assert n == per_rank[rank], (n, per_rank)
        torch.cuda.synchronize()
        dist.all_gather(outs, mine_t if on_gpu else mine_t.cpu(), group=group)
        for r in range(world):
            if r != rank:
                peer = outs[r] if on_gpu else outs[r].cuda()
                ses.import_submaps(r, world, merged, peer.data_ptr())
        torch.cuda.synchronize()
        sizes = merged
        all_edges = [None] * world
        dist.all_gather_object(all_edges, edges1, group=group)
        edges1 = sorted((e for es in all_edges for e in es), key=lambda e: e["window"])       # stable: window order, each window's edges in its own order
    t_2 = time.perf_counter()
    shard_top = world > 1 and S > vxba.MAX_WIN
    if shard_top and ("topf", S) not in ctx:
        topf = ses.top_factor()
        ok = False
        keep = None
        if dist.get_backend(group) == "nccl":
            try:
                attach_rccl(topf, group)
                ok = True
            except Exception:      # noqa: BLE001 -- fall back together below
                ok = False
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        if not int(t.item()):
            if ok:
                topf.rccl_detach()
            keep = attach_allreduce(topf, group)
        ctx[("topf", S)] = (topf, keep)
    c, f = (coarse.sharded(rank, world), fine.sharded(rank, world)) if shard_top else (coarse, fine)
    t_3 = time.perf_counter()
    top = ses.top(poses, c, f, top_max_iter)
    t_4 = time.perf_counter()
    for e in edges1:
        e.pop("window", None)
    return dict(edges1=edges1, edges2=top["edges2"], submap_ids=[b for b, _ in wins], submap_poses=top["submap_poses"], submap_sizes=[int(x) for x in sizes],
                top_rounds=top["top_rounds"], windows_of_rank=list(range(rank, S, world)), n_threads_used=bot["n_threads_used"],
                phase_s=dict(bottom=t_1 - t_0, exchange=t_2 - t_1, attach=t_3 - t_2, top=t_4 - t_3))
