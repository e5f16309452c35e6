"""The multi-process path with two REAL ranks on one GPU (torch.distributed over gloo, exchange buffers reduced through host memory --
RCCL refuses two ranks on one device): each rank owns a voxel shard, and the sharded damping_iter / lm_steps must reproduce the
single-factor run on the whole window.  Complements the in-process two-shard emulation: here ranks are processes, the collective is
torch.distributed's, and a missing stream ordering or a stale buffer that is summed again shows (neither does with one rank)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("spec,collective", [("1", "hook"), ("0", "hook"), ("1", "peer"), ("0", "peer")])
def test_two_process_ranks_reproduce_the_single_factor_run(spec, collective):
    """collective = "peer": the one-shot all-reduce through hipIpc-mapped mailboxes (vxba_peer_*) -- two processes mapping each other's
    device memory and summing in rank order inside the device-resident loop; on an 8-GPU node the same reads cross xGMI."""
    env = dict(os.environ, VXBA_SPEC_COLLECTIVE=spec, VXBA_TWO_RANK_COLLECTIVE=collective, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29560 + int(spec) + (2 if collective == "peer" else 0)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "scripts", "dbg_two_rank.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    # the ranks print concurrently: records may share a line, so split on the record header rather than on newlines
    recs = re.split(r"(?=rank \d (?:damping_iter|lm_steps|lm_steps_easy|li_damping_iter|wide damping_iter):)", out.stdout)
    recs = [r for r in recs if re.match(r"rank \d (damping_iter|lm_steps|lm_steps_easy|li_damping_iter|wide damping_iter):", r)]
    assert len(recs) == 10, out.stdout[-2000:]
    for ln in recs:
        m = re.search(r"pose diff ([0-9.e+-]+) ([0-9.e+-]+)", ln)
        assert m and float(m.group(1)) < 1e-9 and float(m.group(2)) < 1e-9, ln
        if "damping_iter" in ln:
            a, b = re.search(r"trace accept (\[[^\]]*\]) vs (\[[^\]]*\])", ln).groups()
            assert a == b and "1." in a, ln                        # same schedule, with accepted steps in it ...
            if re.match(r"rank \d damping_iter:", ln):
                assert "0." in a, ln                               # ... and, on the hard window, rejected ones
        else:
            a, b = re.search(r"stats (\{[^}]*\}) vs (\{[^}]*\})", ln).groups()
            assert a == b, ln
            if "lm_steps_easy" in ln:
                assert "'rejected': 0" in a, ln


@pytest.mark.parametrize("world,K,wd,mg", [(2, 45, 6, 3), (3, 105, 10, 5)])
def test_hierarchical_ba_sharded_over_process_ranks(world, K, wd, mg):
    """BASELINE configs[4]'s any-N driver (dist.hba_pass: vxba_hba_bottom over the rank's windows | packed submaps all-gathered | vxba_hba_top) on real
    GPU code paths (ranks = processes on one GPU, gloo): the top-level window is filtered to the rank's root voxels ON THE DEVICE
    (vxba_voxelize_params.shard_*), held as a wide factor per rank (the session's top-level factor) and summed through the all-reduce hook.
    Against the one-rank pass (vxba_hba_pass): identical submaps, shards that partition every round's factor set, the same poses on all ranks
    bit for bit and within round-off of the one-rank ones.  (105 keyframes / 10 / 5: a 21-pose wide top level incl. the closing window.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HBA_K=str(K), HBA_WD=str(wd), HBA_MG=str(mg))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
                          str(29580 + world), os.path.join(ROOT, "scripts", "dbg_two_rank_hba.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    recs = [r for r in re.split(r"(?=rank \d hba_sharded:)", out.stdout) if r.startswith("rank ")]
    assert len(recs) == world, out.stdout[-2000:]
    for ln in recs:
        m = re.search(r"pose diff ([0-9.e+-]+) ([0-9.e+-]+), same bits on all ranks (\w+), top voxels per round (\[[^\]]*\]) sum (\[[^\]]*\]) vs (\[[^\]]*\]), submap sizes equal (\w+), edges (\d+) (\d+) vs (\d+) (\d+)", ln)
        assert m, ln
        assert float(m.group(1)) < 1e-8 and float(m.group(2)) < 1e-8, ln
        assert m.group(3) == "True" and m.group(7) == "True", ln
        assert m.group(5) == m.group(6), ln                      # the shards partition the factor set of every round
        assert (m.group(8), m.group(9)) == (m.group(10), m.group(11)), ln


def _bench_line(extra, timeout=900):
    """`python bench.py --gpus 2 ...` exactly as the driver would type it (no launcher: the script starts its own ranks), both ranks on device 0
    over gloo (the development switches of bench.py: RCCL refuses two ranks on one device)."""
    import json
    env = dict(os.environ, VXBA_BENCH_DEVICE="0", VXBA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_2_starts_its_own_ranks_and_reports_both_scalings_and_carriers():
    d = _bench_line(["--steps", "20", "--warmup", "5", "--repeats", "3", "--prewarm-seconds", "0.05"])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["value"] > 0
    assert d["config"]["ranks_seen"] == 2
    assert d["scaling"] == "strong" and abs(d["value"] - d["window_iterations_per_s"]) < 1e-9 * d["value"]
    assert abs(d["config"]["collectives_per_lm_step"] - 1) < 0.1      # one per LM iteration (+ the one that opens a call)
    # both scalings measured in the one job; the mailbox carrier ran on both, RCCL is reported as not available over gloo
    for scal, vox in (("strong", 25000), ("weak", 50000)):
        peer = d["carriers"][scal]["peer"]
        assert peer["available"] and peer["ranks_seen"] == 2 and peer["voxels_per_gpu"] == vox and peer["allreduce_us_avg"] > 0, peer
        assert peer["lm_steps_accepted"] == 20 and abs(peer["collectives_per_lm_step"] - 1) < 0.1
        assert d["carriers"][scal]["rccl"]["available"] is False
    w = d["weak_scaling"]
    assert abs(w["value"] - 2 * w["window_iterations_per_s"]) < 1e-9 * w["value"]
    # the two shards of the strong leg ARE the cfg2 window: same final residual as the weak leg's 2x larger window would not give
    assert d["carriers"]["strong"]["peer"]["final_residual"] != d["carriers"]["weak"]["peer"]["final_residual"]


def test_two_rank_cfg5_pass_costs_what_the_one_rank_pass_costs():
    """Round-5 review: the N > 1 pass ran another, slower implementation (a Python orchestration) than N = 1.  Now both are the same C-ABI calls.  Two
    process ranks SHARING one GPU: the bottom level -- the part the ranks split, half the windows each -- must cost what the one-rank bottom level
    costs (<= 1.15 x: the same kernels from two processes instead of one; on two GPUs it halves).  The rest of a two-rank pass is priced for this
    box only: the packed submaps and every all-reduce of the top level's 485 KB system cross HOST memory under gloo (RCCL refuses two ranks on one
    device), and both ranks voxelise the whole top level on the one GPU they share -- the whole pass stays below 3 x the one-rank pass even so
    (round 6: the one-rank pass of this session went from 0.09 to 0.044 s while the gloo exchange of the submaps through host memory stayed at
    0.04 s -- half of a two-rank pass on this box, none of it on two GPUs)."""
    # One stream per rank (as bench.py gives ranks that share a device: several streams in each of two processes oversubscribe the GPU's queues and
    # the driver time-slices whole processes, round 6).  The parity half of the test must hold on every attempt; the timing half is a measurement on a
    # box shared with other tenants and gets three attempts.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HBA_K="205", HBA_WD="10", HBA_MG="5", HBA_PTS="20000", HBA_THREADS="1")
    timings = []
    for attempt in range(3):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(29589 + attempt),
                              os.path.join(ROOT, "scripts", "dbg_two_rank_hba.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        recs = [r for r in re.split(r"(?=rank \d hba_sharded:)", out.stdout) if r.startswith("rank ")]
        assert len(recs) == 2, out.stdout[-2000:]
        ok = True
        for ln in recs:
            m = re.search(r"pose diff ([0-9.e+-]+) ([0-9.e+-]+), same bits on all ranks (\w+)", ln)
            assert m and float(m.group(1)) < 1e-9 and float(m.group(2)) < 1e-9 and m.group(3) == "True", ln
            ratio = float(re.search(r"\(ratio ([0-9.]+)\)", ln).group(1))
            bottom = float(re.search(r"\(bottom ratio ([0-9.]+)\)", ln).group(1))
            timings.append((bottom, ratio))
            ok &= bottom <= 1.15 and ratio <= 3.0
        if ok:
            return
    raise AssertionError("two ranks on one GPU: (bottom ratio, pass ratio) per rank and attempt %s -- wanted <= 1.15 and <= 3.0" % timings)


def test_bench_cfg5_gpus_2_starts_its_own_ranks():
    d = _bench_line(["--config", "cfg5", "--keyframes", "105", "--keyframe-points", "5000", "--steps", "1", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert "vxba_hba_bottom over windows rank, rank + 2" in d["config"]["parallelism"]
    assert d["roofline"] and d["roofline"]["frac"] > 0 and d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0          # the N > 1 line carries both, like N = 1
