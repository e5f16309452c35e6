"""File the rocprofv3 output of scripts/gpu_profile_cfg.sh (gpurun_out/prof_<tag>_{trace,fetch,write}/) under profiles/<name>/:
kernel_stats.csv (the --kernel-trace --stats summary), pmc_hbm_counters.json (FETCH_SIZE / WRITE_SIZE mean per launch and kernel, KB as
rocprofv3 reports them; stamped with the hash of the kernel sources) and roofline.json: for the two sweeps, algorithmic bytes,
average duration, achieved GB/s against 8 TB/s, counter traffic = FETCH_SIZE x 2 (the guide's gfx950 correction) + WRITE_SIZE.
    python scripts/collect_profile_cfg.py r04_cfg4 cfg4 [tag]"""
import csv, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name, cfg = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else cfg
out = os.path.join(ROOT, "profiles", name)
os.makedirs(out, exist_ok=True)
stats = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_trace", "t_kernel_stats.csv")
shutil.copy(stats, os.path.join(out, "kernel_stats.csv"))
res = {}
for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    path = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{kind}", "t_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = defaultdict(lambda: [0.0, 0])      # per kernel: sum over launches, launches (a table reduced by scripts/reduce_counters.py carries means + counts)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            n = int(r["Launches"]) if r.get("Launches") else 1
            a = acc[r["Kernel_Name"].split("(")[0]]
            a[0] += float(r["Counter_Value"]) * n; a[1] += n
    res[f"{counter}_KB_mean_per_launch"] = {k: {"n": v[1], "mean": v[0] / v[1]} for k, v in acc.items() if "vxk" in k}
import bench as bench_mod
from voxel_slam_amd import synth
res["kernel_source_sha256"] = bench_mod.kernel_source_hash()
res["kernel_sources"] = list(bench_mod.KERNEL_SOURCES)
res["config"] = cfg
json.dump(res, open(os.path.join(out, "pmc_hbm_counters.json"), "w"), indent=1)
if cfg not in synth.CONFIGS:
    # cfg5 (hierarchical pass): no closed-form bytes per launch (606 cluster builds of different sizes per pass; bench.py measures the algorithmic
    # bytes live) -- file the counter traffic of the pass's own kernels beside their trace durations
    roof = {"config": cfg, "hbm_peak_GBs": 8000.0, "kernels": []}
    F, Wr = res.get("FETCH_SIZE_KB_mean_per_launch", {}), res.get("WRITE_SIZE_KB_mean_per_launch", {})
    for r in csv.DictReader(open(stats)):
        k = r["Name"].split("(")[0]
        if k in F and k in Wr:
            tb = (2.0 * F[k]["mean"] + Wr[k]["mean"]) * 1024.0
            roof["kernels"].append({"kernel": k, "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "share_of_kernel_time_pct": float(r["Percentage"]),
                                    "traffic_bytes_per_launch": tb, "traffic_GBs": tb / float(r["AverageNs"]), "traffic_frac_of_hbm_peak": tb / float(r["AverageNs"]) / 8000.0})
    json.dump(roof, open(os.path.join(out, "roofline.json"), "w"), indent=1)
    for e in roof["kernels"][:8]:
        print("%-60s calls %6d avg %8.2f us  traffic %8.1f KB/launch = %6.0f GB/s" % (e["kernel"][:60], e["calls"], e["avg_us"], e["traffic_bytes_per_launch"] / 1024, e["traffic_GBs"]))
    sys.exit(0)
c = synth.CONFIGS[cfg]
V, W = c["n_voxels"], c["win_size"]
nnz = V * W if c.get("p_obs", 1.0) == 1.0 else None
roof = {"config": cfg, "voxels": V, "win_size": W, "hbm_peak_GBs": 8000.0, "infinity_cache_bytes": 256 * 2**20}
if nnz:
    alg = {"k3_hessian_kernel": 80.0 * nnz + 136.0 * V, "k2_residual_kernel": 80.0 * nnz + 88.0 * V + 176.0 * V}
    alg["k23_fused_kernel"] = alg["k3_hessian_kernel"] + alg["k2_residual_kernel"]     # round 6: solve | residual sweep | Hessian sweep in one launch -- both sweeps' bytes
    roof["working_set_bytes_per_step"] = alg["k3_hessian_kernel"] + alg["k2_residual_kernel"]
    roof["fits_infinity_cache"] = roof["working_set_bytes_per_step"] < roof["infinity_cache_bytes"]
    for r in csv.DictReader(open(stats)):
        for key, ab in alg.items():
            if key in r["Name"] and f"<{W}," in r["Name"].replace(" ", ""):
                avg_ns = float(r["AverageNs"])
                e = {"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_us": avg_ns / 1e3, "algorithmic_bytes": ab,
                     "achieved_GBs": ab / avg_ns, "frac_of_hbm_peak": ab / avg_ns / 8000.0}
                f = [v["mean"] for k, v in res.get("FETCH_SIZE_KB_mean_per_launch", {}).items() if key in k and f"<{W}," in k.replace(" ", "")]
                w = [v["mean"] for k, v in res.get("WRITE_SIZE_KB_mean_per_launch", {}).items() if key in k and f"<{W}," in k.replace(" ", "")]
                if f and w:
                    e["traffic_bytes"] = (2.0 * f[0] + w[0]) * 1024.0
                    e["traffic_over_algorithmic"] = e["traffic_bytes"] / ab
                roof.setdefault("kernels", []).append(e)
json.dump(roof, open(os.path.join(out, "roofline.json"), "w"), indent=1)
for r in list(csv.DictReader(open(stats)))[:8]:
    print("%-70s calls %5s avg %8.2f us %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
print(json.dumps(roof.get("kernels", []), indent=1))
