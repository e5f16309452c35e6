"""Turn the rocprofv3 output of scripts/gpu_profile.sh (under gpurun_out/) into a committed profiles/<name>/ directory:
kernel_stats.csv (the --kernel-trace --stats summary), pmc_hbm_counters.json (FETCH_SIZE / WRITE_SIZE mean per launch and
kernel, KB as rocprofv3 reports them) and bench.json (the bench line of the same build, if present)."""
import csv, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
bench = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "bench_full.json")
out = os.path.join(ROOT, "profiles", name)
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(ROOT, "gpurun_out", "prof_trace", "t_kernel_stats.csv"), os.path.join(out, "kernel_stats.csv"))
res = {}
for tag, counter in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
    path = os.path.join(ROOT, "gpurun_out", tag, "t_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    res[f"{counter}_KB_mean_per_launch"] = {k: {"n": len(v), "mean": sum(v) / len(v)} for k, v in acc.items() if k.startswith("void vxk") or k.startswith("vxk")}
if res:
    sys.path.insert(0, ROOT)
    import bench as bench_mod
    res["kernel_source_sha256"] = bench_mod.kernel_source_hash()      # bench.py refuses the file for any other kernel source
    res["kernel_sources"] = list(bench_mod.KERNEL_SOURCES)
    json.dump(res, open(os.path.join(out, "pmc_hbm_counters.json"), "w"), indent=1)
if os.path.exists(bench):
    shutil.copy(bench, os.path.join(out, "bench.json"))
for r in list(csv.DictReader(open(os.path.join(out, "kernel_stats.csv"))))[:8]:
    print("%-62s calls %5s avg %8.2f us %5s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
print(json.dumps({k: {kk: round(vv["mean"], 1) for kk, vv in v.items() if "k3_hessian" in kk or "k2_residual" in kk} for k, v in res.items() if isinstance(v, dict)}))
