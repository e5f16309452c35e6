#!/bin/bash
# rocprofv3 passes of the cfg3 bench in VXBA_PRECISION_MIXED_F32_CLUSTERS: kernel trace + stats, then FETCH_SIZE / WRITE_SIZE in their own passes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python $GRAFT_REPO_ROOT/bench.py --config cfg3 --precision ${PRECISION:-mixed_f32_clusters} --steps 90 --warmup 9 --no-cpu-baseline --no-li-ba"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/f32_trace" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/f32_trace.log" 2>&1; echo "trace rc=$?"
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/f32_fetch" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/f32_fetch.log" 2>&1; echo "fetch rc=$?"
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/f32_write" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/f32_write.log" 2>&1; echo "write rc=$?"
cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/f32_*/t_kernel_trace.csv     # large, not needed: the stats and the counter files are
python3 - <<'PY'
import csv
from collections import defaultdict
for r in list(csv.DictReader(open("gpurun_out/f32_trace/t_kernel_stats.csv")))[:4]:
    print("%-70s calls %5s avg %8.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
for tag, counter in (("f32_fetch", "FETCH_SIZE"), ("f32_write", "WRITE_SIZE")):
    acc = defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/%s/t_counter_collection.csv" % tag)):
        if r["Counter_Name"] == counter and ("k2_residual" in r["Kernel_Name"] or "k3_hessian" in r["Kernel_Name"]):
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%s %-60s n %5d mean %.1f KB" % (counter, k[:60], len(v), sum(v) / len(v)))
PY
