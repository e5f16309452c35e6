#!/bin/bash
# kernel-trace stats of the bench on the single-rank RCCL path (--force-dist): what does the collective add?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/prof_dist
export HSA_ENABLE_IPC_MODE_LEGACY=0
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_dist" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --force-dist --steps ${STEPS:-90} --warmup 9 --no-cpu-baseline --no-li-ba > "$GRAFT_REPO_ROOT/gpurun_out/prof_dist.log" 2>&1 )
grep metric gpurun_out/prof_dist.log | cut -c1-200
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_dist/t_kernel_stats.csv")))
for r in rows[:12]:
    print("%-60s calls %5s avg %9.2f us  total %8.2f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
tr=list(csv.DictReader(open("gpurun_out/prof_dist/t_kernel_trace.csv")))
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
# one steady-state step: find a k3_hessian in the middle and print the following 8 kernels with gaps
idx=[i for i,r in enumerate(tr) if "k3_hessian" in r["Kernel_Name"]]
i0=idx[len(idx)//2]
prev_end=None
for r in tr[i0:i0+9]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-50s dur %7.2f us  gap before %7.2f us" % (r["Kernel_Name"][:50], (e-s)/1e3, 0 if prev_end is None else (s-prev_end)/1e3))
    prev_end=e
PY
