#!/bin/bash
# kernel-trace stats of the bench (CSV) -> prints the per-kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/prof_trace
export HSA_ENABLE_IPC_MODE_LEGACY=0
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_trace" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps ${STEPS:-90} --warmup 9 --no-cpu-baseline --no-li-ba > "$GRAFT_REPO_ROOT/gpurun_out/prof_trace.log" 2>&1 )
grep -v amdgpu.ids gpurun_out/prof_trace.log | tail -1 | cut -c1-400
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_trace/t_kernel_stats.csv")))
for r in rows[:14]:
    print("%-60s calls %5s avg %9.2f us  total %8.2f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
