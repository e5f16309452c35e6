#!/bin/bash
# round 6, step 14: the device-resident scan cycle -- kernels (rocprofv3 --kernel-trace --stats) and host API calls (--hip-trace --stats)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
timeout 300 python scripts/dbg_map_cycle.py 2>&1 | grep -v amdgpu.ids | tail -12
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_map" -o t -- python $R/scripts/dbg_map_cycle.py > "$R/gpurun_out/prof_map.log" 2>&1; echo "map trace rc=$?"
timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d "$R/gpurun_out/prof_map_hip" -o t -- python $R/scripts/dbg_map_cycle.py > "$R/gpurun_out/prof_map_hip.log" 2>&1; echo "map hip rc=$?"
cd "$R"; find gpurun_out/prof_map gpurun_out/prof_map_hip -type f -name "*_trace.csv" -size +4M -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_map/t_kernel_stats.csv')))
for r in rows[:34]: print("%-70s calls %5s avg %9.2f us total %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
rows=list(csv.DictReader(open('gpurun_out/prof_map_hip/t_hip_api_stats.csv')))
for r in rows[:16]: print("%-40s calls %6s avg %9.2f us total %9.1f ms" % (r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
