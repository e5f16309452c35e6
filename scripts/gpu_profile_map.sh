#!/bin/bash
# rocprofv3 kernel trace of the device-resident scan cycle (scripts/dbg_map_cycle.py) -> gpurun_out/prof_map
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
timeout 300 python scripts/dbg_map_cycle.py 2>&1 | grep -v amdgpu.ids | tail -10
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_map" -o t -- python $R/scripts/dbg_map_cycle.py > "$R/gpurun_out/prof_map.log" 2>&1; echo "map trace rc=$?"
cd "$R"; python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_map/t_kernel_stats.csv')))
for r in rows[:40]: print("%-78s calls %5s avg %9.2f us total %9.1f us" % (r["Name"][:78], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
