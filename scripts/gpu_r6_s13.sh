#!/bin/bash
# round 6, step 13: the host side of a hierarchical pass -- HIP API calls by count and time (rocprofv3 --hip-trace --stats, no counters)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg5_hip -o t -- python $R/bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_cfg5_hip.log 2>&1; echo "rc=$?"
cd $R; find gpurun_out/prof_cfg5_hip -type f -name "*_trace.csv" -size +4M -delete; ls -la gpurun_out/prof_cfg5_hip | head; head -30 gpurun_out/prof_cfg5_hip/t_hip_api_stats.csv 2>/dev/null | cut -c1-150
