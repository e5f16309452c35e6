#!/bin/bash
# rocprofv3 kernel traces of the two secondary paths: the LiDAR-inertial shell (scripts/dbg_li_rate.py) and a wide top-level window
# (scripts/dbg_wide_timing.py).  Output: gpurun_out/prof_li, gpurun_out/prof_wide (+ the scripts' own timing lines in the .log files).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_li" -o t -- python $R/scripts/dbg_li_rate.py > "$R/gpurun_out/prof_li.log" 2>&1; echo "li trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_wide" -o t -- python $R/scripts/dbg_wide_timing.py > "$R/gpurun_out/prof_wide.log" 2>&1; echo "wide trace rc=$?"
cd "$R"; grep -v amdgpu.ids gpurun_out/prof_li.log | tail -3; grep -v amdgpu.ids gpurun_out/prof_wide.log | tail -12
VXBA_LI_TIMING=1 timeout 120 python scripts/dbg_li_rate.py 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/li_rate.txt; cat gpurun_out/li_rate.txt
