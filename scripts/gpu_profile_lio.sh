#!/bin/bash
# rocprofv3 passes for the odometry path (scripts/dbg_lio_timing.py): kernel trace + stats, then HBM counters in their own passes.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python $GRAFT_REPO_ROOT/scripts/dbg_lio_timing.py"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/lio_trace" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/lio_trace.log" 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/lio_fetch" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/lio_fetch.log" 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/lio_write" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/lio_write.log" 2>&1; echo "write rc=$?"
grep -h "sweep:\|state_estimation:" "$GRAFT_REPO_ROOT/gpurun_out/lio_trace.log"
