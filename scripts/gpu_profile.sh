#!/bin/bash
# rocprofv3 passes for the bench command: kernel trace + stats (CSV), then HBM counters in their own passes.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-90} --warmup 9 --no-cpu-baseline --no-li-ba"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_trace" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/prof_trace.log" 2>&1; echo "trace rc=$?"
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log" 2>&1; echo "fetch rc=$?"
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_write" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/prof_write.log" 2>&1; echo "write rc=$?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof_* -type f | head -30; du -sh gpurun_out
