#!/bin/bash
# round 6, step 2: timeline of the fused launch (s_memtime stamps of the instrumented build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/dbg_timeline.py k23 > gpurun_out/r6_s2_timeline_k23.txt 2>&1; echo rc=$?; grep -v "amdgpu.ids" gpurun_out/r6_s2_timeline_k23.txt | tail -40
