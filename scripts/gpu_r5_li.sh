#!/bin/bash
# round 5: where the LiDAR-inertial entry point's time goes on the closing tree (host split + the residual-sweep launch's stamps, LiDAR-only beside it)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_li
export HSA_ENABLE_IPC_MODE_LEGACY=0
VXBA_LI_TIMING=1 timeout 300 python scripts/dbg_li_rate.py > gpurun_out/r5_li/li_rate.txt 2>&1
grep -v "^\[vxba li queued\]" gpurun_out/r5_li/li_rate.txt | tail -4; grep "^\[vxba li queued\]" gpurun_out/r5_li/li_rate.txt | tail -3
timeout 300 python scripts/dbg_timeline.py fused > gpurun_out/r5_li/timeline_fused.txt 2>&1; head -12 gpurun_out/r5_li/timeline_fused.txt
timeout 300 python scripts/dbg_timeline.py fused_li > gpurun_out/r5_li/timeline_fused_li.txt 2>&1; head -12 gpurun_out/r5_li/timeline_fused_li.txt
