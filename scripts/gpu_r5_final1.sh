#!/bin/bash
# round 5, closing collection 1: rocprofv3 evidence from the final kernel sources (kernel trace + the two HBM counter passes at cfg2 / cfg3 / cfg4, the cold-L3
# rotation), the instrumented sweep's timeline inside the LM loop
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
CONFIG=cfg2 STEPS=90 bash scripts/gpu_profile_cfg.sh
CONFIG=cfg3 STEPS=60 bash scripts/gpu_profile_cfg.sh
CONFIG=cfg4 STEPS=30 bash scripts/gpu_profile_cfg.sh
TAG=cold CMD="python $GRAFT_REPO_ROOT/scripts/dbg_cold_l3.py cfg2" bash scripts/gpu_profile_cfg.sh
timeout 300 python scripts/dbg_timeline.py k3lm > gpurun_out/r5_final_timeline_k3lm.txt 2>&1; tail -5 gpurun_out/r5_final_timeline_k3lm.txt
du -sh gpurun_out
