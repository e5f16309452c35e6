#!/bin/bash
# round 6, closing collection 1: rocprofv3 evidence from the final kernel sources -- kernel trace + the two HBM counter passes at cfg2 / cfg3 / cfg4 / cfg5, the cold-L3
# rotation, the instrumented fused launch's timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
CONFIG=cfg2 STEPS=90 bash scripts/gpu_profile_cfg.sh
CONFIG=cfg3 STEPS=60 bash scripts/gpu_profile_cfg.sh
CONFIG=cfg4 STEPS=30 bash scripts/gpu_profile_cfg.sh
TAG=cfg5 CMD="python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline" bash scripts/gpu_profile_cfg.sh
TAG=cold CMD="python $GRAFT_REPO_ROOT/scripts/dbg_cold_l3.py cfg2" bash scripts/gpu_profile_cfg.sh
timeout 300 python scripts/dbg_timeline.py k23 > gpurun_out/r6_final_timeline_k23.txt 2>&1; tail -5 gpurun_out/r6_final_timeline_k23.txt
du -sh gpurun_out
