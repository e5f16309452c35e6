#!/bin/bash
# round 4, session 5: rocprofv3 (kernel trace + the two HBM counter passes) at cfg2, cfg3, cfg4 and of the cold-L3 rotation; mixed precision A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_late4mixed.so gpurun_ab/libvxba_base.so" ROUNDS=2 STEPS=200 BENCH_ARGS="--config cfg3 --precision mixed" bash scripts/gpu_abn.sh
CONFIG=cfg2 STEPS=90 bash scripts/gpu_profile_cfg.sh
CONFIG=cfg3 STEPS=60 bash scripts/gpu_profile_cfg.sh
CONFIG=cfg4 STEPS=30 bash scripts/gpu_profile_cfg.sh
TAG=cold CMD="python $GRAFT_REPO_ROOT/scripts/dbg_cold_l3.py cfg2" bash scripts/gpu_profile_cfg.sh
tail -2 gpurun_out/prof_cold_trace.log
du -sh gpurun_out
