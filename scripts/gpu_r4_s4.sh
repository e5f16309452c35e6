#!/bin/bash
# round 4, session 4: whole GPU suite on the tree (requests interleaved with phase M as the default), all eight requests behind the first K-step (late8)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -6
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_late8.so" ROUNDS=2 STEPS=300 bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_late8.so" ROUNDS=2 STEPS=100 BENCH_ARGS="--config cfg4" bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_late8.so" ROUNDS=2 STEPS=200 BENCH_ARGS="--config cfg3" bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_base.so" ROUNDS=1 STEPS=200 BENCH_ARGS="--config cfg3 --precision mixed" bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_base.so" ROUNDS=1 STEPS=300 BENCH_ARGS="--config cfg1" bash scripts/gpu_abn.sh
