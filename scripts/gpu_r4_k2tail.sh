#!/bin/bash
# round 4: the residual sweep's tail (coe requested with the other loads, the partial out before the cache stores) against the previous build: parity, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_li_ba.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -2
LIBS="gpurun_ab/libvxba_prev.so voxel-slam_amd/csrc/libvxba.so" ROUNDS=3 STEPS=300 bash scripts/gpu_abn.sh
LIBS="gpurun_ab/libvxba_prev.so voxel-slam_amd/csrc/libvxba.so" ROUNDS=2 STEPS=200 BENCH_ARGS="--config cfg3" bash scripts/gpu_abn.sh
LIBS="gpurun_ab/libvxba_prev.so voxel-slam_amd/csrc/libvxba.so" ROUNDS=2 STEPS=67 BENCH_ARGS="--config cfg4" bash scripts/gpu_abn.sh
