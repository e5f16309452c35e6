#!/bin/bash
# round 4, session 12: K3 rows stored as they are finished (z row, G rows) instead of all at the end of phase A: parity, A/B, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_wide.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
LIBS="gpurun_ab/libvxba_prev.so voxel-slam_amd/csrc/libvxba.so" ROUNDS=3 STEPS=300 bash scripts/gpu_abn.sh
LIBS="gpurun_ab/libvxba_prev.so voxel-slam_amd/csrc/libvxba.so" ROUNDS=2 STEPS=100 BENCH_ARGS="--config cfg4" bash scripts/gpu_abn.sh
LIBS="gpurun_ab/libvxba_prev.so voxel-slam_amd/csrc/libvxba.so" ROUNDS=2 STEPS=200 BENCH_ARGS="--config cfg3" bash scripts/gpu_abn.sh
LIBS="gpurun_ab/libvxba_prev.so voxel-slam_amd/csrc/libvxba.so" ROUNDS=1 STEPS=200 BENCH_ARGS="--config cfg3 --precision mixed" bash scripts/gpu_abn.sh
VXBA_LIB=$PWD/voxel-slam_amd/csrc/libvxba.so timeout 200 python scripts/dbg_timeline.py k3 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_s12_timeline.txt; tail -14 gpurun_out/r4_s12_timeline.txt
