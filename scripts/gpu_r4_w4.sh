#!/bin/bash
# round 4: the Hessian sweep as FOUR-wave workgroups, two per CU (K3_BLOCK_V=256) against the tree's eight-wave one: parity, then same-box A/B at cfg2 / cfg3 / cfg4
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
VXBA_LIB=$PWD/gpurun_ab/libvxba_w4.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_w4.so" ROUNDS=2 STEPS=300 bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_w4.so" ROUNDS=2 STEPS=200 BENCH_ARGS="--config cfg3" bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_w4.so" ROUNDS=2 STEPS=67 BENCH_ARGS="--config cfg4" bash scripts/gpu_abn.sh
