#!/bin/bash
# round 4: the ragged step of the Hessian sweep taken in the fill (-DK3_RAGGED_FIRST=1) against the tree: parity, then same-box A/B at cfg2 / cfg3 / cfg4
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
V=${VARIANT:-gpurun_ab/libvxba_ragf.so}
VXBA_LIB=$PWD/$V timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
LIBS="voxel-slam_amd/csrc/libvxba.so $V" ROUNDS=3 STEPS=300 bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so $V" ROUNDS=2 STEPS=200 BENCH_ARGS="--config cfg3" bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so $V" ROUNDS=2 STEPS=67 BENCH_ARGS="--config cfg4" bash scripts/gpu_abn.sh
