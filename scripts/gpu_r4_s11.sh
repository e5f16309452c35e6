#!/bin/bash
# round 4, session 11: K3 with pair-local flags instead of the step barrier (review's experiment (a)), operands two K-steps ahead ((b)): parity of the pair build, same-box A/B, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
VXBA_LIB=$PWD/gpurun_ab/libvxba_pair.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_pair.so gpurun_ab/libvxba_pf2.so gpurun_ab/libvxba_pairpf2.so" ROUNDS=2 STEPS=300 bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_pair.so gpurun_ab/libvxba_pf2.so gpurun_ab/libvxba_pairpf2.so" ROUNDS=2 STEPS=100 BENCH_ARGS="--config cfg4" bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_pair.so gpurun_ab/libvxba_pf2.so gpurun_ab/libvxba_pairpf2.so" ROUNDS=1 STEPS=200 BENCH_ARGS="--config cfg3" bash scripts/gpu_abn.sh
VXBA_LIB=$PWD/gpurun_ab/libvxba_pair.so timeout 200 python scripts/dbg_timeline.py k3 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_s11_timeline_pair.txt; tail -16 gpurun_out/r4_s11_timeline_pair.txt
VXBA_LIB=$PWD/voxel-slam_amd/csrc/libvxba.so timeout 200 python scripts/dbg_timeline.py k3 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_s11_timeline_tree.txt; tail -16 gpurun_out/r4_s11_timeline_tree.txt
