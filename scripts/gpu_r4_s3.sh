#!/bin/bash
# round 4, session 3: the next batch's requests interleaved with the K-steps of phase M (1 / 2 / 4 per K-step) against requests in front of the barrier;
# LI entry point base vs tree; new parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "lm_steps" 2>&1 | tail -5
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_late1.so gpurun_ab/libvxba_late2.so gpurun_ab/libvxba_late4.so" ROUNDS=2 STEPS=300 bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_late1.so gpurun_ab/libvxba_late2.so gpurun_ab/libvxba_late4.so" ROUNDS=1 STEPS=100 BENCH_ARGS="--config cfg4" bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_late1.so gpurun_ab/libvxba_late2.so gpurun_ab/libvxba_late4.so" ROUNDS=1 STEPS=200 BENCH_ARGS="--config cfg3" bash scripts/gpu_abn.sh
for lib in gpurun_ab/libvxba_base.so voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_base.so voxel-slam_amd/csrc/libvxba.so; do echo "== $lib"; VXBA_LIB=$PWD/$lib timeout 300 python scripts/dbg_li_rate.py 2>&1 | grep -v amdgpu.ids | tail -2; done
