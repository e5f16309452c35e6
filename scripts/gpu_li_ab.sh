#!/bin/bash
# LiDAR-inertial shells: parity tests, then the same timing script alternating two builds of libvxba.so on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_li_ba.py tests/test_gpu_dropin.py -m gpu -q -x --timeout 600 2>&1 | tail -5
for rep in 1 2; do
  for lib in ${LIBS:-gpurun_ab/libvxba_prev.so voxel-slam_amd/csrc/libvxba.so}; do
    echo "== $lib"
    VXBA_LIB=$PWD/$lib VXBA_LI_TIMING=${TIMING:-0} timeout 300 python scripts/dbg_li_rate.py 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-6}
  done
done
