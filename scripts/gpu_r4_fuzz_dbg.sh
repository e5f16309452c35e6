#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
for lib in gpurun_ab/libvxba_base.so gpurun_ab/libvxba_prev.so voxel-slam_amd/csrc/libvxba.so; do
  echo "=== $lib"
  VXBA_LIB=$PWD/$lib FUZZ_KINDS=lm,li,gravity,mixed,lm,li timeout 900 python scripts/fuzz_parity.py 71 250 2>&1 | grep -v amdgpu | grep -B2 -A6 "MISMATCH" | head -40
done
