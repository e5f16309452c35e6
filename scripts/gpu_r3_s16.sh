#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
for lib in voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_nooff1.so; do echo "== $lib"; VXBA_LIB=$PWD/$lib VXBA_FINALIZE_IN_LAUNCH=1 timeout 300 python scripts/dbg_timeline.py fused 2>&1 | grep "kernel entry\|in-launch\|since kernel entry"; done
