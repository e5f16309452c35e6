#!/bin/bash
# build gpurun_ab/libvxba_<name>.so = the tree's library with vxba_kernels.hip recompiled under extra -D flags (same-box A/B: scripts/gpu_abn.sh)
#   scripts/build_variant.sh pose "-DK3_POSE_REGS=1"
set -e
cd "$(dirname "$0")/../voxel-slam_amd/csrc"
name=$1; shift
mkdir -p ../../gpurun_ab
/opt/rocm/bin/hipcc -I/opt/rocm/include -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=14 -Wno-unused-value -Wno-unused-result "$@" -c vxba_kernels.hip -o /tmp/vxba_kernels_$name.o
objs=$(ls *.o | grep -v '^vxba_kernels.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_ab/libvxba_$name.so /tmp/vxba_kernels_$name.o $objs -ldl
echo built gpurun_ab/libvxba_$name.so
