#!/bin/bash
# round 4: LI shell -- solve_seq published before the host's sequence word (one PCIe store acknowledgement less on the critical path): parity, stress, rate A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_li_ba.py tests/test_gpu_local_mapping_cycle.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -2
timeout 600 python scripts/dbg_li_stress.py 2>&1 | grep -v amdgpu | tail -2
for r in 1 2 3; do
  for lib in gpurun_ab/libvxba_k2tail.so voxel-slam_amd/csrc/libvxba.so; do
    echo "$lib: $(VXBA_LIB=$PWD/$lib timeout 300 python scripts/dbg_li_rate.py 2>&1 | grep -v amdgpu | tail -2 | tr '\n' ' ' | cut -c1-330)"
  done
done
