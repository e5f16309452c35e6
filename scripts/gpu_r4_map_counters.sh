#!/bin/bash
# round 4: the map's counter block through a reset kernel / a publish kernel into mapped host memory instead of two blit copies per stage: tests, then stage times against the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_map.py tests/test_gpu_local_mapping_cycle.py -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -2
for r in 1 2 3; do
  for lib in gpurun_ab/libvxba_mapload.so voxel-slam_amd/csrc/libvxba.so; do
    echo "== $lib $(VXBA_LIB=$PWD/$lib timeout 300 python scripts/dbg_map_cycle.py 2>&1 | grep -v amdgpu.ids | tail -8 | awk '{print $(NF-1)}' | tr '\n' ' ')"
  done
done
