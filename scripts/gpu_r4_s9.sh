#!/bin/bash
# round 4, session 9: map push with eight lanes per leaf + descend without same-address atomics: parity (bit-exact map tests), cycle stage times, kernel trace; LI fallback test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_map.py tests/test_gpu_local_mapping_cycle.py tests/test_gpu_dropin.py tests/test_gpu_li_ba.py -m gpu -q -x --timeout 1200 -p no:cacheprovider 2>&1 | tail -12
for v in 1 0 1 0; do echo "== VXBA_MAP_PUSH1=$v"; VXBA_MAP_PUSH1=$v timeout 300 python scripts/dbg_map_cycle.py 2>&1 | grep -v amdgpu.ids | tail -10; done
bash scripts/gpu_profile_map.sh 2>&1 | tail -45
