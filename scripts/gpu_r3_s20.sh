#!/bin/bash
# round 3, session 20: local map -- margi_points as a workgroup per node; parity, then the scan cycle's stage times
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_map.py tests/test_gpu_local_mapping_cycle.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -2
for r in 1 2; do timeout 300 python scripts/dbg_map_cycle.py 2>&1 | grep -v amdgpu.ids | tail -9; done
