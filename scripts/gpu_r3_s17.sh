#!/bin/bash
# round 3, session 17: LI shell -- information matrices re-used across calls, light state reset
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_li_ba.py tests/test_gpu_edges.py tests/test_gpu_local_mapping_cycle.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -3
timeout 300 python scripts/dbg_li_phases.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python scripts/dbg_li_stress.py 3000 300 2>&1 | grep -v amdgpu.ids | tail -2
