#!/bin/bash
# round 4, session 7: vxba_map_release on a 2000-scan drive; the map suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_map.py -m gpu -q -x --timeout 1200 -p no:cacheprovider -k "release" 2>&1 | tail -25
timeout 1500 python -m pytest tests/test_gpu_map.py tests/test_gpu_local_mapping_cycle.py tests/test_gpu_dropin.py -m gpu -q -x --timeout 1200 -p no:cacheprovider 2>&1 | tail -5
