#!/bin/bash
# round 6: the hierarchical pass after its restructuring (closing window, bottom / export / import / top, dist.hba_pass)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_hba.py tests/test_gpu_two_rank.py -q --timeout 900 -p no:cacheprovider -x -k "hba or hierarchical or cfg5" 2>&1 | tail -25
timeout 600 python bench.py --config cfg5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -c 1500
