#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_li_ba.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" | tail -6
for n in 300 3000 20000; do timeout 300 python scripts/dbg_li_stress.py 4000 $n 2>&1 | grep -v amdgpu.ids | tail -2; done
