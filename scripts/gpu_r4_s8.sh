#!/bin/bash
# round 4, session 8: LI entry point -- fallback test, host split of a call, 12 000-solve stress (log for profiles/), kernel trace of the LI loop
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_li_ba.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -4
VXBA_LI_TIMING=1 timeout 300 python scripts/dbg_li_rate.py 2>&1 | grep -v amdgpu.ids | tail -8
timeout 900 python scripts/dbg_li_stress.py 12000 20000 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r4_li_stress_12000.txt; cat gpurun_out/r4_li_stress_12000.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_li4" -o t -- python $GRAFT_REPO_ROOT/scripts/dbg_li_rate.py > "$GRAFT_REPO_ROOT/gpurun_out/prof_li4.log" 2>&1; echo "li trace rc=$?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof_li4 -name "*_kernel_trace.csv" -size +12M -delete; du -sh gpurun_out/prof_li4
