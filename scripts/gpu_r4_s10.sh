#!/bin/bash
# round 4, session 10: LI record in host-written device memory: parity + stress, A/B against the record in mapped host memory
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_li_ba.py tests/test_gpu_local_mapping_cycle.py tests/test_gpu_edges.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -4
for v in 0 1 0 1; do echo "== VXBA_LI_REC_VRAM=$v"; VXBA_LI_REC_VRAM=$v VXBA_LI_TIMING=1 timeout 300 python scripts/dbg_li_rate.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-330; done
timeout 600 python scripts/dbg_li_stress.py 6000 20000 2>&1 | grep -v amdgpu.ids | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_li5" -o t -- python $GRAFT_REPO_ROOT/scripts/dbg_li_rate.py > "$GRAFT_REPO_ROOT/gpurun_out/prof_li5.log" 2>&1; echo "li trace rc=$?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof_li5 -name "*_kernel_trace.csv" -size +12M -delete; head -4 gpurun_out/prof_li5/t_kernel_stats.csv | cut -c1-160
