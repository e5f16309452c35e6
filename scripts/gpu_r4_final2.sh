#!/bin/bash
# round 4, final 2: rocprofv3 evidence from the final kernel sources (kernel trace + the two HBM counter passes at cfg2 / cfg3 / cfg4, the cold-L3 rotation,
# the LI loop, the scan cycle), the cfg5-size oracle comparison, cfg5 bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
CONFIG=cfg2 STEPS=90 bash scripts/gpu_profile_cfg.sh
CONFIG=cfg3 STEPS=60 bash scripts/gpu_profile_cfg.sh
CONFIG=cfg4 STEPS=30 bash scripts/gpu_profile_cfg.sh
TAG=cold CMD="python $GRAFT_REPO_ROOT/scripts/dbg_cold_l3.py cfg2" bash scripts/gpu_profile_cfg.sh
VXBA_RUN_SLOW=1 timeout 1800 python -m pytest tests/test_gpu_hba.py -m gpu -q -s -k cfg5_size --timeout 1700 -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/r4_final_cfg5_size_test.txt; cat gpurun_out/r4_final_cfg5_size_test.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_li6" -o t -- python $GRAFT_REPO_ROOT/scripts/dbg_li_rate.py > "$GRAFT_REPO_ROOT/gpurun_out/prof_li6.log" 2>&1; echo "li trace rc=$?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof_li6 -name "*_kernel_trace.csv" -size +12M -delete
bash scripts/gpu_profile_map.sh > gpurun_out/r4_final_map_profile.txt 2>&1; tail -3 gpurun_out/r4_final_map_profile.txt
timeout 900 python bench.py --config cfg5 --steps 2 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r4_final_cfg5_n1.json; cut -c1-300 gpurun_out/r4_final_cfg5_n1.json
du -sh gpurun_out
