#!/bin/bash
# round 6, step 3: fused launch on / off, interleaved repeats, cfg2 / cfg3 / cfg4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python scripts/dbg_fused.py rate cfg2 cfg2 cfg3 cfg4 > gpurun_out/r6_s3_fused_rates.txt 2>&1; echo rc=$?; grep -v "amdgpu.ids" gpurun_out/r6_s3_fused_rates.txt | tail -40
