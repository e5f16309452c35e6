#!/bin/bash
# round 6, step 4: lane-pair residual half + LDS hand-over of the first step's plane parameters -- parity, then rates with the pair mode on / off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python scripts/dbg_fused.py parity cfg1 cfg2 cfg3 > gpurun_out/r6_s4_parity.txt 2>&1; echo rc=$?; grep -v "amdgpu.ids" gpurun_out/r6_s4_parity.txt | tail -20
timeout 600 python scripts/dbg_fused.py rate cfg2 cfg2 cfg3 > gpurun_out/r6_s4_rates_pair.txt 2>&1; echo rc=$?; grep -v "amdgpu.ids" gpurun_out/r6_s4_rates_pair.txt | tail -20
VXBA_K23_PAIR=0 timeout 600 python scripts/dbg_fused.py rate cfg2 cfg2 > gpurun_out/r6_s4_rates_nopair.txt 2>&1; echo rc=$?; grep -v "amdgpu.ids" gpurun_out/r6_s4_rates_nopair.txt | tail -20
