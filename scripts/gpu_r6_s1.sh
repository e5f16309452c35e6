#!/bin/bash
# round 6, step 1: first light of the fused residual + Hessian launch -- parity against the three-launch path, then rates
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/dbg_fused.py ${ARGS:-} > gpurun_out/r6_s1_fused.txt 2>&1; echo rc=$?; grep -v "amdgpu.ids" gpurun_out/r6_s1_fused.txt | tail -40
