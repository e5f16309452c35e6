#!/bin/bash
# round 6, step 16: the residual half's three forms at the metric's size -- parity, then interleaved rates (VXBA_K23_MODE = 2 staged in LDS / 1 lane pair / 0 ring)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/dbg_fused.py parity cfg1 cfg2 2>&1 | grep -v amdgpu | tail -12
out=gpurun_out/r6_s16_modes.txt; : > $out
for r in 1 2 3; do
  for m in 2 1 0; do
    echo "== mode $m" >> $out
    VXBA_K23_MODE=$m timeout 600 python scripts/dbg_fused.py rate cfg2 2>&1 | grep -v amdgpu.ids | grep "fused=1" >> $out
  done
done
cat $out
