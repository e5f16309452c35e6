#!/bin/bash
# round 4: randomised GPU-vs-oracle sweep over windows large enough for the Hessian sweep's steady-state loop at every window size
# (1-3 full steps + a ragged one per workgroup, 12k-100k voxels), LM / LiDAR-inertial / mixed kinds
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/fuzz_r4_big.log
for seed in ${SEEDS:-91 92 93}; do
  FUZZ_KINDS=lm_big,li_big,mixed_big,lm_big timeout 1500 python scripts/fuzz_parity.py $seed ${CASES:-40} 2>&1 | grep -v amdgpu | tail -${TAIL:-1} >> gpurun_out/fuzz_r4_big.log
done
cat gpurun_out/fuzz_r4_big.log
