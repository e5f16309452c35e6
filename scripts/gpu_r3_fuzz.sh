#!/bin/bash
# round 3: randomised GPU-vs-oracle sweep over the round's kernels (four-wave solve, residual sweep, LI shell with the in-launch pose solve)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/fuzz_r3.log
for seed in 31 32 33 34 35 36; do
  FUZZ_KINDS=lm,li,gravity,mixed,lm,li timeout 600 python scripts/fuzz_parity.py $seed 120 2>&1 | grep -v amdgpu | tail -2 >> gpurun_out/fuzz_r3.log
done
for seed in 41 42; do timeout 900 python scripts/fuzz_parity.py $seed 110 2>&1 | grep -v amdgpu | tail -2 >> gpurun_out/fuzz_r3.log; done
cat gpurun_out/fuzz_r3.log
