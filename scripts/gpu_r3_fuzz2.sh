#!/bin/bash
# round 3, closing tree: a second randomised sweep with fresh seeds (the kernels changed since the first: prologue, argument preload, solve's done test)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/fuzz_r3b.log
for seed in 51 52 53 54 55 56 57 58; do
  FUZZ_KINDS=lm,li,gravity,mixed,lm,li timeout 900 python scripts/fuzz_parity.py $seed 250 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/fuzz_r3b.log
done
for seed in 61 62 63; do timeout 900 python scripts/fuzz_parity.py $seed 200 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/fuzz_r3b.log; done
cat gpurun_out/fuzz_r3b.log
