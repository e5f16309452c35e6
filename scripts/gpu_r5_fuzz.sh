#!/bin/bash
# round 5, closing tree: the randomised sweep on the rebuilt Hessian sweep (every LM kind, small and big windows, every entry point) + the new `hba` kind
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/fuzz_r5.log
for seed in 301 302 303; do
  FUZZ_KINDS=lm,li,gravity,mixed,lm,li timeout 900 python scripts/fuzz_parity.py $seed 300 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/fuzz_r5.log
done
for seed in 311 312; do timeout 900 python scripts/fuzz_parity.py $seed 220 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/fuzz_r5.log; done
for seed in 321 322 323; do FUZZ_KINDS=lm_big,li_big,mixed_big,lm_big timeout 1500 python scripts/fuzz_parity.py $seed 60 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/fuzz_r5.log; done
FUZZ_KINDS=hba timeout 1500 python scripts/fuzz_parity.py 331 40 2>&1 | grep -v amdgpu | grep -E "MISMATCH|cases" | tail -6 >> gpurun_out/fuzz_r5.log
cat gpurun_out/fuzz_r5.log
