#!/bin/bash
# round 4, closing tree: every kind of the randomised sweep with fresh seeds (small and big windows, every entry point, release drives)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/fuzz_r4_all.log
for seed in 201 202 203 204; do
  FUZZ_KINDS=lm,li,gravity,mixed,lm,li timeout 900 python scripts/fuzz_parity.py $seed 300 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/fuzz_r4_all.log
done
for seed in 211 212 213; do timeout 900 python scripts/fuzz_parity.py $seed 220 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/fuzz_r4_all.log; done
for seed in 221 222; do FUZZ_KINDS=lm_big,li_big,mixed_big,lm_big timeout 1500 python scripts/fuzz_parity.py $seed 60 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/fuzz_r4_all.log; done
FUZZ_KINDS=map_release timeout 1500 python scripts/fuzz_parity.py 231 30 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/fuzz_r4_all.log
cat gpurun_out/fuzz_r4_all.log
