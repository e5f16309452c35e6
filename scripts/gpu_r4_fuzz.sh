#!/bin/bash
# round 4, closing tree: randomised GPU-vs-oracle sweep with fresh seeds on the round-4 kernels (gated / interleaved requests, rows stored as finished,
# LI record in device memory, sharded voxelisation): LM / LI / gravity / mixed kinds with many cases, then every entry point
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/fuzz_r4.log
for seed in 71 72 73 74 75 76 77 78; do
  FUZZ_KINDS=lm,li,gravity,mixed,lm,li timeout 900 python scripts/fuzz_parity.py $seed 250 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/fuzz_r4.log
done
for seed in 81 82 83; do timeout 900 python scripts/fuzz_parity.py $seed 200 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/fuzz_r4.log; done
cat gpurun_out/fuzz_r4.log
