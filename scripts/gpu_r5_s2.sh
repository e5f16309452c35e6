#!/bin/bash
# round 5, session 2: K3 timeline of the rebuilt sweep, same-box A/B against the round-4 library, the self-launching 2-rank bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_s2
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/dbg_timeline.py k3 > gpurun_out/r5_s2/timeline_k3.txt 2>&1; tail -40 gpurun_out/r5_s2/timeline_k3.txt
ROUNDS=2 STEPS=300 bash scripts/gpu_ab.sh 2>&1 | tee gpurun_out/r5_s2/ab_cfg2.txt
timeout 900 python -m pytest tests/test_gpu_two_rank.py -m gpu -q --timeout 900 -p no:cacheprovider -k "bench" 2>&1 | tail -15 | tee gpurun_out/r5_s2/pytest_bench2.txt
