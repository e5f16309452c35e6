#!/bin/bash
# round 3, closing call: whole GPU suite + smoke, profile round (bench line, kernel trace, HBM counters), then the bench line once more (with the counters of this tree attached)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash scripts/gpu_full_tests.sh 2>&1 | tail -7
STEPS=90 bash scripts/gpu_profile_round.sh 2>&1 | tail -2
