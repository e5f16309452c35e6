#!/bin/bash
# round 3, session 11: where the first 6k cycles of the Hessian sweep go (prologue stamps)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for m in k3 k3lm; do echo "== timeline $m"; timeout 300 python scripts/dbg_timeline.py $m 2>&1 | grep -v amdgpu.ids | head -22; done
