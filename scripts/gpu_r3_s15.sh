#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 0 1; do echo "== VXBA_FINALIZE_IN_LAUNCH=$v"; VXBA_FINALIZE_IN_LAUNCH=$v timeout 300 python scripts/dbg_timeline.py fused 2>&1 | grep -v amdgpu.ids | head -14; done
