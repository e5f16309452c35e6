#!/bin/bash
# round 3, session 7: LiDAR-inertial shell with the reduced pose system solved inside the residual-sweep launch
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_li_ba.py tests/test_gpu_edges.py tests/test_gpu_dropin.py tests/test_golden.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" > gpurun_out/pytest_li.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -25 gpurun_out/pytest_li.log
timeout 300 python scripts/dbg_li_pose_solve_ab.py 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/dbg_li_stress.py 3000 2>&1 | grep -v amdgpu.ids | tail -3
