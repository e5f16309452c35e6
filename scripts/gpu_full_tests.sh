#!/bin/bash
# the whole -m gpu suite (no -x: collect every failure), then smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" > gpurun_out/pytest_gpu_full.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -40 gpurun_out/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
