#!/bin/bash
# One call: the full bench line (with cpu_baseline, li_ba, ...) + rocprofv3 kernel trace + HBM counter passes of the bench command.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py 2> gpurun_out/bench_full.err | grep "^{" > gpurun_out/bench_full.json; echo "bench rc=$? $(wc -c < gpurun_out/bench_full.json) bytes"; tail -3 gpurun_out/bench_full.err
STEPS=${STEPS:-90} bash scripts/gpu_profile.sh 2>&1 | tail -8
