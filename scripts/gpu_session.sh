#!/bin/bash
# Run on the GPU box through gpurun: smoke, GPU parity tests, bench, rocprofv3 kernel trace.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -E "gfx|Compute Unit|Marketing" | head -6 > gpurun_out/rocminfo.txt
nproc > gpurun_out/host.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> gpurun_out/host.txt
echo "== smoke (no torch)"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --steps 150 --warmup 15 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -5 gpurun_out/bench.log
echo "== rocprof"; ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 60 --warmup 6 --no-cpu-baseline --no-li-ba > "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log" 2>&1 ); echo "rocprof rc=$?"; tail -3 gpurun_out/rocprof.log
find gpurun_out/prof -name "*stats*" | head; 
