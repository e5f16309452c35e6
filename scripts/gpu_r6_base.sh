#!/bin/bash
# round 6, baseline: the round-5 tree on today's box -- full GPU suite, smoke, the driver's bench line and a 150-step line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" > gpurun_out/r6_base_pytest.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -5 gpurun_out/r6_base_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r6_base_bench20.json 2> gpurun_out/r6_base_bench20.err; tail -c 600 gpurun_out/r6_base_bench20.json
timeout 600 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-li-ba > gpurun_out/r6_base_bench150.json 2> gpurun_out/r6_base_bench150.err; tail -c 400 gpurun_out/r6_base_bench150.json
