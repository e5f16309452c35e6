#!/bin/bash
# round 6: the whole -m gpu suite (no -x: collect every failure), smoke, the driver's bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" > gpurun_out/r6_pytest_gpu_full.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -30 gpurun_out/r6_pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench20.json 2> gpurun_out/r6_bench20.err; python - <<'PY'
import json
for l in open("gpurun_out/r6_bench20.json"):
    try: d = json.loads(l)
    except Exception: continue
    print("bench --steps 20:", d["value"], d["ms_per_step"], json.dumps(d["roofline"])[:600])
PY
