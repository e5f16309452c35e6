#!/bin/bash
# round 3, session 5: full suite after reverting the residual sweep's arithmetic trims; the whole default bench line (scan_cycle, LI reference baseline)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" > gpurun_out/pytest_gpu_full.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -8 gpurun_out/pytest_gpu_full.log
( time timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | tail -3
tail -5 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k in ("li_ba", "scan_cycle", "cpu_baseline", "reject_window"):
    print(k, json.dumps(d.get(k))[:1500])
PY
