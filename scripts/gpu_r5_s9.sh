#!/bin/bash
# round 5, session 9: the hierarchical-BA pass below the C ABI -- tests, bench line (cpu_baseline + roofline), kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_s9
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest tests/test_gpu_hba.py -m gpu -q -s --timeout 900 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" | tail -15 ) 2>&1 | tee gpurun_out/r5_s9/pytest_hba.txt
for t in 1 2; do
  timeout 600 python bench.py --config cfg5 --steps 3 --warmup 1 --hba-threads $t --no-cpu-baseline 2>gpurun_out/r5_s9/bench_cfg5_t$t.err | tee gpurun_out/r5_s9/bench_cfg5_t$t.json | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('cfg5 threads $t: %.3f s per pass, roofline %s' % (d['ms_per_step'] / 1e3, json.dumps({k: d['roofline'][k] for k in ('achieved', 'frac', 'avg_launch_ms', 'launches')}) if d['roofline'] else None))
"
done
timeout 900 python bench.py --config cfg5 --steps 3 --warmup 1 2>gpurun_out/r5_s9/bench_cfg5.err | tee gpurun_out/r5_s9/bench_cfg5.json | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('cfg5: %.3f s per pass; cpu_baseline %s' % (d['ms_per_step'] / 1e3, json.dumps(d['cpu_baseline'])))
"
tail -3 gpurun_out/r5_s9/bench_cfg5.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r5_s9/prof_cfg5_trace" -o t -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r5_s9/prof_cfg5.log" 2>&1; echo "cfg5 trace rc=$?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/r5_s9/prof_cfg5_trace -type f -name "*_kernel_trace.csv" -size +8M -delete
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_s9/prof_cfg5.log
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r5_s9/prof_cfg5_trace/**/t_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print('total kernel ms over the run (1 warm-up + 3 timed + 1 measurement pass): %.1f' % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6))
for r in rows[:8]:
    print('%-60s calls %6s tot %8.2f ms avg %8.1f us' % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
