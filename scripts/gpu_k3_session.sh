#!/bin/bash
# K3 development session on the GPU box: parity first (fail fast), then the bench line and the per-wave timeline.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== parity (K1-K4, LM, mixed)"; timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 2>&1 | tail -15
echo "== bench"; timeout 300 python bench.py --steps ${STEPS:-150} --warmup 15 --no-cpu-baseline --no-li-ba 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_k3.json | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    r = d['roofline']
    print('it/s %.0f  us/step %.2f  k3 %.2f us (frac %.3f)  k2 %.2f us  k3fin %.2f us  solve+k2 %.2f us acc %s rej %s' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['frac'], r['k2_residual']['avg_launch_ms']*1e3, r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0), d['config']['lm_steps_accepted'], d['config']['lm_steps_rejected']))
"
echo "== timeline"; timeout 200 python scripts/dbg_timeline.py k3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/timeline_k3.txt
if [ -n "$FULL" ]; then echo "== full gpu suite"; timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -8; fi
