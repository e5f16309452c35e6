#!/bin/bash
# round 5, session 1: the rebuilt Hessian sweep (4x4x4_4b blocks-as-K) -- K3 parity tests first, then the whole suite, then a bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_s1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" > gpurun_out/r5_s1/pytest_parity.log; echo "parity rc=${PIPESTATUS[0]}"; tail -15 gpurun_out/r5_s1/pytest_parity.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" > gpurun_out/r5_s1/pytest_gpu_full.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -25 gpurun_out/r5_s1/pytest_gpu_full.log
for cfg in cfg2 cfg4; do
timeout 600 python bench.py --config $cfg --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba --no-cold-l3 2>gpurun_out/r5_s1/bench_$cfg.err | tee gpurun_out/r5_s1/bench_$cfg.json | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    r = d['roofline']
    print('$cfg it/s %.0f  us/step %.2f  k3 %.2f us (%.3f)  k2 %.2f us  k3fin %.2f us  solve+k2 %.2f us acc %s' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['frac'], r['k2_residual']['avg_launch_ms']*1e3, r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0), d['config']['lm_steps_accepted']))
"
done
