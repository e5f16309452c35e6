#!/bin/bash
# round 5, session 12: the nnz-proportional Hessian sweep for sparse / banded windows -- parity, then same-process A/B by option on cfg2_sparse, and cfg2 (must not move)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_s12
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_sparse.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" | tail -25
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -4
line() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']
    print('$1 it/s %.0f  us/step %.2f  k3 %.2f us (%.3f of its own %.1f MB)  k2 %.2f us  solve+k2 %.2f us acc %s' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['frac'], r['algorithmic_bytes_per_launch']/1e6, r['k2_residual']['avg_launch_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0), d['config']['lm_steps_accepted']))
"; }
for r in 1 2; do
  for cfg in cfg2_sparse cfg2; do
    VXBA_SPARSE_SWEEP=0 timeout 300 python bench.py --config $cfg --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | line ${cfg}_dense_sweep
    VXBA_SPARSE_SWEEP=1 timeout 300 python bench.py --config $cfg --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | line ${cfg}_sparse_sweep
  done
done 2>&1 | tee gpurun_out/r5_s12/ab_sparse.txt
