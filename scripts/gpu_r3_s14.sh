#!/bin/bash
# round 3, session 14: the Hessian reduction as a phase of the residual-sweep launch (VXBA_OPT_FINALIZE_IN_LAUNCH) -- parity, then on / off on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -2
run() { timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba $2 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    r = d['roofline']
    print('$1 $2 it/s %.0f  us/step %.2f  k3 %.2f  k2 %.2f  fin %.2f  solve+k2 %.2f acc %s' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['k2_residual']['avg_launch_ms']*1e3, r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0), d['config'].get('lm_steps_accepted')))
"; }
for r in 1 2 3; do
  VXBA_FINALIZE_IN_LAUNCH=0 run own_kernel
  VXBA_FINALIZE_IN_LAUNCH=1 run in_launch
done
for cfg in cfg4 cfg1; do
  VXBA_FINALIZE_IN_LAUNCH=0 run own_kernel "--config $cfg"
  VXBA_FINALIZE_IN_LAUNCH=1 run in_launch "--config $cfg"
done
