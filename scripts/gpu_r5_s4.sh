#!/bin/bash
# round 5, session 4: where do kernel arguments live?  HIP_FORCE_DEV_KERNARG A/B on the LM loop (the Hessian sweep's prologue waits for its argument tail)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_s4
export HSA_ENABLE_IPC_MODE_LEGACY=0
export VXBA_K3_PREFETCH=0
line() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']
    print('$1 it/s %.0f  us/step %.2f  k3 %.2f us (%.3f)  k2 %.2f us  k3fin %.2f us  solve+k2 %.2f us acc %s' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['frac'], r['k2_residual']['avg_launch_ms']*1e3, r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0), d['config']['lm_steps_accepted']))
"; }
for r in 1 2; do
  HIP_FORCE_DEV_KERNARG=0 timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | line devkernarg0
  HIP_FORCE_DEV_KERNARG=1 timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | line devkernarg1
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | line default
done 2>&1 | tee gpurun_out/r5_s4/ab_kernarg.txt
