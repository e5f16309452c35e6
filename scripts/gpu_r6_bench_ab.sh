#!/bin/bash
# round 6: the driver's bench line with the fused residual + Hessian launch on / off (VXBA_FUSED_SWEEPS), alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/${OUT:-r6_bench_ab.txt}; : > $out
for r in $(seq 1 ${ROUNDS:-3}); do
  for fz in 1 0; do
    for st in ${STEPS:-20 150}; do
      VXBA_FUSED_SWEEPS=$fz timeout 600 python bench.py --steps $st --warmup 5 --no-cpu-baseline --no-li-ba --no-cold-l3 ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']; rp = d['repeats']
    print('fused=$fz steps=$st  it/s %.0f  us/step %.2f (min %.2f max %.2f)  %s %.2f us  fin %.2f  solve+k2 %.2f' % (d['value'], 1e3*d['ms_per_step'], 1e3*rp['ms_per_step_min'], 1e3*rp['ms_per_step_max'], r['kernel'], r['avg_launch_ms']*1e3, r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0)))
" >> $out
    done
  done
done
cat $out
