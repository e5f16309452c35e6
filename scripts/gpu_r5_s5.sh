#!/bin/bash
# round 5, session 5: first batches before / behind the prologue barrier on the rebuilt sweep (2 / 4 / 6 / 8 waves), same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_s5
export HSA_ENABLE_IPC_MODE_LEGACY=0
line() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']
    print('$1 it/s %.0f  us/step %.2f  k3 %.2f us (%.3f)  k2 %.2f us  k3fin %.2f us  solve+k2 %.2f us acc %s' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['frac'], r['k2_residual']['avg_launch_ms']*1e3, r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0), d['config']['lm_steps_accepted']))
"; }
for r in 1 2; do
  for v in cur fw2 fw6 fw8; do
    VXBA_LIB=$PWD/gpurun_ab/libvxba_$v.so timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | line $v
  done
done 2>&1 | tee gpurun_out/r5_s5/ab_first_waves.txt
for v in cur fw8; do
  VXBA_LIB=$PWD/gpurun_ab/libvxba_$v.so timeout 300 python bench.py --config cfg4 --steps 200 --warmup 20 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | line ${v}_cfg4
done 2>&1 | tee -a gpurun_out/r5_s5/ab_first_waves.txt
