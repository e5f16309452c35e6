#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in ${VARIANTS:-0 2 3 4}; do
  echo "== VXBA_K3_SGB=$v"
  VXBA_K3_SGB=$v timeout 600 python bench.py --steps ${STEPS:-90} --warmup 9 --no-cpu-baseline --no-li-ba 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    r = d['roofline']
    print('it/s %.0f  ms/step %.4f  k3 %.2f us (%.1f%% hbm)  k2 %.2f us (%.1f%%)  k3fin %.2f us' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, 100*r['frac'], r['k2_residual']['avg_launch_ms']*1e3, 100*r['k2_residual']['frac'], r['k3_finalize_avg_ms']*1e3))
"
done
