#!/bin/bash
# quick iteration loop on the GPU box: parity tests + bench variants
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -5
for v in ${VARIANTS:-0}; do
  echo "== VXBA_K3_SGB=$v"
  VXBA_K3_SGB=$v timeout 600 python bench.py --steps ${STEPS:-150} --warmup 15 --no-cpu-baseline --no-li-ba 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    r = d['roofline']
    print('it/s %.0f  ms/step %.4f  k3 %.2f us (%.1f%% hbm)  k2 %.2f us (%.1f%%)  k3fin %.2f us  solve+k2 %.2f us' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, 100*r['frac'], r['k2_residual']['avg_launch_ms']*1e3, 100*r['k2_residual']['frac'], r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0)))
"
done
