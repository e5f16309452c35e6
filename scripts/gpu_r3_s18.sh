#!/bin/bash
# round 3, session 18: head start of the solve over the voxel waves' loads, re-swept after the argument preload
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
for r in 1 2; do for hs in 100 0 40 70 140 200; do
VXBA_K2_HEAD_START=$hs timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']
    print('head_start $hs  it/s %.0f  us/step %.2f  k3 %.2f  solve+k2 %.2f' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0)))
"; done; done
