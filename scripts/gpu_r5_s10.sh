#!/bin/bash
# round 5, session 10: the whole -m gpu suite + smoke on the committed tree; the finalize-in-launch option with its stderr
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_s10
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" > gpurun_out/r5_s10/pytest_gpu_full.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -12 gpurun_out/r5_s10/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
VXBA_FINALIZE_IN_LAUNCH=1 timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba --no-cold-l3 2>gpurun_out/r5_s10/fin_in_launch.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('fin_in_launch it/s %.0f  us/step %.2f  k3 %.2f us  solve+k2 %.2f us' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0)))
"; tail -5 gpurun_out/r5_s10/fin_in_launch.err
