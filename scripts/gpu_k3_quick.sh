#!/bin/bash
# bench line + per-wave timeline only (no tests): a 30 s look at the Hessian sweep
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for lib in ${LIBS:-voxel-slam_amd/csrc/libvxba.so}; do
echo "== $lib"
VXBA_LIB=$PWD/$lib timeout 300 python bench.py --steps ${STEPS:-150} --warmup 15 --no-cpu-baseline --no-li-ba --no-cold-l3 ${BENCH_ARGS:-} 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    r = d['roofline']
    print('it/s %.0f  us/step %.2f  k3 %.2f us (frac %.3f)  k2 %.2f us  k3fin %.2f us  solve+k2 %.2f us acc %s rej %s res %.9g' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['frac'], r['k2_residual']['avg_launch_ms']*1e3, r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0), d['config']['lm_steps_accepted'], d['config']['lm_steps_rejected'], d['config']['final_residual']))
"
[ -n "$TIMELINE" ] && VXBA_LIB=$PWD/$lib timeout 200 python scripts/dbg_timeline.py k3 2>&1 | grep -v amdgpu.ids
done
