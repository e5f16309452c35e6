#!/bin/bash
# within-run A/B of two builds of libvxba.so (box-to-box variance is larger than most kernel changes):
#   gpurun_ab/libvxba_base.so  vs  voxel-slam_amd/csrc/libvxba.so, alternating, ROUNDS times
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
for r in $(seq 1 ${ROUNDS:-3}); do
  for lib in gpurun_ab/libvxba_base.so voxel-slam_amd/csrc/libvxba.so; do
    VXBA_LIB=$PWD/$lib timeout 600 python bench.py --steps ${STEPS:-300} --warmup 30 --no-cpu-baseline --no-li-ba --no-cold-l3 ${BENCH_ARGS:-} 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    r = d['roofline']
    print('$lib it/s %.0f  us/step %.2f  k3 %.2f us  k2 %.2f us  k3fin %.2f us  solve+k2 %.2f us acc %s' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['k2_residual']['avg_launch_ms']*1e3, r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0), d['config']['lm_steps_accepted']))
"
  done
done
