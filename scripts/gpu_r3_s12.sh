#!/bin/bash
# round 3, session 12: Hessian sweep prologue -- padding-only clear of the tiles, wave 0's batch before its pose wait; kernarg placement
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_hba.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_dropin.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -3
for m in k3 k3lm; do echo "== timeline $m"; timeout 300 python scripts/dbg_timeline.py $m 2>&1 | grep -v amdgpu.ids | sed -n 11,17p; done
run() { VXBA_LIB=$PWD/$1 timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba $2 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    r = d['roofline']
    print('$1 $2 $3 it/s %.0f  us/step %.2f  k3 %.2f  k2 %.2f  fin %.2f  solve+k2 %.2f acc %s' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['k2_residual']['avg_launch_ms']*1e3, r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0), d['config'].get('lm_steps_accepted')))
"; }
for r in 1 2; do
  for lib in gpurun_ab/libvxba_p0r0.so voxel-slam_amd/csrc/libvxba.so; do run $lib; done
done
for cfg in cfg4 cfg1; do for lib in gpurun_ab/libvxba_p0r0.so voxel-slam_amd/csrc/libvxba.so; do run $lib "--config $cfg"; done; done
