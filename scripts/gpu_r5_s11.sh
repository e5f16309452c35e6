#!/bin/bash
# round 5, session 11: residual sweep's eigen-solver leaves after two sweeps when the whole wave is at round-off (same-box A/B); parity after the
# removal of the in-launch reduction
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_s11
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py tests/test_gpu_li_ba.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -4
line() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']
    print('$1 it/s %.0f  us/step %.2f  k3 %.2f us (%.3f)  k2 %.2f us  k3fin %.2f us  solve+k2 %.2f us acc %s' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['frac'], r['k2_residual']['avg_launch_ms']*1e3, r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0), d['config']['lm_steps_accepted']))
"; }
for r in 1 2 3; do
  VXBA_LIB=$PWD/gpurun_ab/libvxba_cur.so timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | line cur
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | line eig2
done 2>&1 | tee gpurun_out/r5_s11/ab.txt
for cfg in cfg4; do
  VXBA_LIB=$PWD/gpurun_ab/libvxba_cur.so timeout 300 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | line cur_$cfg
  timeout 300 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | line eig2_$cfg
done 2>&1 | tee -a gpurun_out/r5_s11/ab.txt
