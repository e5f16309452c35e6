#!/bin/bash
# round 5 experiment: the residual sweep with a voxel's frames over a lane pair (VXBA_K2_PAIRS=1) against the shipped one lane per voxel, same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_k2pairs
export HSA_ENABLE_IPC_MODE_LEGACY=0
VXBA_K2_PAIRS=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_li_ba.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -x -q -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn\|test_gpu_parity.py::" | tail -5
for r in 1 2 3; do
  for m in 0 1; do
    VXBA_K2_PAIRS=$m timeout 300 python bench.py --steps 150 --warmup 15 --repeats 7 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('pairs=$m it/s %.0f  us/step %.2f  k3 %.2f  k2 %.2f us (%.3f)  solve+k2 %.2f' % (d['value'], 1e3 * d['ms_per_step'], 1e3 * r['avg_launch_ms'], 1e3 * r['k2_residual']['avg_launch_ms'], r['k2_residual']['frac'], 1e3 * r['solve_plus_k2_launch_avg_ms']))"
  done
done 2>&1 | tee gpurun_out/r5_k2pairs/ab.txt
for m in 0 1; do
  VXBA_K2_PAIRS=$m timeout 300 python bench.py --config cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('cfg4 pairs=$m it/s %.0f  us/step %.2f  k2 %.2f us (%.3f)  solve+k2 %.2f' % (d['value'], 1e3 * d['ms_per_step'], 1e3 * r['k2_residual']['avg_launch_ms'], r['k2_residual']['frac'], 1e3 * r['solve_plus_k2_launch_avg_ms']))"
done 2>&1 | tee -a gpurun_out/r5_k2pairs/ab.txt
