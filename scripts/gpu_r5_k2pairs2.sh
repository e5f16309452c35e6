#!/bin/bash
# round 5 experiment, part 2: where the lane-pair residual sweep starts to pay (cfg3 = 100k voxels; cfg2 again for the box)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_k2pairs
export HSA_ENABLE_IPC_MODE_LEGACY=0
for r in 1 2; do
for cfg in cfg3 cfg2 cfg4; do
  for m in 0 1; do
    VXBA_K2_PAIRS=$m timeout 300 python bench.py --config $cfg --steps 60 --warmup 6 --repeats 5 --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$cfg pairs=$m it/s %.0f  us/step %.2f  k3 %.2f  k2 %.2f us (%.3f)  solve+k2 %.2f' % (d['value'], 1e3 * d['ms_per_step'], 1e3 * r['avg_launch_ms'], 1e3 * r['k2_residual']['avg_launch_ms'], r['k2_residual']['frac'], 1e3 * r['solve_plus_k2_launch_avg_ms']))"
  done
done
done 2>&1 | tee gpurun_out/r5_k2pairs/ab2.txt
