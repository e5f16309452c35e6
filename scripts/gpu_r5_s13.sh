#!/bin/bash
# round 5, session 13: K1 for packed cells with sixteen lanes per cell -- bit-exactness tests, then cfg5 with the old kernel (VXBA_K1_LANE_PER_CELL=1) and the new one
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_s13
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_voxelize.py tests/test_gpu_hba.py tests/test_gpu_map.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -4
for r in 1 2; do
  for k in 1 0; do
    VXBA_K1_LANE_PER_CELL=$k timeout 600 python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']
    print('cfg5 lane-per-cell=$k: %.4f s per pass; k1 %.1f us avg over %d launches, %.0f GB/s = %.3f of HBM' % (d['ms_per_step'] / 1e3, 1e3 * r['avg_launch_ms'], r['launches'], r['achieved'], r['frac']))
"
  done
done 2>&1 | tee gpurun_out/r5_s13/ab_k1.txt
timeout 600 python bench.py --no-li-ba --no-cpu-baseline --no-cold-l3 --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('cfg2 scan_cycle', d.get('scan_cycle'))" | cut -c1-600
