#!/bin/bash
# round 6, step 5: the voxeliser's layers >= 1 as a partition of the previous layer's order + counts posted into pinned memory: parity tests, then cfg5 A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_voxelize.py tests/test_gpu_hba.py tests/test_gpu_wide.py tests/test_gpu_map.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" | tail -15
for r in 1 2; do
  for pz in 1 0; do
    VXBA_VOXELIZE_PARTITION=$pz timeout 600 python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('partition=$pz  passes/s %.3f  ms/pass %.2f' % (d['value'], d['ms_per_step']), json.dumps(d.get('roofline', {}))[:300])
"
  done
done 2>&1 | tee gpurun_out/r6_s5_cfg5_ab.txt
