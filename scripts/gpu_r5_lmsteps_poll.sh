#!/bin/bash
# round 5: vxba_lm_steps completing by polling against the library that sleeps in hipStreamSynchronize (gpurun_ab/libvxba_cur.so), the driver's flags and the defaults
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_lmsteps
export HSA_ENABLE_IPC_MODE_LEGACY=0
for r in 1 2 3; do
  for lib in "$GRAFT_REPO_ROOT/gpurun_ab/libvxba_cur.so" ""; do
    for fl in "--steps 20 --warmup 5" "--steps 300 --warmup 30"; do
      VXBA_LIB=$lib timeout 300 python bench.py $fl --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-24s it/s %.0f (%.0f .. %.0f)  us/step %.2f' % ('old' if '$lib' else 'new', '$fl', d['value'], d['repeats']['value_min'], d['repeats']['value_max'], 1e3 * d['ms_per_step']))"
    done
  done
done 2>&1 | tee gpurun_out/r5_lmsteps/ab.txt
