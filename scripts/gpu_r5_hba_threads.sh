#!/bin/bash
# round 5: hierarchical pass (cfg5) with 1 .. 8 host threads / streams at the bottom level, same box; A = the library with two workers / two arena slots
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_hba_threads
export HSA_ENABLE_IPC_MODE_LEGACY=0
python scripts/dbg_launch_latency.py
one() {  # $1 label, $2 threads, $3 library ("" = the tree's)
  VXBA_LIB=$3 timeout 600 python bench.py --config cfg5 --hba-threads $2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5_hba_threads/bench_$1.json 2> gpurun_out/r5_hba_threads/bench_$1.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5_hba_threads/bench_$1.json").read().strip().splitlines()[-1])
print("$1 threads $2: %.1f ms per pass" % d["ms_per_step"])
PY
}
for r in 1 2; do
  one A2_$r 2 $GRAFT_REPO_ROOT/gpurun_ab/libvxba_cur.so
  one B1_$r 1 ""
  one B2_$r 2 ""
  one B4_$r 4 ""
  one B6_$r 6 ""
done
timeout 1200 python -m pytest tests/test_gpu_hba.py tests/test_gpu_voxelize.py tests/test_gpu_local_mapping_cycle.py -x -q 2>&1 | tail -3
