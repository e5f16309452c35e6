#!/bin/bash
# rocprofv3 kernel durations (GPU time stamps) of the LM loop with the previous build and with the tree's, cfg2 and cfg4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for cfg in cfg2 cfg4; do
for lib in gpurun_ab/libvxba_prev.so voxel-slam_amd/csrc/libvxba.so; do
  tag=$(basename $lib .so)_$cfg
  steps=90; [ $cfg = cfg4 ] && steps=30
  VXBA_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_ab_$tag" -o t -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps $steps --warmup 9 --repeats 3 --no-cpu-baseline --no-li-ba --no-cold-l3 > /dev/null 2>&1
  echo "== $tag"; python - "$GRAFT_REPO_ROOT/gpurun_out/prof_ab_$tag/t_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print("  %-60s calls %5s avg %8.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_ab_$tag"
done; done
