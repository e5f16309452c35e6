#!/bin/bash
# round 6, step 6: cfg5 under rocprofv3 --stats with the partition on / off (why is one of them 8x slower on some boxes?)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pz in 1 0; do
  VXBA_VOXELIZE_PARTITION=$pz timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg5_p$pz -o t -- python $R/bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_cfg5_p$pz.log 2>&1; echo "p$pz rc=$?"
  grep '^{' $R/gpurun_out/prof_cfg5_p$pz.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('partition=$pz ms/pass %.2f' % d['ms_per_step'])
"
done
cd $R; find gpurun_out/prof_cfg5_p* -type f -name "*_kernel_trace.csv" -size +8M -delete; du -sh gpurun_out/prof_cfg5_p*
