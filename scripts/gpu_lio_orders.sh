#!/bin/bash
# odometry sweep under rocprofv3 for three scan orders: shuffled over all planes (worst case), few planes shuffled, few planes in grid order
cd /tmp && export TMPDIR=/tmp
for m in shuffled few_planes coherent; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$m -o t -- python /root/repo/scripts/dbg_lio_timing.py 100000 $m > /tmp/p_$m.log 2>&1
  grep "sweep:" /tmp/p_$m.log
  python3 - $m <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/p_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True):
    for row in csv.DictReader(open(f)):
        if 'lio_sweep' in row['Name']:
            print('%s: lio_sweep_kernel avg %.2f us over %s calls' % (sys.argv[1], float(row['AverageNs']) / 1e3, row['Calls']))
PY
done
