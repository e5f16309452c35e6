"""Shrink a rocprofv3 t_counter_collection.csv in place to one row per (kernel, counter): Counter_Value = mean over the launches, Launches = how many
(scripts/collect_profile_cfg.py reads both forms).  The per-dispatch table of a hierarchical-BA pass is 50 MB; gpurun merges 64 MiB back."""
import csv, sys
from collections import defaultdict
for path in sys.argv[1:]:
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            a = acc[(r["Kernel_Name"], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Launches"])
        for (k, c), (s, n) in sorted(acc.items()):
            w.writerow([k, c, repr(s / n), n])
    print(path, len(acc), "rows")
