#!/bin/bash
# rocprofv3 kernel trace of a few LI_BA_Optimizer calls: start / end timestamps per kernel -> where the GPU idles inside a call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/li_trace" -o t -- python $R/scripts/dbg_li_phases.py > "$R/gpurun_out/li_trace.log" 2>&1; echo "rc=$?"
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/li_trace/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in rows]
ks.sort()
# last ~40 kernels: the last calls
tail = ks[-36:]
t0 = tail[0][0]
prev_end = None
for s, e, n in tail:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%9.1f us  +%6.1f gap  dur %7.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, n))
    prev_end = e
PY
