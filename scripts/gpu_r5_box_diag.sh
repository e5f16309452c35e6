#!/bin/bash
# round 5: why the hierarchical pass with several host threads is 3-5x slower on some boxes -- what the box gives the container
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_box
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "nproc $(nproc)  loadavg $(cat /proc/loadavg)"
echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  cpuset $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | cut -c1-80)"
echo "cfs quota $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) period $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
grep -i "Cpus_allowed_list" /proc/self/status
cat /sys/fs/cgroup/cpu.stat 2>/dev/null
python scripts/dbg_launch_latency.py
for t in 0 1 4 0 1 4; do
  thr0=$(grep throttled_usec /sys/fs/cgroup/cpu.stat | cut -d" " -f2); use0=$(grep usage_usec /sys/fs/cgroup/cpu.stat | cut -d" " -f2); t0=$(date +%s.%N)
  timeout 600 python bench.py --config cfg5 --hba-threads $t --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/r5_box/err_$t.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads $t: %.1f ms per pass' % d['ms_per_step'], d['host'])"
  thr1=$(grep throttled_usec /sys/fs/cgroup/cpu.stat | cut -d" " -f2); use1=$(grep usage_usec /sys/fs/cgroup/cpu.stat | cut -d" " -f2); t1=$(date +%s.%N)
  python -c "print('   wall %.1f s, cpu used %.1f s, throttled %.3f s' % ($t1 - $t0, ($use1 - $use0) / 1e6, ($thr1 - $thr0) / 1e6))"
done
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -6
echo "loadavg $(cat /proc/loadavg)"
} 2>&1 | tee gpurun_out/r5_box/diag.txt
