#!/bin/bash
# round 4, session 2: phase A of the Hessian sweep under finer stamps; pose-in-registers / parameters-first variants; the new bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
VXBA_LIB=$PWD/voxel-slam_amd/csrc/libvxba.so timeout 200 python scripts/dbg_timeline.py k3 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_s2_timeline.txt; tail -12 gpurun_out/r4_s2_timeline.txt
LIBS="gpurun_ab/libvxba_base.so voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_pose.so gpurun_ab/libvxba_pfirst.so" ROUNDS=2 STEPS=300 bash scripts/gpu_abn.sh
LIBS="voxel-slam_amd/csrc/libvxba.so gpurun_ab/libvxba_pose.so gpurun_ab/libvxba_pfirst.so" ROUNDS=1 STEPS=100 BENCH_ARGS="--config cfg4" bash scripts/gpu_abn.sh
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_s2_bench_driver_flags.json 2> gpurun_out/r4_s2_bench_driver_flags.err; tail -c 3000 gpurun_out/r4_s2_bench_driver_flags.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'repeats', d['repeats']); r = d['roofline']; print({k: r[k] for k in ('bound','frac','avg_launch_ms','working_set_bytes','fits_infinity_cache','cold_l3','mfma')})
print('li_ba', d.get('li_ba', {}).get('ms_per_iteration_inside_the_call'), 'scan', d.get('scan_cycle', {}).get('stage_ms'))
"
tail -3 gpurun_out/r4_s2_bench_driver_flags.err
