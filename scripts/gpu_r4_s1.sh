#!/bin/bash
# round 4, session 1: K3 with the next batch's loads behind descriptors instead of a branch (no phi copies / vmcnt waits in front of the barrier)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
ROUNDS=3 STEPS=300 bash scripts/gpu_ab.sh
VXBA_LIB=$PWD/voxel-slam_amd/csrc/libvxba.so timeout 200 python scripts/dbg_timeline.py k3 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_s1_timeline.txt; tail -25 gpurun_out/r4_s1_timeline.txt
for cfg in cfg3 cfg4; do for lib in gpurun_ab/libvxba_base.so voxel-slam_amd/csrc/libvxba.so; do
VXBA_LIB=$PWD/$lib timeout 300 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-li-ba 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    r = d['roofline']
    print('$cfg $lib it/s %.0f  us/step %.2f  k3 %.2f us frac %.3f' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['frac']))
"
done; done
