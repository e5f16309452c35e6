#!/bin/bash
# round 3, session 13: kernarg preload everywhere + flat K2 arguments -- whole GPU suite, smoke, A/B against the build before it, LI and map rates
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash scripts/gpu_full_tests.sh 2>&1 | tail -14
LIBS="gpurun_ab/libvxba_pre.so voxel-slam_amd/csrc/libvxba.so" ROUNDS=3 bash scripts/gpu_abn.sh
for lib in gpurun_ab/libvxba_pre.so voxel-slam_amd/csrc/libvxba.so; do
  echo "== $lib: default bench (LI + scan cycle lines)"
  VXBA_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    c = d['config']
    print('it/s %.0f us/step %.2f' % (d['value'], 1e3*d['ms_per_step']))
    for k in ('li_ba', 'scan_cycle'):
        if k in d: print(k, json.dumps(d[k])[:600])
        elif k in c: print(k, json.dumps(c[k])[:600])
"
done
