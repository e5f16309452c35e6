#!/bin/bash
# round 3, session 1: the four-wave solve.  Full GPU suite, timelines of the fused launch for the base and the new build, same-box A/B,
# head-start / voxels-per-wave sweeps of the new build.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" > gpurun_out/pytest_gpu_full.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -15 gpurun_out/pytest_gpu_full.log
for lib in gpurun_ab/libvxba_base.so voxel-slam_amd/csrc/libvxba.so; do
  echo "== timeline $lib"
  VXBA_LIB=$PWD/$lib timeout 300 python scripts/dbg_timeline.py fused 2>&1 | grep -v amdgpu.ids | tail -8
done
ROUNDS=2 STEPS=300 bash scripts/gpu_ab.sh
run() { timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-li-ba 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    r = d['roofline']
    print('$1 it/s %.0f  us/step %.2f  k3 %.2f  k2 %.2f  fin %.2f  solve+k2 %.2f' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_ms']*1e3, r['k2_residual']['avg_launch_ms']*1e3, r['k3_finalize_avg_ms']*1e3, 1e3*r.get('solve_plus_k2_launch_avg_ms', 0)))
"; }
for hs in 0 30 60 100 160; do VXBA_K2_HEAD_START=$hs run "head_start=$hs"; done
for vpb in 49 56; do VXBA_K2_VPB=$vpb run "vpb=$vpb hs=100"; VXBA_K2_VPB=$vpb VXBA_K2_HEAD_START=30 run "vpb=$vpb hs=30"; done
