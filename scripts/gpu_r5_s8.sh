#!/bin/bash
# round 5, session 8: where a cfg5 pass spends its time today (kernel trace + host profile), the 500-keyframe comparison timed, the self-launching bench test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_s8
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_two_rank.py -m gpu -q --timeout 900 -p no:cacheprovider -k "bench" 2>&1 | tail -3
( time VXBA_RUN_SLOW=1 timeout 900 python -m pytest tests/test_gpu_hba.py -m gpu -q -s --timeout 900 -p no:cacheprovider -k "cfg5_size" 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r5_s8/cfg5_test.txt
timeout 600 python -c "
import cProfile, pstats, sys, time
sys.path.insert(0, '.')
import torch
from voxel_slam_amd import hba, synth, vxba
clouds, poses, gt = synth.corridor_session(500, 20000, synth.MASTER_SEED + 5000)
coarse = vxba.VoxelizeParams(voxel_size=2.0, max_layer=2, min_points=10, min_eigen_value=0.02, eigen_ratio=(1 / 9, 1 / 9, 1 / 9, 1 / 9))
fine = vxba.VoxelizeParams(voxel_size=1.0, max_layer=2, min_points=10, min_eigen_value=0.01, eigen_ratio=(1 / 16, 1 / 16, 1 / 9, 1 / 9))
hba.hierarchical_ba(clouds, poses, coarse, fine, wdsize=10, mgsize=5, top_max_iter=2)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
out = hba.hierarchical_ba(clouds, poses, coarse, fine, wdsize=10, mgsize=5, top_max_iter=2)
pr.disable()
print('pass seconds', time.perf_counter() - t0)
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
" 2>&1 | grep -v "^$" | head -70 | tee gpurun_out/r5_s8/cfg5_host_profile.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r5_s8/prof_cfg5_trace" -o t -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 2 --warmup 1 > "$GRAFT_REPO_ROOT/gpurun_out/r5_s8/prof_cfg5.log" 2>&1; echo "cfg5 trace rc=$?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/r5_s8/prof_cfg5_trace -type f -name "*_kernel_trace.csv" -size +8M -delete
head -25 gpurun_out/r5_s8/prof_cfg5_trace/*/t_kernel_stats.csv 2>/dev/null | cut -c1-150 || find gpurun_out/r5_s8/prof_cfg5_trace | head
tail -3 gpurun_out/r5_s8/prof_cfg5.log | cut -c1-400
