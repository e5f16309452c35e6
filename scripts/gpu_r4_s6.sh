#!/bin/bash
# round 4, session 6: voxel-sharded voxelisation + the multi-rank hierarchical BA (ranks = processes on one GPU, gloo), cfg5 bench line on 1 and 2 ranks
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_two_rank.py tests/test_gpu_hba.py tests/test_gpu_voxelize.py tests/test_gpu_wide.py tests/test_gpu_li_ba.py tests/test_gpu_map.py -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -8
timeout 900 python bench.py --config cfg5 --steps 2 --warmup 1 2> gpurun_out/r4_s6_cfg5_n1.err | tee gpurun_out/r4_s6_cfg5_n1.json | cut -c1-900; tail -2 gpurun_out/r4_s6_cfg5_n1.err
VXBA_BENCH_BACKEND=gloo VXBA_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --config cfg5 --steps 2 --warmup 1 2> gpurun_out/r4_s6_cfg5_n2.err | tee gpurun_out/r4_s6_cfg5_n2.json | cut -c1-900; tail -3 gpurun_out/r4_s6_cfg5_n2.err
