#!/bin/bash
# round 6, step 7: two process ranks on one GPU (gloo) through dist.hba_pass at the bench's session size -- where do 23 s per pass go?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
HBA_K=${HBA_K:-500} HBA_WD=10 HBA_MG=5 HBA_PTS=${HBA_PTS:-20000} HBA_THREADS=2 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 scripts/dbg_two_rank_hba.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r6_s7_two_rank_hba.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_launch_is" --timeout 600 -p no:cacheprovider 2>&1 | grep -E "^E|assert|passed|failed" | head -20 | tee gpurun_out/r6_s7_fused_test.txt
