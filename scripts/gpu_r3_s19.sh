#!/bin/bash
# round 3, session 19: the sharded paths with one rank (plumbing): RCCL direct, torch.distributed hook, peer mailboxes refuse politely; torchrun N=1
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
p() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    c = d['config']
    print('$1: it/s %.0f us/step %.2f collective %s per step %s allreduce_us %s ranks %s scaling %s' % (d['value'], 1e3*d['ms_per_step'], c.get('collective_used'), c.get('collectives_per_lm_step'), c.get('allreduce_us_avg'), c.get('ranks_seen'), d['scaling']))
"; }
timeout 300 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-li-ba --force-dist 2>gpurun_out/fd1.err | p force-dist-rccl; tail -2 gpurun_out/fd1.err | grep -i "error\|Traceback"
timeout 300 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-li-ba --force-dist --hook-allreduce 2>gpurun_out/fd2.err | p force-dist-hook; tail -2 gpurun_out/fd2.err | grep -i "error\|Traceback"
timeout 300 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-li-ba --force-dist --scaling strong 2>gpurun_out/fd3.err | p force-dist-strong; tail -2 gpurun_out/fd3.err | grep -i "error\|Traceback"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 150 --warmup 15 --no-cpu-baseline --no-li-ba 2>gpurun_out/fd4.err | p torchrun-n1; tail -2 gpurun_out/fd4.err | grep -i "error\|Traceback"
