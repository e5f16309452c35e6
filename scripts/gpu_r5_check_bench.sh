#!/bin/bash
# round 5: the bench line with its `host` object, and the self-launched 2-rank runs as the tests type them
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py --steps 20 --warmup 5 --no-li-ba --no-cold-l3 --cpu-seconds 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['host'], d['cpu_baseline'].get('all_cores'))"
timeout 1500 python -m pytest tests/test_gpu_two_rank.py -x -q -p no:cacheprovider 2>&1 | tail -3
