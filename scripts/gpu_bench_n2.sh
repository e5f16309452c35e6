#!/bin/bash
# bench.py on two process ranks sharing ONE GPU (gloo; RCCL refuses two ranks on a device): the multi-rank code path of the bench, both scalings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
p=29640
for sc in weak strong; do
  p=$((p + 1))
  VXBA_BENCH_BACKEND=gloo VXBA_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --steps 20 --warmup 5 --scaling $sc 2> gpurun_out/r4_bench_n2_$sc.err | tail -1 > gpurun_out/r4_bench_n2_$sc.json
  python - $sc <<'PY'
import json, sys
sc = sys.argv[1]
d = json.loads(open("gpurun_out/r4_bench_n2_%s.json" % sc).read())
print(sc, "value %.0f window it/s %.0f repeats %d collective: %s, %s per LM step, ranks seen %s, scaling %s, allreduce %.1f us" % (
    d["value"], d["window_iterations_per_s"], d["repeats"]["n"], d["config"]["collective_used"], d["config"]["collectives_per_lm_step"], d["config"]["ranks_seen"], d["scaling"],
    d["config"]["allreduce_us_avg"] or -1))
PY
  grep -v "amdgpu.ids\|Gloo\|c10d\|^$" gpurun_out/r4_bench_n2_$sc.err | tail -3
done
