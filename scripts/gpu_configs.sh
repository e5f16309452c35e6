#!/bin/bash
# the bench line for every single-GPU configuration (cfg4 = 8 x cfg2 is the multi-GPU one; it also fits one GPU)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in cfg1 cfg2 cfg2_sparse cfg2_fix cfg3 cfg4; do
  timeout 600 python bench.py --config $c --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | tail -1 > gpurun_out/bench_$c.json
  python3 - $c <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_%s.json" % sys.argv[1]))
r = d["roofline"]
print("%-12s %8.0f it/s  %.4f ms/step  K3 %.1f us (%.1f%% hbm)  K2 %.1f us (%.1f%% hbm)  accepted %s" % (sys.argv[1], d["value"], d["ms_per_step"], 1e3 * r["avg_launch_ms"], 100 * r["frac"],
      1e3 * r["k2_residual"]["avg_launch_ms"], 100 * r["k2_residual"]["frac"], d["config"].get("accepted_steps")))
PY
done
timeout 600 python bench.py --config cfg3 --precision mixed --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | tail -1 > gpurun_out/bench_cfg3_mixed.json
python3 -c "
import json; d = json.load(open('gpurun_out/bench_cfg3_mixed.json')); print('cfg3 mixed  %8.0f it/s  K3 %.1f us  K2 %.1f us' % (d['value'], 1e3 * d['roofline']['avg_launch_ms'], 1e3 * d['roofline']['k2_residual']['avg_launch_ms']))"
timeout 600 python bench.py --config cfg3 --precision mixed_f32_clusters --no-cpu-baseline --no-li-ba --no-cold-l3 2>/dev/null | tail -1 > gpurun_out/bench_cfg3_mixed_f32_clusters.json
python -c "
import json; d = json.load(open('gpurun_out/bench_cfg3_mixed_f32_clusters.json')); r = d['roofline']['k2_residual']; print('cfg3 mixed + f32 cluster rows  %8.0f it/s  K3 %.1f us  K2 %.1f us (%.0f %% of the HBM roofline on %.1f MB)' % (d['value'], 1e3 * d['roofline']['avg_launch_ms'], 1e3 * r['avg_launch_ms'], 100 * r['frac'], r['algorithmic_bytes_per_launch'] / 1e6))"
