#!/bin/bash
# round 4, final 4: rocprofv3 evidence re-collected from the closing kernel sources (the stamp in pmc_hbm_counters.json must be the hash of the tree's sources), bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
CONFIG=cfg2 STEPS=90 bash scripts/gpu_profile_cfg.sh | tail -3
CONFIG=cfg3 STEPS=60 bash scripts/gpu_profile_cfg.sh | tail -3
CONFIG=cfg4 STEPS=30 bash scripts/gpu_profile_cfg.sh | tail -3
TAG=cold CMD="python $GRAFT_REPO_ROOT/scripts/dbg_cold_l3.py cfg2" bash scripts/gpu_profile_cfg.sh | tail -3
# file the summaries here as well (the same command is run on the tracked tree afterwards): the bench lines below then carry roofline.traffic
for c in cfg2 cfg3 cfg4; do python scripts/collect_profile_cfg.py r04_$c $c > /dev/null; done
python scripts/collect_profile_cfg.py r04_cfg2_cold_l3 cfg2 cold > /dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_final_bench_driver_flags.json 2> gpurun_out/r4_final_bench_driver_flags.err; echo "bench(driver flags) rc=$?"
timeout 900 python bench.py > gpurun_out/r4_final_bench.json 2> gpurun_out/r4_final_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for fn in ("gpurun_out/r4_final_bench_driver_flags.json", "gpurun_out/r4_final_bench.json"):
    d = json.loads(open(fn).read().strip().splitlines()[-1]); r = d["roofline"]
    print(fn, "value %.0f (min %.0f max %.0f) us/step %.2f | K3 %.2f us frac %.3f traffic %s | cold K3 %.2f us frac %.3f | li %.4f | scan %.3f" % (
        d["value"], d["repeats"]["value_min"], d["repeats"]["value_max"], 1e3 * d["ms_per_step"], 1e3 * r["avg_launch_ms"], r["frac"], r["traffic"],
        1e3 * r["cold_l3"]["k3_avg_launch_ms"], r["cold_l3"]["frac"], d["li_ba"]["ms_per_iteration_inside_the_call"], d["scan_cycle"]["ms_per_scan"]))
PY
