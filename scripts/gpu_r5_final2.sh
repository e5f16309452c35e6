#!/bin/bash
# round 5, closing collection 2: the whole GPU suite + smoke, the bench line as the driver runs it and with the defaults, the configuration table, cfg5
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_final
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" | tail -6 | tee gpurun_out/r5_final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/r5_final/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_final/bench_driver_flags.json 2> gpurun_out/r5_final/bench_driver_flags.err; echo "bench(driver flags) rc=$?"
timeout 900 python bench.py > gpurun_out/r5_final/bench.json 2> gpurun_out/r5_final/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for fn in ("gpurun_out/r5_final/bench_driver_flags.json", "gpurun_out/r5_final/bench.json"):
    d = json.loads(open(fn).read().strip().splitlines()[-1]); r = d["roofline"]
    print(fn, "value %.0f (min %.0f max %.0f) us/step %.2f | K3 %.2f us frac %.3f traffic %s | cold K3 %.2f us frac %.3f | K2 %.2f fin %.2f solve+K2 %.2f" % (
        d["value"], d["repeats"]["value_min"], d["repeats"]["value_max"], 1e3 * d["ms_per_step"], 1e3 * r["avg_launch_ms"], r["frac"], r["traffic"],
        1e3 * r["cold_l3"]["k3_avg_launch_ms"], r["cold_l3"]["frac"], 1e3 * r["k2_residual"]["avg_launch_ms"], 1e3 * r["k3_finalize_avg_ms"], 1e3 * r["solve_plus_k2_launch_avg_ms"]))
    print("   li_ba inside %.4f ms/iter (mirror %.4f) | scan %.3f ms %s | cpu %s %.2f it/s all-cores %s | reject_window %.0f" % (
        d["li_ba"]["ms_per_iteration_inside_the_call"], d["li_ba"]["ms_per_iteration"], d["scan_cycle"]["ms_per_scan"], {k: round(v, 3) for k, v in d["scan_cycle"]["stage_ms"].items()},
        d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("all_cores"), d["reject_window"]["iterations_per_s"]))
PY
bash scripts/gpu_configs.sh 2>&1 | tail -9 | tee gpurun_out/r5_final/config_table.txt
cp gpurun_out/bench_cfg*.json gpurun_out/r5_final/ 2>/dev/null
timeout 900 python bench.py --config cfg5 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r5_final/bench_cfg5.json; cut -c1-400 gpurun_out/r5_final/bench_cfg5.json
VXBA_BENCH_DEVICE=0 VXBA_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r5_final/bench_2_ranks_on_one_gpu_gloo.json; cut -c1-300 gpurun_out/r5_final/bench_2_ranks_on_one_gpu_gloo.json
