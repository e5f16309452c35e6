#!/bin/bash
# round 4, final 1: the whole GPU suite (+ the cfg5-size oracle comparison), smoke, the driver's bench line, the configuration table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
VXBA_RUN_SLOW=1 timeout 1800 python -m pytest tests/test_gpu_hba.py -m gpu -q -s -k cfg5_size --timeout 1700 -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r4_final_cfg5_size_test.txt; cat gpurun_out/r4_final_cfg5_size_test.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_final_bench_driver_flags.json 2> gpurun_out/r4_final_bench_driver_flags.err; echo "bench(driver flags) rc=$?"
timeout 900 python bench.py > gpurun_out/r4_final_bench.json 2> gpurun_out/r4_final_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for fn in ("gpurun_out/r4_final_bench_driver_flags.json", "gpurun_out/r4_final_bench.json"):
    d = json.loads(open(fn).read().strip().splitlines()[-1]); r = d["roofline"]
    print(fn, "value %.0f (min %.0f max %.0f) us/step %.2f | K3 %.2f us frac %.3f traffic %s | cold K3 %.2f us frac %.3f | K2 %.2f fin %.2f solve+K2 %.2f" % (
        d["value"], d["repeats"]["value_min"], d["repeats"]["value_max"], 1e3 * d["ms_per_step"], 1e3 * r["avg_launch_ms"], r["frac"], r["traffic"],
        1e3 * r["cold_l3"]["k3_avg_launch_ms"], r["cold_l3"]["frac"], 1e3 * r["k2_residual"]["avg_launch_ms"], 1e3 * r["k3_finalize_avg_ms"], 1e3 * r["solve_plus_k2_launch_avg_ms"]))
    print("   li_ba inside %.4f ms/iter (mirror %.4f) | scan %.3f ms %s | cpu %s %.2f it/s all-cores %s | reject_window %.0f" % (
        d["li_ba"]["ms_per_iteration_inside_the_call"], d["li_ba"]["ms_per_iteration"], d["scan_cycle"]["ms_per_scan"], {k: round(v, 3) for k, v in d["scan_cycle"]["stage_ms"].items()},
        d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("all_cores"), d["reject_window"]["iterations_per_s"]))
PY
bash scripts/gpu_configs.sh 2>&1 | tail -9
