#!/bin/bash
# round 6, closing collection 2: whole GPU suite + smoke, the bench lines (driver flags, defaults, every configuration, cfg5 with its CPU baseline, one thread), the
# pieces of the --gpus N prediction, two ranks on one GPU, the randomised sweep
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_final
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" | tail -6 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $O/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench(driver flags) rc=$?"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for fn in ("gpurun_out/r6_final/bench_driver_flags.json", "gpurun_out/r6_final/bench.json"):
    d = json.loads(open(fn).read().strip().splitlines()[-1]); r = d["roofline"]
    print(fn, "value %.0f (min %.0f max %.0f) us/step %.2f (instrumented %.2f) | %s %.2f us frac %.3f traffic %s | li_ba %.4f | scan %.3f %s" % (
        d["value"], d["repeats"]["value_min"], d["repeats"]["value_max"], 1e3 * d["ms_per_step"], 1e3 * d["repeats"]["instrumented"]["ms_per_step"], r["kernel"], 1e3 * r["avg_launch_ms"], r["frac"], r["traffic"],
        d["li_ba"]["ms_per_iteration_inside_the_call"], d["scan_cycle"]["ms_per_scan"], {k: round(v, 3) for k, v in d["scan_cycle"]["stage_ms"].items()}))
PY
bash scripts/gpu_configs.sh 2>&1 | tail -9 | tee $O/config_table.txt
cp gpurun_out/bench_cfg*.json $O/ 2>/dev/null
timeout 900 python bench.py --config cfg5 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_cfg5.json; cut -c1-330 $O/bench_cfg5.json; echo
timeout 900 python bench.py --config cfg5 --steps 3 --warmup 1 --hba-threads 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg5_one_thread.json; cut -c200-330 $O/bench_cfg5_one_thread.json; echo
timeout 600 python scripts/dbg_scaling_pieces.py 2>&1 | grep -v amdgpu.ids | tee $O/scaling_pieces.txt
timeout 300 python scripts/dbg_profiling_cost.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/profiling_cost.txt
VXBA_BENCH_DEVICE=0 VXBA_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_2_ranks_on_one_gpu_gloo.json; cut -c1-200 $O/bench_2_ranks_on_one_gpu_gloo.json; echo
VXBA_BENCH_DEVICE=0 VXBA_BENCH_BACKEND=gloo timeout 900 python bench.py --config cfg5 --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg5_2_ranks_on_one_gpu_gloo.json; cut -c1-200 $O/bench_cfg5_2_ranks_on_one_gpu_gloo.json; echo
: > $O/fuzz_r6.log
timeout 1500 python scripts/fuzz_parity.py 611 300 2>&1 | grep -v amdgpu | grep -E "MISMATCH|cases" | tail -6 >> $O/fuzz_r6.log
FUZZ_KINDS=hba timeout 900 python scripts/fuzz_parity.py 641 60 2>&1 | grep -v amdgpu | grep -E "MISMATCH|cases" | tail -3 >> $O/fuzz_r6.log
cat $O/fuzz_r6.log
