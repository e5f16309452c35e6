#!/bin/bash
# round 5, closing collection 3 (after the hierarchical pass went to four polled streams): whole GPU suite + smoke, the bench lines (driver flags, defaults, cfg5 with its CPU
# baseline, one and four threads), rocprofv3 of the cfg5 pass, two ranks on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_final3
O=gpurun_out/r5_final3
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | grep -v "RuntimeWarning\|ev_ref\|^$\|Docs:\|warnings.warn" | tail -6 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $O/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench(driver flags) rc=$?"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for fn in ("gpurun_out/r5_final3/bench_driver_flags.json", "gpurun_out/r5_final3/bench.json"):
    d = json.loads(open(fn).read().strip().splitlines()[-1]); r = d["roofline"]
    print(fn, "value %.0f (min %.0f max %.0f) us/step %.2f | K3 %.2f us frac %.3f traffic %s | li_ba %.4f | scan %.3f" % (
        d["value"], d["repeats"]["value_min"], d["repeats"]["value_max"], 1e3 * d["ms_per_step"], 1e3 * r["avg_launch_ms"], r["frac"], r["traffic"],
        d["li_ba"]["ms_per_iteration_inside_the_call"], d["scan_cycle"]["ms_per_scan"]))
PY
timeout 900 python bench.py --config cfg5 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_cfg5.json; cut -c1-330 $O/bench_cfg5.json; echo
timeout 900 python bench.py --config cfg5 --steps 3 --warmup 1 --hba-threads 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg5_one_thread.json; cut -c200-330 $O/bench_cfg5_one_thread.json; echo
python -c "import json; [print(f, json.loads(open('$O/'+f).read().strip().splitlines()[-1])['host']) for f in ('bench_cfg5.json','bench_cfg5_one_thread.json')]"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg5_r5f3 -o t -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rocprof_cfg5.log 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof_cfg5_r5f3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cfg5_kernel_stats.csv
find gpurun_out/prof_cfg5_r5f3 -type f ! -name "*stats.csv" -delete 2>/dev/null
head -8 $O/cfg5_kernel_stats.csv | cut -c1-60,200-300
VXBA_BENCH_DEVICE=0 VXBA_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_2_ranks_on_one_gpu_gloo.json; cut -c1-200 $O/bench_2_ranks_on_one_gpu_gloo.json
# the randomised sweep on what changed last (voxeliser read-backs, the pass on 1 .. 8 threads) and once over every kind
: > $O/fuzz_r5_close.log
FUZZ_KINDS=hba timeout 1500 python scripts/fuzz_parity.py 341 160 2>&1 | grep -v amdgpu | grep -E "MISMATCH|cases" | tail -6 >> $O/fuzz_r5_close.log
timeout 900 python scripts/fuzz_parity.py 351 260 2>&1 | grep -v amdgpu | tail -1 >> $O/fuzz_r5_close.log
cat $O/fuzz_r5_close.log
