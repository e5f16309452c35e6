#!/bin/bash
# round 5: SQ / GRBM counters of the two sweeps at cfg2 (separate passes, --kernel-trace only): effective clock under the fp64 load, issue / wait split,
# matrix-core busy cycles, LDS bank conflicts -- beside the stamp-based cycle budget of DESIGN.md 5.1
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python $GRAFT_REPO_ROOT/bench.py --config ${CONFIG:-cfg2} --steps 90 --warmup 9 --repeats 3 --no-cpu-baseline --no-li-ba --no-cold-l3"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > "$GRAFT_REPO_ROOT/gpurun_out/pmc_names.txt"
grep -c . "$GRAFT_REPO_ROOT/gpurun_out/pmc_names.txt"
grep "MFMA\|LDS_BANK\|LDS_IDX\|GUI_ACTIVE\|WAVE_CYCLES\|BUSY_CYCLES\|WAIT_ANY\|WAIT_INST_ANY\|ACTIVE_INST_ANY\|INSTS_VALU\b" "$GRAFT_REPO_ROOT/gpurun_out/pmc_names.txt" | tr '\n' ' '; echo
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_sq_grbm" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/prof_sq_grbm.log" 2>&1; echo "grbm rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_sq_a" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/prof_sq_a.log" 2>&1; echo "sq a rc=$?"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_sq_b" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/prof_sq_b.log" 2>&1; echo "sq b rc=$?"
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, json, collections
out = {}
for tag in ("grbm", "sq_a", "sq_b"):
    for path in glob.glob(f"gpurun_out/prof_sq_{tag}/**/t_counter_collection.csv", recursive=True) + glob.glob(f"gpurun_out/prof_{tag}/**/t_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "k3_hessian" in k or "k2_residual" in k or "k3_finalize" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            out.setdefault(k, {}).update({c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()})
json.dump(out, open("gpurun_out/pmc_sq_counters.json", "w"), indent=1)
for k, cs in out.items():
    print(k[:60], {c: round(v["mean"]) for c, v in cs.items()})
PY
find gpurun_out/prof_sq_* -type f -name "*.csv" -size +4M -delete
tail -3 gpurun_out/prof_sq_b.log | cut -c1-200
