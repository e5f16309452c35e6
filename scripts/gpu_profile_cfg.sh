#!/bin/bash
# rocprofv3 of the bench command at one configuration: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in passes of their own.
#   CONFIG=cfg4 STEPS=40 bash scripts/gpu_profile_cfg.sh        -> gpurun_out/prof_${CONFIG}_{trace,fetch,write}/  (scripts/collect_profile_cfg.py files them)
#   CMD="python scripts/dbg_cold_l3.py" TAG=cold bash scripts/gpu_profile_cfg.sh   -> the same passes for another command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
CFG=${CONFIG:-cfg2}
TAG=${TAG:-$CFG}
B=${CMD:-python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps ${STEPS:-90} --warmup 9 --repeats 3 --no-cpu-baseline --no-li-ba --no-cold-l3 ${BENCH_ARGS:-}}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_trace" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_trace.log" 2>&1; echo "$TAG trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_fetch" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_fetch.log" 2>&1; echo "$TAG fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_write" -o t -- $B > "$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_write.log" 2>&1; echo "$TAG write rc=$?"
cd "$GRAFT_REPO_ROOT"
# keep what the collector reads, drop the per-dispatch traces of the counter passes beyond it (gpurun_out is merged back up to 64 MiB)
find gpurun_out/prof_${TAG}_* -type f -name "*_kernel_trace.csv" -size +8M -delete
find gpurun_out/prof_${TAG}_* -type f -name "*_counter_collection.csv" -size +4M -exec python scripts/reduce_counters.py {} \;
du -sh gpurun_out/prof_${TAG}_* 2>/dev/null
