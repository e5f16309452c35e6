#!/bin/bash
# VXBA_PRECISION_MIXED_F32_CLUSTERS: its parity tests, then cfg3 in the three precision modes (K2 / K3 times from the bench line)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 200 -k "f32 or mixed" 2>&1 | tail -8
for p in f64 mixed mixed_f32_clusters; do
  timeout 300 python bench.py --config cfg3 --precision $p --no-cpu-baseline --no-li-ba 2>/dev/null | tail -1 > gpurun_out/bench_cfg3_$p.json
  python3 - $p <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_cfg3_%s.json" % sys.argv[1])); r = d["roofline"]; k = r["k2_residual"]
print("cfg3 %-20s %7.0f it/s  %.1f us/step  K3 %.1f us  K2 %.1f us alone (%.1f MB, %.0f %% of the HBM roofline)  solve+K2 %.1f us  accepted %s rejected %s  residual %.9g" % (
    sys.argv[1], d["value"], 1e3 * d["ms_per_step"], 1e3 * r["avg_launch_ms"], 1e3 * k["avg_launch_ms"], k["algorithmic_bytes_per_launch"] / 1e6, 100 * k["frac"],
    1e3 * r.get("solve_plus_k2_launch_avg_ms", 0), d["config"]["lm_steps_accepted"], d["config"]["lm_steps_rejected"], d["config"]["final_residual"]))
PY
done
