#!/bin/bash
# Diagnostic (GPU box): the closed loops of BASELINE configs[4] with and without the restoration phase.  Usage: bash tests/gpu_stream_round.sh TAG
TAG=${1:-r05_x}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 500 python bench_stream.py --only "warm,budget800us,budget700us" > gpurun_out/${TAG}_stream_default.json 2> gpurun_out/${TAG}_stream_default.err || { echo "default run failed"; tail -5 gpurun_out/${TAG}_stream_default.err; exit 1; }
timeout -k 10 300 python bench_stream.py --only "warm" --no-restoration > gpurun_out/${TAG}_stream_noresto.json 2>/dev/null || { echo "no-restoration run failed"; exit 1; }
timeout -k 10 300 python bench_stream.py --only "warm" --resto-cap 40 > gpurun_out/${TAG}_stream_cap40.json 2>/dev/null || { echo "cap-40 run failed"; exit 1; }
timeout -k 10 400 python bench_stream.py --only "warm" --resto-cap 150 --max-iter 200 > gpurun_out/${TAG}_stream_cap150.json 2>/dev/null || { echo "cap-150 run failed"; exit 1; }
python - <<PY
import json
for f in ("default", "noresto", "cap40", "cap150"):
    d = json.loads(open(f"gpurun_out/${TAG}_stream_{f}.json").read().strip().splitlines()[-1])
    print("==", f, "value", d["value"], d["mode_reported"])
    for r in d["results"]:
        t = r["tube_compliance_of_the_measured_states"]
        print("  %-48s p50 %.2f p99 %.2f ms | its %.1f slowest %s | applied %.3f plan %.3f | tube pos %.2e (max %.1e m) rot %.2e (max %.1e rad) streams %d | skipped %d | within 1e-2 rad %.2f"
              % (r["mode"][:48], r["tick_ms_p50"], r["tick_ms_p99"], r["mean_iters"], list(r["slowest_stream_iterations_per_tick"].values()), r["applied_tick_fraction"],
                 r["streams_with_a_plan_at_the_end"], t["fraction_outside_the_position_tube"], t["largest_position_excess_m"], t["fraction_outside_the_orientation_tube"],
                 t["largest_orientation_excess_rad"], t["streams_ever_outside_a_tube"], r["streams_skipped_at_the_end"], r["streams_within_1e-2_rad_rms"]))
        a = t["applied_plans_first_stage_rows_reference_form"]
        print("      applied plans %d: position rows > 1e-6: %.2e (max %.1e m^2), orientation rows > 1e-6: %.2e (max %.1e rad^2)" % (a["applied_plans"], a["fraction_with_a_position_row_above_1e-6"],
              a["largest_position_row_m2"], a["fraction_with_an_orientation_row_above_1e-6"], a["largest_orientation_row_rad2"]))
PY
