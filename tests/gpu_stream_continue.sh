#!/bin/bash
# Diagnostic (GPU box): converged closed loops of configs[4] with a hard iteration limit per tick and continuation across ticks (a tick that ends
# unconverged -- main phase or restoration phase -- hands its iterate to the next tick's warm start): does spreading the long solves of the streams
# that are in trouble over several ticks keep their plans at a bounded tick time?  Usage: bash tests/gpu_stream_continue.sh TAG
TAG=${1:-r05_x}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for mi in ${MIS:-36 45 60}; do for cap in ${CAPS:-16 24}; do
  timeout -k 10 300 python bench_stream.py --only "warm" --max-iter $mi --resto-cap $cap > gpurun_out/${TAG}_stream_mi${mi}_cap${cap}.json 2>/dev/null || { echo "run $mi $cap failed"; continue; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_stream_mi${mi}_cap${cap}.json").read().strip().splitlines()[-1])
for r in d["results"]:
    print("max_iter $mi cap $cap  %-40s p50 %.2f p99 %.2f ms | its %.1f slowest %s | applied %.3f plan %.3f" % (r["mode"][:40], r["tick_ms_p50"], r["tick_ms_p99"], r["mean_iters"],
          list(r["slowest_stream_iterations_per_tick"].values()), r["applied_tick_fraction"], r["streams_with_a_plan_at_the_end"]))
PY
done; done
