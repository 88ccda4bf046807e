#!/bin/bash
# Diagnostic (GPU box): A/B timing of alternative builds of the solver library (build/ab*/lib_*.so, BOUNDMPC_HIP_LIB) on the bench
# configurations.  Usage: bash tests/gpu_ab.sh DIR
cd "${GRAFT_REPO_ROOT:-.}"
for lib in ${1:-build/ab3}/*.so; do
  echo "== $lib"
  for cfg in "--steps 50" "--config 2 --steps 20" "--config 3 --steps 3 --warmup 1"; do
    BOUNDMPC_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', '$cfg', round(r['value']), 'kernel_ms %.3f'%r['roofline']['kernel_ms'], 'max its', r['config']['max_iters'], 'ok %.3f'%r['config']['solved_fraction'])"
  done
done
