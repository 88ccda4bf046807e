#!/bin/bash
# Diagnostic (GPU box): PMC counter passes over the bench command, one rocprofv3 run per counter group (no trace domains
# combined with --pmc); stops at the first pass that fails or times out.  Usage: bash tests/gpu_pmc.sh TAG [extra bench.py arguments]
set -u
TAG=$1; shift
EXTRA="$*"
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_FLAT SQ_INSTS_WAVE32_LDS"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --seed-sweep 0 --workers 1 $EXTRA > $OUT/p$i.log 2>&1
  rc=$?
  echo "pass $i ($grp) exit $rc"
  if [ $rc -ne 0 ]; then echo "stopping after failed pass $i"; break; fi
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "bmpc_solve_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
with open("$OUT/summary.txt", "w") as o:
    for k in sorted(tot):
        line = f"{k:32s} per-dispatch {tot[k]/n[k]:.6g}  (dispatches {n[k]})"
        print(line); o.write(line + "\n")
# machine-readable copy for bench.py (profiles/pmc_current.json): per-dispatch means of every counter
import json
json.dump({k: tot[k] / n[k] for k in tot}, open("$OUT/summary.json", "w"), indent=1)
PY
