#!/bin/bash
# Diagnostic (GPU box): instruction-fetch counters of the solver kernel (is the 165 KB kernel text streaming through the 64 KB
# instruction cache?).  Lists the available counters first; each group is its own rocprofv3 --pmc pass.
set -u
TAG=$1; shift
EXTRA="$*"
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmci_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 -L > $OUT/avail.txt 2>&1
grep -o -i -E "\b(SQ_[A-Z_0-9]*IFETCH[A-Z_0-9]*|SQC_ICACHE[A-Z_0-9]*|SQ_INST_LEVEL[A-Z_0-9]*|SQ_WAIT_IFETCH|SQ_INSTS_BRANCH|SQ_BRANCH[A-Z_0-9]*|SQ_VALU_DEP[A-Z_0-9_]*|SQ_INST_CYCLES[A-Z_0-9_]*|SQC_INST[A-Z_0-9_]*|SQ_THREAD_CYCLES_VALU|SQ_ACTIVE_INST_MISC|SQ_INSTS_VSKIPPED|SQ_CYCLES|SQ_LEVEL_WAVES|SQ_ACCUM_PREV)\b" $OUT/avail.txt | sort -u > $OUT/names.txt
cat $OUT/names.txt
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workers 1 $EXTRA > $OUT/p$i.log 2>&1
  echo "pass $i ($grp) exit $?"
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
PY
