#!/bin/bash
# Diagnostic (GPU box): PMC counter passes over the bench command, one rocprofv3 run per counter group (no trace domains
# combined with --pmc); stops at the first pass that fails or times out.  Usage: bash tests/gpu_pmc.sh TAG [extra bench.py arguments]
set -u
TAG=$1; shift
EXTRA="$*"
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
# an unprofiled run of the same command first: its kernel time and configuration go into the JSON next to the counters
timeout -k 10 300 python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --seed-sweep 0 --workers 1 $EXTRA > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
c = d["config"]
json.dump({"batch": c["global_batch"], "N": c["horizon"], "tight": "tight tubes" in c["workload"], "kernel_ms": d["roofline"]["kernel_ms"]}, open("$OUT/cfg.json", "w"))
PY
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_FLAT SQ_INSTS_WAVE32_LDS" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --seed-sweep 0 --workers 1 $EXTRA > $OUT/p$i.log 2>&1
  rc=$?
  echo "pass $i ($grp) exit $rc"
  if [ $rc -ne 0 ]; then echo "stopping after failed pass $i"; break; fi
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter(); rtot = collections.defaultdict(float); rn = collections.Counter()
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        # the batch kernel of the configuration: one wave per problem (bmpc_solve_kernel) or teams (bmpc_team_solve_kernel, batches <= 256); the
        # restoration kernel that follows every launch (bmpc_resto_kernel: returns at once when nothing jammed) is counted separately
        if "bmpc_solve_kernel" in r["Kernel_Name"] or "bmpc_team_solve_kernel" in r["Kernel_Name"] or "bmpc_pair_solve_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        elif "bmpc_resto_kernel" in r["Kernel_Name"]:      # the restoration kernel behind every batch launch (bench.py's kernel_ms brackets both): its own entry, added to the traffic below
            rtot[r["Counter_Name"]] += float(r["Counter_Value"]); rn[r["Counter_Name"]] += 1
if not tot:
    raise SystemExit("gpu_pmc.sh: no dispatch of a solver kernel in the counter files -- nothing to summarise")
with open("$OUT/summary.txt", "w") as o:
    for k in sorted(tot):
        line = f"{k:32s} per-dispatch {tot[k]/n[k]:.6g}  (dispatches {n[k]})"
        print(line); o.write(line + "\n")
# machine-readable copy for bench.py (profiles/pmc_current.json): per-dispatch means of every counter, tied to the kernel text they
# were taken on (bench.kernel_text_hash = boundmpc_amd.build.source_hash: every source file + the compiler flags) and to the configuration of the bench command
import json, sys
sys.path.insert(0, "$ROOT")
import bench
rec = {k: tot[k] / n[k] for k in tot}
json.dump(rec, open("$OUT/summary.json", "w"), indent=1)
cfg = json.load(open("$OUT/cfg.json")) if __import__("os").path.exists("$OUT/cfg.json") else {}
cur = {"kernel_hash": bench.kernel_text_hash(), "batch": cfg.get("batch", 1024), "N": cfg.get("N", 10), "tight": cfg.get("tight", False)}
rrec = {k: rtot[k] / rn[k] for k in rtot}
if "FETCH_SIZE" in rec: cur["FETCH_SIZE_KB"] = rec["FETCH_SIZE"] + rrec.get("FETCH_SIZE", 0.0)      # batch kernel + the restoration kernel behind it (what kernel_ms brackets)
if "WRITE_SIZE" in rec: cur["WRITE_SIZE_KB"] = rec["WRITE_SIZE"] + rrec.get("WRITE_SIZE", 0.0)
cur["restoration_kernel"] = {k: rrec[k] for k in ("FETCH_SIZE", "WRITE_SIZE") if k in rrec}
try:
    cur["iterations_of_the_launch"] = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])["config"]["mean_iters"] * cur["batch"]
except Exception:
    pass
for k in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
          "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
          "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_THREAD_CYCLES_VALU"):
    if k in rec: cur[k] = rec[k]
cur["kernel_ms"] = cfg.get("kernel_ms")
cur["source"] = "tests/gpu_pmc.sh $TAG: separate rocprofv3 --pmc passes of python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --seed-sweep 0 --workers 1 $EXTRA; kernel_ms from an unprofiled bench run of the same call"
cur["note"] = ("FETCH_SIZE + WRITE_SIZE (KB x 1024) per launch, separate --pmc passes; raw counter values: the kernel's 8-B-per-lane accesses are an uncalibrated "
               "width in the guide (no x2 correction applied); this is per-wave workspace traffic between L2 and Infinity Cache/HBM, not input re-reads")
json.dump(cur, open("$OUT/pmc_current.json", "w"), indent=1)
print("wrote $OUT/pmc_current.json (copy to profiles/ when this is the kernel of the round):", cur["kernel_hash"])
PY
