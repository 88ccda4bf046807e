#!/bin/bash
# Diagnostic (GPU box): kernel timeline of real-time ticks (rocprofv3 --kernel-trace of tests/gpu_stream_latency.py): durations of the
# pack / solver / post kernels of a tick and the gaps between them.  Usage: bash tests/gpu_stream_trace.sh [K]
ROOT=$GRAFT_REPO_ROOT; K=${1:-4}
OUT=$ROOT/gpurun_out/stream_trace; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tests/gpu_stream_latency.py $K > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, numpy as np
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
# ticks: pack -> (memset kernel?) -> solve -> post
seq = [(n.split("(")[0].split()[-1], s, e) for n, s, e in rows]
names = [n for n, _, _ in seq]
ticks = []
for i in range(len(seq) - 2):
    if "stream_pack" in seq[i][0]:
        j = i + 1
        while j < len(seq) and "solve_kernel" not in seq[j][0] and j < i + 4: j += 1
        if j < len(seq) - 1 and "solve_kernel" in seq[j][0] and "stream_post" in seq[j + 1][0]:
            ticks.append((seq[i], seq[j], seq[j + 1], j - i - 1))
ticks = ticks[3:]
us = lambda a: np.array(a) / 1e3
print(len(ticks), "ticks; kernels between pack and solver:", set(t[3] for t in ticks))
print("pack   median %.1f us" % np.median(us([t[0][2] - t[0][1] for t in ticks])))
print("gap pack->solver median %.1f us" % np.median(us([t[1][1] - t[0][2] for t in ticks])))
print("solver median %.1f us  max %.1f" % (np.median(us([t[1][2] - t[1][1] for t in ticks])), us([t[1][2] - t[1][1] for t in ticks]).max()))
print("gap solver->post median %.1f us" % np.median(us([t[2][1] - t[1][2] for t in ticks])))
print("post   median %.1f us" % np.median(us([t[2][2] - t[2][1] for t in ticks])))
print("pack start -> post end median %.1f us" % np.median(us([t[2][2] - t[0][1] for t in ticks])))
PY
