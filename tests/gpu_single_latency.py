"""Diagnostic (GPU box): wall-clock latency of the drop-in single-problem call solver(x0=, ..., p=) (NlpSolverShim -> bmpc_solve_batch_host:
H2D copy, one-problem launch, D2H copies) on the recorded experiment1 closed-loop ticks, next to the CPU oracle on one thread."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boundmpc_amd import BatchedOCPSolver, NlpSolverShim
from oracle import c_oracle
d = np.load(os.path.join(ROOT, "tests", "golden", "g7_closedloop_exp1.npz"))
s = BatchedOCPSolver(10, 4, 0.1); shim = NlpSolverShim(s)
for warm in range(3):
    shim(x0=d["x0"][0], p=d["p"][0])
tg, tc, it = [], [], []
for t in range(0, 155, 3):
    t0 = time.perf_counter(); sol = shim(x0=d["x0"][t], p=d["p"][t]); tg.append(time.perf_counter() - t0); it.append(shim.stats()["iter_count"])
    t0 = time.perf_counter(); c_oracle.solve(d["p"][t], d["x0"][t], 10, 4, 0.1, nthreads=1); tc.append(time.perf_counter() - t0)
tg, tc = np.array(tg) * 1e3, np.array(tc) * 1e3
print(f"single-problem call over {len(tg)} recorded ticks (mean {np.mean(it):.1f} iterations): GPU shim p50 {np.percentile(tg,50):.2f} ms p99 {np.percentile(tg,99):.2f} ms; "
      f"CPU oracle (1 thread) p50 {np.percentile(tc,50):.2f} ms p99 {np.percentile(tc,99):.2f} ms")
