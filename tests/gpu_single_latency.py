"""Diagnostic (GPU box): wall-clock latency of the drop-in single-problem call solver(x0=, ..., p=) (NlpSolverShim -> bmpc_solve_batch_host:
H2D copy, one-problem launch, D2H copies) on the recorded experiment1 closed-loop ticks, next to the CPU oracle on one thread."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boundmpc_amd import BatchedOCPSolver, NlpSolverShim
from oracle import c_oracle
d = dict(np.load(os.path.join(ROOT, "tests", "golden", "g7_closedloop_exp1.npz")))   # materialise: an NpzFile re-reads the archive on every access
s = BatchedOCPSolver(10, 4, 0.1); shim = NlpSolverShim(s)
s.set_timing(True)
for warm in range(3):
    shim(x0=d["x0"][0], p=d["p"][0])
tg, tc, it, tk, tgi = [], [], [], [], []
ticks = list(range(0, 155, 3))
for t in ticks:      # back-to-back calls
    t0 = time.perf_counter(); sol = shim(x0=d["x0"][t], p=d["p"][t]); tg.append(time.perf_counter() - t0); it.append(shim.stats()["iter_count"]); tk.append(s.last_kernel_ms())
for t in ticks:      # the same calls with the device idle in between (the CPU oracle solves the tick on one thread: ~10 ms)
    t0 = time.perf_counter(); c_oracle.solve(d["p"][t], d["x0"][t], 10, 4, 0.1, nthreads=1); tc.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); sol = shim(x0=d["x0"][t], p=d["p"][t]); tgi.append(time.perf_counter() - t0)
tg, tc, tgi = np.array(tg) * 1e3, np.array(tc) * 1e3, np.array(tgi) * 1e3
print(f"kernel time of the one-problem launch (HIP events): p50 {np.percentile(tk,50):.2f} ms p99 {np.percentile(tk,99):.2f} ms")
print(f"single-problem call over {len(tg)} recorded ticks (mean {np.mean(it):.1f} iterations): GPU shim back-to-back p50 {np.percentile(tg,50):.2f} ms p99 {np.percentile(tg,99):.2f} ms; "
      f"after ~10 ms of device idle p50 {np.percentile(tgi,50):.2f} ms p99 {np.percentile(tgi,99):.2f} ms; CPU oracle (1 thread) p50 {np.percentile(tc,50):.2f} ms p99 {np.percentile(tc,99):.2f} ms")

# the same calls at the REFERENCE's own convergence tolerance (ipopt 'tol': 10e-6, BoundMPC.py:121) -- the drop-in's default is 1e-8
s5 = BatchedOCPSolver(10, 4, 0.1, tol=1e-5); shim5 = NlpSolverShim(s5); s5.set_timing(True)
for warm in range(3):
    shim5(x0=d["x0"][0], p=d["p"][0])
t5, it5, k5 = [], [], []
for t in ticks:
    t0 = time.perf_counter(); shim5(x0=d["x0"][t], p=d["p"][t]); t5.append(time.perf_counter() - t0); it5.append(shim5.stats()["iter_count"]); k5.append(s5.last_kernel_ms())
t5 = np.array(t5) * 1e3
print(f"at the reference's tolerance 1e-5 (mean {np.mean(it5):.1f} iterations): call p50 {np.percentile(t5,50):.2f} ms p99 {np.percentile(t5,99):.2f} ms, kernel p50 {np.percentile(k5,50):.2f} ms; "
      f"waves per problem: {s.team_info(1)['waves']}")
