"""Diagnostic (GPU box): per-problem comparison of the N=30 tight sample of test_long_horizon_tight_tubes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boundmpc_amd import BatchedOCPSolver, workload
from oracle import c_oracle
P, X, _ = workload.make_batch(64, seed=2, N=30, tight=True)
s = BatchedOCPSolver(30, 4, 0.1)
o = s.solve_host(P, X)
r = c_oracle.solve(P, X, 30, 4, 0.1)
d = (o["x"] - r["x"]).reshape(64, 30, 44)[:, :, 8:15]
per = np.sqrt((d ** 2).mean(axis=(1, 2)))
for b in np.argsort(-per)[:6]:
    print(b, "rms", per[b], "iters", o["iters"][b], r["iters"][b], "status", o["status"][b], r["status"][b], "f", o["f"][b], r["f"][b], "kkt", o["kkt"][b], r["kkt"][b])
