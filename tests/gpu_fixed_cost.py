"""Diagnostic (GPU box): launch time of the solver kernel against the iteration cap K (K = 0 is evaluation only): fixed cost and
per-iteration cost of a launch at a given batch size.  Usage: python tests/gpu_fixed_cost.py [one] [B ...]   (one: one wave per problem at every batch size,
instead of the library's choice -- teams for B <= 256)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, workload

ONE = "one" in sys.argv[1:]
for B in [int(a) for a in sys.argv[1:] if a != "one"] or [256, 1024]:
    P, X, _ = workload.make_batch(B, seed=3, N=10)
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    for K in (0, 1, 2, 3, 4, 6, 8, 12, 14, 16, 20):
        s = BatchedOCPSolver(10, 4, 0.1, max_iter=K)
        if ONE: s.set_team_waves(1)
        for _ in range(3): s.solve_batch(p, x0)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); o = s.solve_batch(p, x0); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"B={B} K={K}: solve_batch p50 {np.median(ts)*1e3:.0f} us  min {min(ts)*1e3:.0f} us  mean iters {o['iters'].float().mean().item():.2f}", flush=True)
        s.close()
