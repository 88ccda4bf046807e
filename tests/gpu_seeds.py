"""Diagnostic (GPU box): kernel time of the 1024-problem launch for the random batches of seeds 0..11 (bench.py times seed 0)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, workload
s = BatchedOCPSolver(10, 4, 0.1); s.set_timing(20)
rows = []
for seed in range(12):
    P, X, _ = workload.make_batch(1024, seed=seed, N=10)
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    for _ in range(3): o = s.solve_batch(p, x0)
    torch.cuda.synchronize()
    for _ in range(20): o = s.solve_batch(p, x0)
    torch.cuda.synchronize()
    ms = np.mean([s.kernel_ms(i) for i in range(20)])
    it = o["iters"].cpu().numpy(); ok = float((o["status"] == 0).double().mean().item())
    rows.append(ms)
    print(f"seed {seed:2d}: kernel {ms:.3f} ms = {1024/ms:.0f} k solves/s, iterations mean {it.mean():.2f} max {it.max()}, converged {ok:.4f}", flush=True)
print(f"seeds 0..11: kernel time min {min(rows):.2f} / median {np.median(rows):.2f} / max {max(rows):.2f} ms")
