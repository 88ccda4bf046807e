"""Diagnostic (GPU box): BASELINE configs[3] (N = 30, tight tubes, seed 2, 8192 problems) per restoration mode and with / without the second attempt of the
status-2 solves: kernel time (best of 3), converged fraction, iterations.  Usage: python tests/gpu_c3_modes.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, workload
N, B = 30, 8192
P, X, _ = workload.make_batch(B, seed=2, N=N, tight=True)
p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
for mode, cap, mi, retry in ((2, 40, 500, 0), (2, 40, 500, 100), (2, 40, 500, 150), (1, 40, 500, 0), (1, 40, 500, 100)):
    s = BatchedOCPSolver(N, 4, 0.1, max_iter=mi); s.set_restoration(mode, 6, cap); s.set_second_attempt(retry); s.set_timing(1)
    ms = []
    for _ in range(3):
        o = s.solve_batch(p, x0, out={}, want=("iters", "status")); torch.cuda.synchronize(); ms.append(s.last_kernel_ms())
    st, it = o["status"].cpu().numpy(), o["iters"].cpu().numpy()
    print(f"configs[3] restoration mode {mode} cap {cap}, second attempt {retry}: {min(ms):.1f} ms = {B / min(ms) * 1e3:.0f} solves/s, converged {100 * (st == 0).mean():.2f} % ({int((st != 0).sum())} not: "
          f"{dict(zip(*np.unique(st[st != 0], return_counts=True)))}), iterations mean {it.mean():.2f} max {it.max()}", flush=True)
    s.close()
