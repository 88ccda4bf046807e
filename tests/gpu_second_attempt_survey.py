"""Diagnostic (GPU box): the second attempt of long-horizon solves (bmpc_set_second_attempt) on batches it was NOT found on: converged fraction and kernel
time with the rule (the default) and without, for several horizons, tube widths and seeds.  Usage: python tests/gpu_second_attempt_survey.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, workload
for N, tight, seed, B in ((30, True, 5, 2048), (30, True, 6, 2048), (30, True, 7, 4096), (30, False, 9, 4096), (20, True, 26, 4096), (20, False, 7, 2048), (16, True, 8, 2048), (36, True, 3, 1024), (40, False, 9, 1024), (12, True, 4, 2048)):
    P, X, _ = workload.make_batch(B, seed=seed, N=N, tight=tight)
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    s = BatchedOCPSolver(N, 4, 0.1); s.set_timing(1)
    res = []
    for cap in (0, 100):
        s.set_second_attempt(cap); ms = []
        for _ in range(2):
            o = s.solve_batch(p, x0, out={}, want=("iters", "status")); torch.cuda.synchronize(); ms.append(s.last_kernel_ms())
        st, it = o["status"].cpu().numpy(), o["iters"].cpu().numpy()
        res.append((min(ms), int((st != 0).sum()), int((st == 2).sum()), int(it.max()), it.mean()))
    a, b = res
    print(f"N={N:2d} tight={tight!s:5} seed={seed:2d} B={B}: first attempt alone {a[0]:7.1f} ms, not converged {a[1]:3d} (status 2: {a[2]:3d}), slowest {a[3]:3d} | with the second attempt {b[0]:7.1f} ms ({b[0] / a[0]:.2f}x), "
          f"not converged {b[1]:3d} (status 2: {b[2]:3d}), slowest {b[3]:3d}, mean iterations {a[4]:.2f} -> {b[4]:.2f}", flush=True)
    s.close()
