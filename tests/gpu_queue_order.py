"""Diagnostic (GPU box): the work-queue order of batches that take several rounds of the resident waves (bmpc_set_queue_order: natural order vs
longest-expected-first by the objective at x0) on BASELINE configs[3] (N = 30 tight, B = 8192) and on the configs[2] shard (N = 10, B = 8192):
kernel time (evaluation pass and ranking included), bitwise comparison of the results.  Usage: python tests/gpu_queue_order.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from boundmpc_amd import BatchedOCPSolver, workload  # noqa: E402

for name, N, tight, seed, B in (("configs[3]", 30, True, 2, 8192), ("configs[2] shard 0", 10, False, 1, 8192), ("N=20 tight", 20, True, 26, 4096)):
    P, X, _ = workload.make_batch(B * (8 if name.startswith("configs[2]") else 1), seed=seed, N=N, tight=tight, rows=(0, B))
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    s = BatchedOCPSolver(N, 4, 0.1)
    s.set_timing(1)
    res = {}
    for mode in (0, 1):
        s.set_queue_order(mode)
        ms = []
        for _ in range(3):
            o = s.solve_batch(p, x0, out={}, want=("iters", "status")); torch.cuda.synchronize(); ms.append(s.last_kernel_ms())
        res[mode] = (min(ms), o["x"].clone(), o["iters"].cpu().numpy(), o["status"].cpu().numpy())
    it, st = res[1][2], res[1][3]
    print(f"{name}: B={B} natural order {res[0][0]:.2f} ms ({B / res[0][0] * 1e3:.0f} solves/s), by decreasing f(x0) {res[1][0]:.2f} ms ({B / res[1][0] * 1e3:.0f} solves/s): {res[0][0] / res[1][0]:.3f}x; "
          f"bit-equal {torch.equal(res[0][1], res[1][1])}; converged {100 * (st == 0).mean():.2f} %, iterations mean {it.mean():.1f} max {it.max()}; "
          f"balanced bound {it.sum() / 1024 / it.max():.2f} x the slowest problem", flush=True)
    s.close()
