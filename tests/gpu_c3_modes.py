import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from boundmpc_amd import BatchedOCPSolver, workload
N, B = 30, 8192
P, X, _ = workload.make_batch(B, seed=2, N=N, tight=True)
p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
for mode, cap, mi in ((2, 40, 500), (1, 40, 500), (1, 150, 1000)):
    s = BatchedOCPSolver(N, 4, 0.1, max_iter=mi); s.set_restoration(mode, 6, cap); s.set_timing(1)
    ms = []
    for _ in range(3):
        o = s.solve_batch(p, x0, out={}, want=("iters", "status")); torch.cuda.synchronize(); ms.append(s.last_kernel_ms())
    st, it = o["status"].cpu().numpy(), o["iters"].cpu().numpy()
    print(f"configs[3] restoration mode {mode} cap {cap}: {min(ms):.1f} ms = {B / min(ms) * 1e3:.0f} solves/s, converged {100 * (st == 0).mean():.2f} % ({int((st != 0).sum())} not), iterations mean {it.mean():.2f} max {it.max()}, queue order {s.get_queue_order()}", flush=True)
    s.close()
