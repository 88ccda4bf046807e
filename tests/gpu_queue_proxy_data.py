"""Diagnostic (GPU box): iteration counts and statuses of long-horizon batches, for the study of work-queue proxies (tests/queue_proxy_study.py, CPU)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, workload
out = {}
for name, N, tight, seed, B in (("c3", 30, True, 2, 8192), ("n30s7", 30, True, 7, 4096), ("n20s26", 20, True, 26, 4096), ("n30loose", 30, False, 9, 4096)):
    P, X, _ = workload.make_batch(B, seed=seed, N=N, tight=tight)
    s = BatchedOCPSolver(N, 4, 0.1); s.set_timing(1)
    o = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda"), out={}, want=("iters", "status", "f")); torch.cuda.synchronize()
    out[name + "_iters"] = o["iters"].cpu().numpy(); out[name + "_status"] = o["status"].cpu().numpy()
    print(name, "kernel ms", s.last_kernel_ms(), "iters mean", out[name + "_iters"].mean(), "max", out[name + "_iters"].max(), flush=True)
    s.close()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "queue_proxy_iters.npz"), **out)
