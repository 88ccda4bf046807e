"""Diagnostic (GPU box): where the wall clock of a one-problem call goes (host-buffer entry point vs device tensors)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver
d = np.load(os.path.join(ROOT, "tests", "golden", "g7_closedloop_exp1.npz"))
s = BatchedOCPSolver(10, 4, 0.1)
s.set_timing(True)
P, X = d["p"][30:31].copy(), d["x0"][30:31].copy()
for _ in range(5): s.solve_host(P, X)
def med(f, n=40):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
print("solve_host (H2D, launch, sync, 8 D2H): %.3f ms; kernel %.3f ms" % (med(lambda: s.solve_host(P, X)), s.last_kernel_ms()))
p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
out = {}
def dev_call():
    s.solve_batch(p, x0, out=out, want=("status", "iters")); torch.cuda.synchronize()
for _ in range(5): dev_call()
print("solve_batch on device tensors + synchronize: %.3f ms; kernel %.3f ms" % (med(dev_call), s.last_kernel_ms()))
s.set_timing(False)
print("same without timing events: %.3f ms" % med(dev_call))
print("H2D of p+x0 via torch: %.3f ms; D2H of x: %.3f ms" % (med(lambda: (torch.tensor(P, device='cuda'), torch.tensor(X, device='cuda'))), med(lambda: out['x'].cpu())))
g = s.capture_step(p, x0, None, 0, ("status", "iters"))
def graph_call():
    g.launch(); torch.cuda.synchronize()
for _ in range(5): graph_call()
print("captured hipGraph step + synchronize: %.3f ms" % med(graph_call))
