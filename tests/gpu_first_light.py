"""First-light script for the GPU box: solve the closed-loop fixture problems on the HIP path,
compare with the fixtures (CPU oracle solutions), and time a tiled batch."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boundmpc_amd import BatchedOCPSolver

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
d = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))
P, X0, X = d["p"], d["x0"], d["x"]
s = BatchedOCPSolver(10, 4, 0.1)
print("launch info", s.launch_info(), flush=True)
out = s.solve_host(P[:1], X0[:1])
print("single: iters", out["iters"], "status", out["status"], "kkt", out["kkt"], "dx", np.abs(out["x"] - X[:1]).max(), flush=True)
out = s.solve_host(P, X0)
print("ticks", len(P), "status", np.unique(out["status"], return_counts=True), "iters diff", np.abs(out["iters"] - d["iters"]).max(),
      "max dx", np.abs(out["x"] - X).max(), "rms joint", np.sqrt(np.mean((out["x"] - X).reshape(-1, 10, 44)[:, :, 8:15] ** 2)), flush=True)
for B in (1024, 8192):
    reps = (B + len(P) - 1) // len(P)
    p = torch.tensor(np.tile(P, (reps, 1))[:B], device="cuda"); x0 = torch.tensor(np.tile(X0, (reps, 1))[:B], device="cuda")
    s.set_timing(True)
    o = s.solve_batch(p, x0); torch.cuda.synchronize()
    t = time.time(); o = s.solve_batch(p, x0); torch.cuda.synchronize(); dt = time.time() - t
    print(f"B={B}: wall {dt*1e3:.2f} ms  kernel {s.last_kernel_ms():.2f} ms  -> {B/dt:.0f} solves/s; mean iters {o['iters'].float().mean().item():.2f}",
          "status ok", int((o["status"] == 0).sum()), flush=True)
