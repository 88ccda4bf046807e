"""Diagnostic (GPU box): the 26 status-2 problems of configs[3] (build/c3_failures.npz, tests/gpu_c3_failures.py) at the handle's defaults WITH the second
attempt, against the CPU oracle with the same rule, problem by problem.  Usage: python tests/gpu_c3_failures_vs_oracle.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver
from oracle import c_oracle
D = np.load(os.path.join(ROOT, "build", "c3_failures.npz"))
s = BatchedOCPSolver(30, 4, 0.1)
print("second attempt cap", s.get_second_attempt(), "restoration", s.get_restoration())
r = s.solve_batch(torch.tensor(D["p"], device="cuda"), torch.tensor(D["x0"], device="cuda"), out={}); torch.cuda.synchronize()
o = c_oracle.solve(D["p"], D["x0"], 30, 4, 0.1, nthreads=16)
st, it, x = r["status"].cpu().numpy(), r["iters"].cpu().numpy(), r["x"].cpu().numpy()
print("GPU    status", st.tolist(), "\n       iters ", it.tolist())
print("oracle status", o["status"].tolist(), "\n       iters ", o["iters"].tolist())
ok = (st == 0) & (o["status"] == 0)
d = (x[ok] - o["x"][ok]).reshape(-1, 30, 44)[:, :, 8:15]
print(f"status equal {int((st == o['status']).sum())}/26, |iters diff| max {int(np.abs(it - o['iters']).max())}, joint RMS of the converged {np.sqrt((d ** 2).mean()):.2e}, worst problem {np.sqrt((d ** 2).mean(axis=(1, 2))).max():.2e}")
