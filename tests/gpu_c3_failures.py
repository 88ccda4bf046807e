"""Diagnostic (GPU box): the problems of BASELINE configs[3] (N = 30, tight tubes, seed 2, 8192) that end with status 2 at the handle's defaults (restoration
mode 2 for N > 11: after a numerical breakdown only), solved again with the full restoration phase (mode 1) at several caps: does the l1 feasibility
phase find a feasible point, and does the solve converge from it?  The points are saved for an independent evaluation of the constraints on the CPU
(tests/c3_failures_table.py).  Usage: python tests/gpu_c3_failures.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, workload
N, B = 30, 8192
P, X, _ = workload.make_batch(B, seed=2, N=N, tight=True)
p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
s = BatchedOCPSolver(N, 4, 0.1)
o = s.solve_batch(p, x0, out={}); torch.cuda.synchronize()
st = o["status"].cpu().numpy(); bad = np.where(st != 0)[0]
print("defaults: restoration", s.get_restoration(), "failures", len(bad), bad.tolist(), flush=True)
out = {"idx": bad, "p": P[bad], "x0": X[bad], "x_default": o["x"].cpu().numpy()[bad], "it_default": o["iters"].cpu().numpy()[bad], "st_default": st[bad], "kkt_default": o["kkt"].cpu().numpy()[bad]}
pb, xb = torch.tensor(P[bad], device="cuda"), torch.tensor(X[bad], device="cuda")
for mode, short, cap, mi in ((1, 6, 40, 500), (1, 6, 150, 1000), (1, 6, 400, 2000)):
    s2 = BatchedOCPSolver(N, 4, 0.1, max_iter=mi); s2.set_restoration(mode, short, cap)
    r = s2.solve_batch(pb, xb, out={}); torch.cuda.synchronize()
    tag = f"m{mode}_cap{cap}"
    out["x_" + tag] = r["x"].cpu().numpy(); out["st_" + tag] = r["status"].cpu().numpy(); out["it_" + tag] = r["iters"].cpu().numpy(); out["kkt_" + tag] = r["kkt"].cpu().numpy(); out["f_" + tag] = r["f"].cpu().numpy()
    print(f"mode {mode} short {short} cap {cap} max_iter {mi}: statuses {out['st_' + tag].tolist()} iterations {out['it_' + tag].tolist()}", flush=True)
    s2.close()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "c3_failures.npz"), **out)
