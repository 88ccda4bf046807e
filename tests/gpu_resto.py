"""Diagnostic (GPU box): the restoration phase on fixture g13b (every first failing tick of the 256 closed loops of BASELINE configs[4]) -- kernels
(one wave per problem and teams) against the CPU oracle, problem by problem: status, iterations, objective.  Usage: python tests/gpu_resto.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver
from oracle import c_oracle

d = np.load(os.path.join(ROOT, "tests", "golden", "g13b_first_failures_256_streams.npz"))
feas = (d["slsqp_eq"] < 1e-8) & (d["slsqp_ineq"] < 1e-8) & (d["slsqp_bounds"] < 1e-8)
ref = c_oracle.solve(d["p"], d["x0"], 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=500), nthreads=8)
p, x0 = torch.tensor(d["p"], device="cuda"), torch.tensor(d["x0"], device="cuda")
print("oracle   status", ref["status"].tolist()); print("oracle   iters ", ref["iters"].tolist())
for waves in (1, 4):
    s = BatchedOCPSolver(10, 4, 0.1, max_iter=500); s.set_team_waves(waves); s.set_timing(True)
    o = s.solve_batch(p, x0); torch.cuda.synchronize(); t0 = time.time(); o = s.solve_batch(p, x0); torch.cuda.synchronize()
    st, it, f = o["status"].cpu().numpy(), o["iters"].cpu().numpy(), o["f"].cpu().numpy()
    print(f"waves {waves} status", st.tolist()); print(f"waves {waves} iters ", it.tolist())
    both = (st == 0) & (ref["status"] == 0)
    print(f"   status equal {int((st == ref['status']).sum())}/38, |iters diff| max {int(np.abs(it - ref['iters']).max())}, feasible converged {int((st[feas] == 0).sum())}/10 "
          f"(max iters {int(it[feas][st[feas] == 0].max())}), infeasible status 2: {int((st[~feas] == 2).sum())}/28 (max iters {int(it[~feas].max())}), "
          f"objective vs oracle {np.abs(f[both] - ref['f'][both]).max() / np.abs(ref['f'][both]).max():.1e} rel, kernel {s.last_kernel_ms():.2f} ms")
    s.close()
