"""Diagnostic (GPU box; `--cpu`: oracle only, no GPU): robustness battery of the solver as a replacement for Ipopt -- feasible problems of the synthetic
generator started from deliberately bad points (Gaussian noise on every variable of the reference's cold start, zeros, uniform noise), at the handle's
defaults (rollout of a cold start that is not a trajectory, restoration phase behind it), with x0 taken as given (restoration phase alone) and with neither (round 4's behaviour); GPU against the CPU oracle problem by problem.
Usage: python tests/gpu_robustness.py [--cpu]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boundmpc_amd import workload
from oracle import c_oracle
CPU = "--cpu" in sys.argv
if not CPU:
    import torch
    from boundmpc_amd import BatchedOCPSolver


def opts(N, **kw):
    return c_oracle.default_opts(max_iter=500, **kw) if N <= 11 else c_oracle.default_opts(max_iter=500, mu_init=3.0, slack_push=0.1, stall_window=20, **kw)


def case(name, P, X, N=10):
    rm = 1 if N <= 11 else 2
    o = c_oracle.solve(P, X, N, 4, 0.1, opts=opts(N, restoration=rm), nthreads=16)
    o1 = c_oracle.solve(P, X, N, 4, 0.1, opts=opts(N, restoration=rm, start_rollout=0), nthreads=16)
    o0 = c_oracle.solve(P, X, N, 4, 0.1, opts=opts(N, restoration=0, start_rollout=0), nthreads=16)
    line = (f"{name:28s} oracle, defaults: {int((o['status'] == 0).sum()):3d} of {len(P)} converge (status {np.bincount(o['status'], minlength=4).tolist()}, "
            f"iterations mean {o['iters'].mean():.1f} max {int(o['iters'].max())}); x0 as given, restoration phase alone: {int((o1['status'] == 0).sum()):3d} (mean {o1['iters'].mean():.1f}); "
            f"neither (round 4): {int((o0['status'] == 0).sum()):3d} (status {np.bincount(o0['status'], minlength=4).tolist()})")
    if not CPU:
        s = BatchedOCPSolver(N, 4, 0.1, max_iter=500)
        g = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")); st, it = g["status"].cpu().numpy(), g["iters"].cpu().numpy()
        both = (st == 0) & (o["status"] == 0)
        d = (g["x"].cpu().numpy()[both] - o["x"][both]).reshape(-1, N, 44)[:, :, 8:15]
        per = np.sqrt((d ** 2).mean(axis=(1, 2))) if both.any() else np.zeros(1)
        line += f" | GPU: {int((st == 0).sum()):3d} converge, status equal to the oracle's on {int((st == o['status']).sum())}, |iters diff| max {int(np.abs(it - o['iters']).max())}, problems > 1e-5 rad apart {int((per > 1e-5).sum())}"
        s.close()
    print(line, flush=True)


rng = np.random.default_rng(11)
Pl, Xl, _ = workload.make_batch(128, seed=60, N=10, tight=False)
Pt, Xt, _ = workload.make_batch(128, seed=61, N=10, tight=True)
case("cold start (reference)", Pl, Xl); case("cold start, tight tubes", Pt, Xt)
for nz in (0.1, 0.3, 0.6, 1.0, 2.0):
    case(f"loose, noise {nz}", Pl, Xl + rng.normal(size=Xl.shape) * nz)
    case(f"tight, noise {nz}", Pt, Xt + rng.normal(size=Xt.shape) * nz)
case("loose, x0 = 0", Pl, np.zeros_like(Xl)); case("tight, x0 = 0", Pt, np.zeros_like(Xt))
case("loose, x0 uniform(-1, 1)", Pl, rng.uniform(-1, 1, Xl.shape))
P3, X3, _ = workload.make_batch(64, seed=62, N=30, tight=True)
case("N=30 tight, cold start", P3, X3, 30); case("N=30 tight, noise 0.1", P3, X3 + rng.normal(size=X3.shape) * 0.1, 30); case("N=30 tight, noise 0.3", P3, X3 + rng.normal(size=X3.shape) * 0.3, 30)
P2, X2, _ = workload.make_batch(64, seed=63, N=20)
case("N=20 loose, noise 0.3", P2, X2 + rng.normal(size=X2.shape) * 0.3, 20)
