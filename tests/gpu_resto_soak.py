"""Diagnostic (GPU box): the restoration path under load -- feasible problems from far-off starts over many seeds, x0 taken as given (start_rollout off: the
default would roll these starts out before the first iteration and the phase would hardly be entered), GPU (batch kernel -> restoration kernel, one wave
per problem and teams) against the CPU oracle problem by problem.  `defaults` as a second argument: the handle's defaults instead (the starts rolled out first).
Usage: python tests/gpu_resto_soak.py [seeds=6] [defaults]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boundmpc_amd import BatchedOCPSolver, workload
from oracle import c_oracle
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ROLL = len(sys.argv) > 2 and sys.argv[2] == "defaults"
tot = dict(n=0, conv=0, steq=0, far=0, entered=0, itmax=0)
s1 = BatchedOCPSolver(10, 4, 0.1, max_iter=500, start_rollout=ROLL); s1.set_team_waves(1)
s4 = BatchedOCPSolver(10, 4, 0.1, max_iter=500, start_rollout=ROLL); s4.set_team_waves(4)
s0 = BatchedOCPSolver(10, 4, 0.1, max_iter=500, start_rollout=ROLL); s0.set_restoration(0)
for seed in range(K):
    for tight in (False, True):
        for kind in ("noise 0.3", "noise 1.0", "zeros"):
            B = 512 if kind != "zeros" else 256
            P, X, _ = workload.make_batch(B, seed=100 + seed, N=10, tight=tight)
            rng = np.random.default_rng(1000 + seed)
            X0 = np.zeros_like(X) if kind == "zeros" else X + rng.normal(size=X.shape) * float(kind.split()[1])
            ref = c_oracle.solve(P, X0, 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=500, start_rollout=int(ROLL)), nthreads=16)
            p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X0, device="cuda")
            off = s0.solve_batch(p, x0)["status"].cpu().numpy()
            line = f"seed {100 + seed} {'tight' if tight else 'loose'} {kind:9s} B={B}: oracle converges on {int((ref['status'] == 0).sum())}, phase off {int((off == 0).sum())}"
            for nm, s in (("one wave", s1), ("teams", s4)):
                if nm == "teams" and B > 256:
                    o = {k: torch.cat([s.solve_batch(p[i:i + 256], x0[i:i + 256])[k] for i in range(0, B, 256)]) for k in ("status", "iters", "x")}
                else:
                    o = s.solve_batch(p, x0)
                st, it = o["status"].cpu().numpy(), o["iters"].cpu().numpy()
                both = (st == 0) & (ref["status"] == 0)
                d = (o["x"].cpu().numpy()[both] - ref["x"][both]).reshape(-1, 10, 44)[:, :, 8:15]
                per = np.sqrt((d ** 2).mean(axis=(1, 2)))
                tot["n"] += B; tot["conv"] += int((st == 0).sum()); tot["steq"] += int((st == ref["status"]).sum()); tot["far"] += int((per > 1e-5).sum())
                tot["entered"] += int((off != 0).sum()); tot["itmax"] = max(tot["itmax"], int(np.abs(it - ref["iters"]).max()))
                line += f" | {nm}: {int((st == 0).sum())} converge, status equal {int((st == ref['status']).sum())}, |iters diff| max {int(np.abs(it - ref['iters']).max())}, > 1e-5 rad apart {int((per > 1e-5).sum())}"
            print(line, flush=True)
print(f"TOTAL {tot['n']} solves (each problem on both launch shapes): {tot['conv']} converge, status equal to the oracle's on {tot['steq']}, "
      f"{tot['entered']} of them on problems the main phase alone does not solve from this start, commonly converged pairs > 1e-5 rad apart {tot['far']}, |iters diff| max {tot['itmax']}")
