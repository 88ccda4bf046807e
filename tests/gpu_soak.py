"""Diagnostic (GPU box): soak comparison of the HIP solver with the CPU oracle on several random batches (N=10 and N=30 tight)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, workload
from oracle import c_oracle
worst = 0.0
BIG = len(sys.argv) > 1 and sys.argv[1] == "big"      # ~50 000 problems instead of ~5 000
CASES = ((10, False, 4096, tuple(range(10, 20))), (30, True, 2048, (5, 6)), (5, False, 1024, (6,)), (20, False, 1024, (7,)), (16, True, 1024, (8,)), (40, False, 512, (9,)), (36, True, 256, (3,))) if BIG \
    else ((10, False, 1024, (1, 2, 3, 4)), (30, True, 128, (5,)), (5, False, 256, (6,)))
for (N, tight, B, seeds) in CASES:
    s = BatchedOCPSolver(N, 4, 0.1)
    for seed in seeds:
        P, X, _ = workload.make_batch(B, seed=seed, N=N, tight=tight)
        o = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda"))
        ref = c_oracle.solve(P, X, N, 4, 0.1, nthreads=16)
        st, it, x = o["status"].cpu().numpy(), o["iters"].cpu().numpy(), o["x"].cpu().numpy()
        ok = (st == 0) & (ref["status"] == 0)
        d = (x[ok] - ref["x"][ok]).reshape(-1, N, 44)[:, :, 8:15]
        per = np.sqrt((d ** 2).mean(axis=(1, 2)))
        print(f"N={N} tight={tight} seed={seed}: status equal {int((st == ref['status']).sum())}/{B}, |iters diff| max {int(np.abs(it - ref['iters']).max())}, "
              f"joint RMS {np.sqrt((d ** 2).mean()):.2e}, worst problem {per.max():.2e}, problems > 1e-6: {int((per > 1e-6).sum())}", flush=True)
        worst = max(worst, float(np.sqrt((d ** 2).mean())))
    s.close()
# teams (round 4): batches that fit the resident teams, N <= 10, loose and tight tubes, against the oracle problem by problem
tworst = 0.0
for (N, tight, B, seeds) in (((10, False, 256, tuple(range(20, 36))), (10, True, 256, (36, 37, 38, 39)), (7, True, 256, (40, 41)), (4, False, 200, (42,)), (10, False, 1, (43, 44, 45))) if BIG
                             else ((10, False, 256, (20, 21)), (10, True, 128, (36,)), (6, True, 100, (40,)))):
    s = BatchedOCPSolver(N, 4, 0.1); s.set_team_waves(4)
    for seed in seeds:
        P, X, _ = workload.make_batch(max(B, 2), seed=seed, N=N, tight=tight); P, X = P[:B], X[:B]
        o = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda"))
        ref = c_oracle.solve(P, X, N, 4, 0.1, nthreads=16)
        st, it, x = o["status"].cpu().numpy(), o["iters"].cpu().numpy(), o["x"].cpu().numpy()
        ok = (st == 0) & (ref["status"] == 0)
        d = (x[ok] - ref["x"][ok]).reshape(-1, N, 44)[:, :, 8:15]
        per = np.sqrt((d ** 2).mean(axis=(1, 2)))
        print(f"TEAM N={N} tight={tight} B={B} seed={seed}: status equal {int((st == ref['status']).sum())}/{B}, |iters diff| max {int(np.abs(it - ref['iters']).max())}, "
              f"joint RMS {np.sqrt((d ** 2).mean()):.2e}, worst problem {per.max():.2e}, problems > 1e-6: {int((per > 1e-6).sum())}", flush=True)
        tworst = max(tworst, float(np.sqrt((d ** 2).mean())))
    s.close()
# pairs (round 6): two waves per problem on the one-wave budget (N <= 11), forced for every batch size -- several rounds of the 512 resident pairs included
pworst = 0.0
for (N, tight, B, seeds) in (((10, False, 2048, tuple(range(50, 58))), (10, True, 1024, (58, 59, 60, 61)), (11, True, 512, (62, 63)), (7, True, 512, (64, 65)), (4, False, 300, (66,)), (2, False, 64, (67,)), (1, False, 64, (68,)),
                              (10, False, 1, (69, 70))) if BIG else ((10, False, 600, (50,)), (10, True, 300, (58,)), (11, True, 128, (62,)), (2, False, 32, (67,)))):
    s = BatchedOCPSolver(N, 4, 0.1); s.set_team_waves(2)
    for seed in seeds:
        P, X, _ = workload.make_batch(max(B, 2), seed=seed, N=N, tight=tight); P, X = P[:B], X[:B]
        o = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda"))
        ref = c_oracle.solve(P, X, N, 4, 0.1, nthreads=16)
        st, it, x = o["status"].cpu().numpy(), o["iters"].cpu().numpy(), o["x"].cpu().numpy()
        ok = (st == 0) & (ref["status"] == 0)
        d = (x[ok] - ref["x"][ok]).reshape(-1, N, 44)[:, :, 8:15]
        per = np.sqrt((d ** 2).mean(axis=(1, 2)))
        print(f"PAIR N={N} tight={tight} B={B} seed={seed}: status equal {int((st == ref['status']).sum())}/{B}, |iters diff| max {int(np.abs(it - ref['iters']).max())}, "
              f"joint RMS {np.sqrt((d ** 2).mean()):.2e}, worst problem {per.max():.2e}, problems > 1e-6: {int((per > 1e-6).sum())}", flush=True)
        pworst = max(pworst, float(np.sqrt((d ** 2).mean())))
    s.close()
print("worst batch RMS", worst, "teams", tworst, "pairs", pworst)
