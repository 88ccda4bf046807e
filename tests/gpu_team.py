"""Diagnostic (GPU box): the team kernels (4 cooperating waves per problem) against the one-wave kernels: results, kernel time per batch size,
time per iteration from an iteration-capped sweep.  Usage: python tests/gpu_team.py [B ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from boundmpc_amd import BatchedOCPSolver, workload  # noqa: E402

Bs = [int(a) for a in sys.argv[1:]] or [1, 16, 64, 128, 256]
P, X, _ = workload.make_batch(1024, seed=0, rows=(0, max(Bs)))
p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")


def run(s, B, n=20, **kw):
    s.set_timing(1)
    ms = []
    for _ in range(n):
        o = s.solve_batch(p[:B], x0[:B], out={}, want=("iters", "status", "kkt"), **kw); torch.cuda.synchronize(); ms.append(s.last_kernel_ms())
    return o, float(np.median(ms[2:]))


one, team = BatchedOCPSolver(10, 4, 0.1), BatchedOCPSolver(10, 4, 0.1)
one.set_team_waves(1); team.set_team_waves(4)
print("team info", team.team_info(256), "one-wave grid", one.launch_info())
for B in Bs:
    o1, t1 = run(one, B); o4, t4 = run(team, B)
    same = torch.equal(o1["x"], o4["x"]); it1, it4 = o1["iters"].cpu().numpy(), o4["iters"].cpu().numpy()
    d = (o1["x"] - o4["x"]).reshape(B, 10, 44)[:, :, 8:15]
    print(f"B={B:4d}: one wave {t1:.3f} ms, team {t4:.3f} ms ({t1 / t4:.2f}x); bit-equal {same}, joint RMS diff {float(torch.sqrt((d ** 2).mean())):.2e}, "
          f"iters max {it1.max()} / {it4.max()}, equal iters {bool((it1 == it4).all())}, status0 {(o4['status'] == 0).float().mean().item():.3f}")
# time per iteration: warm entry with an iteration cap (state zeroed = cold start), K = 2 and K = 10
for B in (Bs[0], Bs[-1]):
    for name, s in (("one", one), ("team", team)):
        ts = {}
        for K in (2, 10):
            st = s.new_state(B)
            _, ts[K] = run(s, B, n=12, state=st, max_iter=K) if False else (None, None)
            ms = []
            for _ in range(12):
                st.zero_()
                s.solve_batch(p[:B], x0[:B], out={}, want=("iters", "status"), state=st, max_iter=K); torch.cuda.synchronize(); ms.append(s.last_kernel_ms())
            ts[K] = float(np.median(ms[2:]))
        print(f"B={B:4d} {name:4s}: K=2 {ts[2] * 1e3:.0f} us, K=10 {ts[10] * 1e3:.0f} us -> {(ts[10] - ts[2]) / 8 * 1e3:.1f} us per iteration")
one.close(); team.close()
