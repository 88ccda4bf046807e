"""Diagnostic (GPU box): what a hand-over of the stragglers of a B = 1024 batch to teams could buy -- the one-wave kernel capped at K iterations,
and a team launch of as many problems as are left after K, capped at the iterations the slowest still needs (cold, not resumed: timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from boundmpc_amd import BatchedOCPSolver, workload
P, X, _ = workload.make_batch(1024, seed=0)
p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
s = BatchedOCPSolver(10, 4, 0.1); s.set_timing(True)
def t(fn, n=5):
    ms = []
    for _ in range(n):
        fn(); torch.cuda.synchronize(); ms.append(s.last_kernel_ms())
    return min(ms)
o = s.solve_batch(p, x0); torch.cuda.synchronize()
it = o["iters"].cpu().numpy()
print("full: %.3f ms, max iters %d" % (t(lambda: s.solve_batch(p, x0)), it.max()))
state = s.new_state(1024)
for K in (11, 12, 13, 14, 15):
    def capped():
        state.zero_(); s.solve_batch(p, x0, state=state, max_iter=K)
    t1 = t(capped)
    idx = np.nonzero(it > K)[0]
    pi, xi = p[idx].contiguous(), x0[idx].contiguous(); st2 = s.new_state(len(idx))
    rem = int(it.max() - K)
    def rest():
        st2.zero_(); s.solve_batch(pi, xi, state=st2, max_iter=rem)
    t2 = t(rest)
    print("K=%d: one-wave capped %.3f ms; %d left, team launch of %d iterations %.3f ms (waves %d); sum %.3f" % (K, t1, len(idx), rem, t2, s.team_info(len(idx))["waves"], t1 + t2))
