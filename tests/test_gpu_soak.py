"""The dynamic net behind the ISA lints (boundmpc_amd/build.py), inside the driver-run GPU suite: horizons and tube widths the
other GPU tests do not solve -- N in {5, 16, 20}, loose and tight tubes, 512 random problems each, and N = 40 -- against the CPU oracle, problem by
problem; and the cold / zero-state-warm / repeated-launch determinism check that caught the conditional-load miscompile of round 2
(a build that passed the static lint and produced non-deterministic cold starts).  Bounded: ~3 000 problems, a few seconds of GPU time."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

TOL_PER_PROBLEM = 1e-6      # rad RMS of one problem's joint trajectory against the oracle (measured: <= 2.3e-6 over 48 000 problems incl. N=30)


# problems of a soak batch whose GPU solution is further than TOL_PER_PROBLEM from the oracle's: (N, tight, seed) -> problem indices.  Each is a KKT point to
# 1e-8 reached through another sequence of accept / reject decisions (round-off amplified by an ill-conditioned end game on the last
# barrier level, or a restarted long-horizon solve that ends in a neighbouring minimiser).
KNOWN_FAR = {}


@pytest.mark.parametrize("N,tight,seed", [(5, False, 21), (5, True, 22), (16, False, 23), (16, True, 24), (20, False, 25), (20, True, 26), (40, False, 27)])
def test_soak_other_horizons_against_the_oracle(N, tight, seed):
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    B = 512 if N <= 20 else 192      # N = 40: the longest horizon bmpc_create accepts (kinematics points in two chunks of lanes, four staging passes)
    P, X, _ = workload.make_batch(B, seed=seed, N=N, tight=tight)
    s = BatchedOCPSolver(N, 4, 0.1)
    try:
        o = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda"))
        st, it, x = o["status"].cpu().numpy(), o["iters"].cpu().numpy(), o["x"].cpu().numpy()
    finally:
        s.close()
    ref = c_oracle.solve(P, X, N, 4, 0.1, nthreads=16)
    # the oracle and the kernel run the same algorithm in different arithmetic order: a problem may need an iteration more or less,
    # and on tight tubes a handful sit on the edge of the stall test; statuses must agree on all but those
    agree = float((st == ref["status"]).mean())
    assert agree >= (0.99 if tight else 1.0), agree
    ok = (st == 0) & (ref["status"] == 0)
    assert ok.mean() >= (0.9 if tight else 1.0)
    d = (x[ok] - ref["x"][ok]).reshape(-1, N, 44)[:, :, 8:15]
    per = np.sqrt((d ** 2).mean(axis=(1, 2)))
    # problems that converge to the same minimiser agree to round-off x conditioning; a different local minimiser (sensitivity
    # note) would show as a deviation of 1e-3 rad or more: at most one problem per batch may do that, none may sit in between
    far = per > TOL_PER_PROBLEM
    # the KNOWN cases, by name: a problem that is not on the list fails the test (until round 5 the test allowed any two per batch)
    idx = set(int(i) for i in np.nonzero(ok)[0][far])
    assert idx <= KNOWN_FAR.get((N, tight, seed), set()), (sorted(idx), [float(v) for v in per[far]])
    assert np.sqrt((d[~far] ** 2).mean()) < 1e-7
    # same algorithm in another arithmetic order: the iteration counts agree on almost every problem; a few take another trial point
    # somewhere and arrive at the same minimiser some iterations earlier or later (the accuracy check above is the criterion)
    assert (np.abs(it[ok] - ref["iters"][ok]) <= 2).mean() >= 0.97


@pytest.mark.parametrize("N", [10, 30])
def test_cold_start_is_bitwise_deterministic_on_every_entry_path(N):
    """One batch, five launches: plain cold start twice, cold start through the warm entry (zeroed dual state) twice, and a cold start after
    a warm one on the same handle: all bit-identical (the round-2 miscompile made cold starts read stale registers: results differed
    from launch to launch and between the two entries)."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    B = 256 if N == 10 else 64
    P, X, _ = workload.make_batch(B, seed=31, N=N, tight=(N == 30))
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    s = BatchedOCPSolver(N, 4, 0.1)
    try:
        runs = []
        runs.append(s.solve_batch(p, x0, out={})["x"].clone())
        runs.append(s.solve_batch(p, x0, out={})["x"].clone())
        st1 = s.new_state(B)
        runs.append(s.solve_batch(p, x0, out={}, state=st1)["x"].clone())
        s.solve_batch(p, x0, out={}, state=st1)                              # a genuinely warm solve in between (different code path)
        runs.append(s.solve_batch(p, x0, out={}, state=s.new_state(B))["x"].clone())
        runs.append(s.solve_batch(p, x0, out={})["x"].clone())
        torch.cuda.synchronize()
    finally:
        s.close()
    for r in runs[1:]:
        assert torch.equal(r, runs[0])
    assert bool(torch.isfinite(runs[0]).all())


def test_queue_order_does_not_change_results():
    """A stateless batch larger than the resident waves is handed out longest-expected-first (bmpc_set_queue_order; default for N > 11: an evaluation
    pass ranks the problems by the objective at x0): every output equals the natural order's bit for bit, garbage starts (NaN keys) included."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    s10, s = BatchedOCPSolver(10, 4, 0.1), BatchedOCPSolver(16, 4, 0.1)
    try:
        assert s10.get_queue_order() == 0 and s.get_queue_order() == 1
        B = s.launch_info()["grid"] + 300
        P, X, _ = workload.make_batch(B, seed=41, N=16, tight=True)
        X[5] = np.nan; X[B - 1] = np.nan      # two starts nobody can evaluate: their keys are NaN, they are solved (to a NaN status 3) like the others
        p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
        outs = []
        for mode in (1, 0, 1):
            s.set_queue_order(mode)
            o = s.solve_batch(p, x0, out={}); torch.cuda.synchronize()
            outs.append({k: v.clone() for k, v in o.items()})
        ok = torch.ones(B, dtype=torch.bool, device="cuda"); ok[5] = False; ok[B - 1] = False
        for k in ("x", "g", "lam_g", "f", "iters", "status", "kkt"):
            assert torch.equal(outs[0][k][ok], outs[1][k][ok]) and torch.equal(outs[0][k][ok], outs[2][k][ok]), k
        st = outs[0]["status"].cpu().numpy()
        assert (st[ok.cpu().numpy()] == 0).mean() > 0.95 and st[5] != 0 and st[B - 1] != 0
        assert (outs[1]["status"].cpu().numpy() == st).all()
    finally:
        s10.close(); s.close()

