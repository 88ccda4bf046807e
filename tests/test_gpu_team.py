"""Team kernels (a workgroup of 4 cooperating waves per problem, boundmpc_amd/csrc/bmpc_team.hip) on the GPU: against the CPU oracle, against
the one-wave kernels, bitwise determinism, the automatic choice by batch size, the warm entry and the fused closed-loop tick."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _solvers(N=10, **kw):
    from boundmpc_amd import BatchedOCPSolver
    one, team = BatchedOCPSolver(N, 4, 0.1, **kw), BatchedOCPSolver(N, 4, 0.1, **kw)
    one.set_team_waves(1); team.set_team_waves(4)
    return one, team


@pytest.mark.parametrize("B", [1, 37, 256])
def test_team_kernel_matches_oracle_and_one_wave_kernel(B):
    import torch
    from boundmpc_amd import workload
    from oracle import c_oracle
    P, X, _ = workload.make_batch(1024, seed=0, rows=(0, B))
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    one, team = _solvers()
    try:
        assert team.team_info(B)["waves"] == 4 and one.team_info(B)["waves"] == 1
        want = ("g", "lam_g", "lam_x", "f", "iters", "status", "kkt")
        o1 = {k: v.cpu().numpy() for k, v in one.solve_batch(p, x0, out={}, want=want).items()}
        o4 = {k: v.cpu().numpy() for k, v in team.solve_batch(p, x0, out={}, want=want).items()}
        o4b = {k: v.cpu().numpy() for k, v in team.solve_batch(p, x0, out={}, want=want).items()}
    finally:
        one.close(); team.close()
    ref = c_oracle.solve(P, X, 10, 4, 0.1, nthreads=16)
    assert (o4["status"] == ref["status"]).all() and (o4["status"] == 0).all()
    assert np.abs(o4["iters"] - ref["iters"]).max() <= 2
    d = (o4["x"] - ref["x"]).reshape(B, 10, 44)[:, :, 8:15]
    assert np.sqrt((d ** 2).mean(axis=(1, 2))).max() < 1e-6 and np.sqrt((d ** 2).mean()) < 1e-7      # rad, per problem and over the batch
    assert o4["kkt"].max() <= 1e-8
    # a team changes the order of a few sums that only feed accept / reject decisions: same iterates as the one-wave kernel
    assert np.array_equal(o4["iters"], o1["iters"])
    for k in ("x", "g", "lam_g", "lam_x", "f"):
        np.testing.assert_allclose(o4[k], o1[k], rtol=1e-9, atol=1e-9, err_msg=k)
        assert np.array_equal(o4[k], o4b[k]), k      # bitwise deterministic from launch to launch


@pytest.mark.parametrize("B,N", [(300, 10), (512, 10), (64, 11), (40, 5), (3, 1), (5, 2)])
def test_pair_kernel_matches_oracle_and_one_wave_kernel(B, N):
    """The pair kernel (two cooperating waves per problem on the one-wave budget, csrc/bmpc_pair.hip): statuses and iterations of the CPU oracle,
    the one-wave kernel's iterates, bitwise determinism; horizons 1 and 2 exercise the ends of the helper / sweep barrier pairing, 11 the longest."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    P, X, _ = workload.make_batch(max(B, 64), seed=5, N=N, rows=(0, B))
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    one, pair = BatchedOCPSolver(N, 4, 0.1), BatchedOCPSolver(N, 4, 0.1)
    one.set_team_waves(1); pair.set_team_waves(2)
    try:
        assert pair.team_info(B)["waves"] == 2
        want = ("g", "lam_g", "lam_x", "f", "iters", "status", "kkt")
        o1 = {k: v.cpu().numpy() for k, v in one.solve_batch(p, x0, out={}, want=want).items()}
        o2 = {k: v.cpu().numpy() for k, v in pair.solve_batch(p, x0, out={}, want=want).items()}
        o2b = {k: v.cpu().numpy() for k, v in pair.solve_batch(p, x0, out={}, want=want).items()}
    finally:
        one.close(); pair.close()
    ref = c_oracle.solve(P, X, N, 4, 0.1, nthreads=16)
    assert (o2["status"] == ref["status"]).all() and (o2["status"] == 0).all()
    assert np.abs(o2["iters"] - ref["iters"]).max() <= 2
    d = (o2["x"] - ref["x"]).reshape(B, N, 44)[:, :, 8:15]
    assert np.sqrt((d ** 2).mean(axis=(1, 2))).max() < 1e-6 and np.sqrt((d ** 2).mean()) < 1e-7      # rad, per problem and over the batch
    assert o2["kkt"].max() <= 1e-8
    assert np.array_equal(o2["iters"], o1["iters"])
    # against the one-wave kernel: the same iterates unless an accept / reject decision sits on the rounding of a sum whose order differs (seed 5,
    # B = 300: one problem 3e-9 rad away -- the 4-wave teams differ from the one-wave kernel by exactly the same bits there)
    for k, tol in (("x", 1e-7), ("g", 1e-8), ("lam_g", 1e-5), ("lam_x", 1e-5), ("f", 1e-8)):
        np.testing.assert_allclose(o2[k], o1[k], rtol=tol, atol=tol, err_msg=k)
        assert np.array_equal(o2[k], o2b[k]), k      # bitwise deterministic from launch to launch


def test_pair_kernel_on_tight_tubes():
    """Pairs where the Riccati sweep is abandoned at a stage barrier and started again (tight tubes: regularisation retries): statuses of the
    oracle, the one-wave kernel's solutions.  (The restoration hand-over of jammed problems: test_first_failures_of_the_closed_loops_g13b_on_the_gpu.)"""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    P, X, _ = workload.make_batch(64, seed=9, N=10, tight=True, rows=(0, 48))
    one, pair = BatchedOCPSolver(10, 4, 0.1), BatchedOCPSolver(10, 4, 0.1)
    one.set_team_waves(1); pair.set_team_waves(2)
    try:
        p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
        o1 = {k: v.cpu().numpy() for k, v in one.solve_batch(p, x0, out={}, want=("iters", "status", "f")).items()}
        o2 = {k: v.cpu().numpy() for k, v in pair.solve_batch(p, x0, out={}, want=("iters", "status", "f")).items()}
    finally:
        one.close(); pair.close()
    ref = c_oracle.solve(P, X, 10, 4, 0.1, nthreads=16)
    assert np.array_equal(o2["status"], ref["status"]) and np.array_equal(o2["status"], o1["status"]) and (o2["status"] == 0).mean() > 0.9
    assert (np.abs(o2["iters"] - o1["iters"]) <= 2).mean() >= 0.9
    ok = o2["status"] == 0
    dq = (o2["x"] - o1["x"]).reshape(len(P), 10, 44)[ok][:, :, 8:15]
    assert np.sqrt((dq ** 2).mean(axis=(1, 2))).max() < 1e-5 and np.sqrt((dq ** 2).mean()) < 1e-6


def test_team_choice_is_automatic_by_batch_size():
    from boundmpc_amd import BatchedOCPSolver
    s = BatchedOCPSolver(10, 4, 0.1)
    s11, s30 = BatchedOCPSolver(11, 4, 0.1), BatchedOCPSolver(30, 4, 0.1)
    try:
        r = s.team_info(1)["resident_teams"]
        assert r >= 64 and s.team_info(1)["lds_bytes"] <= 160 * 1024
        assert s.team_info(1)["waves"] == 4 and s.team_info(r)["waves"] == 4
        # beyond the resident teams: pairs (two waves on the one-wave budget: twice as many resident) while the batch fits, then one wave per problem
        rp = s.team_info(r + 1)["resident_teams"]
        assert s.team_info(r + 1)["waves"] == 2 and rp >= 2 * r and s.team_info(r + 1)["lds_bytes"] <= 40 * 1024
        assert s.team_info(rp)["waves"] == 2 and s.team_info(rp + 1)["waves"] == 1 and s.team_info(1024)["waves"] == 1
        # horizons beyond a team's LDS: N = 11 still has its iterate in LDS (pairs), N = 30 runs one wave per problem whatever the batch;
        # asking for teams or pairs there is an error
        assert s11.team_info(1)["waves"] == 2 and s30.team_info(1)["waves"] == 1
        with pytest.raises(Exception):
            s30.set_team_waves(4)
        with pytest.raises(Exception):
            s30.set_team_waves(2)
    finally:
        s.close(); s11.close(); s30.close()


def test_team_kernel_warm_entry_iteration_cap_and_other_horizons():
    import torch
    from boundmpc_amd import workload
    one, team = _solvers()
    try:
        P, X, _ = workload.make_batch(64, seed=4)
        p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
        sa, sb, sc = one.new_state(64), team.new_state(64), team.new_state(64)
        for rep, cap in enumerate((3, 0, 0)):      # capped cold start (status 1), then warm solves from the stored duals
            a = one.solve_batch(p, x0, out={}, want=("iters", "status"), state=sa, max_iter=cap)
            b = team.solve_batch(p, x0, out={}, want=("iters", "status"), state=sb, max_iter=cap)
            c = team.solve_batch(p, x0, out={}, want=("iters", "status"), state=sc, max_iter=cap)
            # the team against itself: bit for bit (iterate and dual state); against the one-wave kernel: the same minimisers (a filter
            # decision on a rounding error may send a warm solve through another trial point: solutions agree to the solve tolerance)
            assert torch.equal(b["x"], c["x"]) and torch.equal(sb, sc) and torch.equal(b["iters"], c["iters"])
            assert torch.equal(a["status"], b["status"]) and int((a["iters"] - b["iters"]).abs().max()) <= 1
            # two solutions to the same KKT tolerance (1e-8, scaled) may differ by tol / curvature in a direction: the joint positions carry
            # weight w_q = 0.01 and bound rows (2 tol / (2 w_q) = 1e-6 rad), the jerks only w_j = 1e-4 (2 tol / (2 w_j) = 1e-4)
            dx = (a["x"] - b["x"]).reshape(64, 10, 44)
            assert float(dx[:, :, 8:15].abs().max()) < 1e-6 and float(dx.abs().max()) < 1e-4
            if cap:
                assert int(b["status"].max()) == 1 and int(b["iters"].max()) == cap
    finally:
        one.close(); team.close()
    for N in (3, 7):
        o, t = _solvers(N)
        try:
            P, X, _ = workload.make_batch(48, seed=5, N=N)
            p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
            a, b = o.solve_batch(p, x0, out={}, want=("iters", "status")), t.solve_batch(p, x0, out={}, want=("iters", "status"))
            assert int(b["status"].max()) == 0 and torch.equal(a["iters"], b["iters"]) and float((a["x"] - b["x"]).abs().max()) < 1e-9
        finally:
            o.close(); t.close()


def test_team_tick_equals_one_wave_tick_in_closed_loop():
    """The fused closed-loop tick {pack, solve, post} by teams (wave 0 packs and post-processes, the team solves) against the one-wave tick
    over 12 ticks of 16 streams: same plant trajectories."""
    import torch
    from boundmpc_amd import workload
    from boundmpc_amd import stream as bstream
    q0s = workload.random_q0(16, seed=3)
    outs = []
    for waves in (1, 4):
        from boundmpc_amd import BatchedOCPSolver
        s = BatchedOCPSolver(10, 4, 0.1)
        s.set_team_waves(waves)
        mpcs, recs = [], []
        for q0 in q0s:
            m, p0fk = workload.make_mpc(q0)
            mpcs.append(m)
            recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0, 0]), np.zeros(7)))
        sb = bstream.StreamBatch(s, mpcs)
        sb.set_robot(np.stack(recs))
        try:
            for t in range(12):
                sb.tick(warm_dual=True, simulate=True)
            torch.cuda.synchronize()
            outs.append((sb.robot.cpu().numpy().copy(), sb.iters.cpu().numpy().copy(), sb.status.cpu().numpy().copy(), sb.traj.cpu().numpy().copy()))
        finally:
            sb.close(); s.close()
    (r1, i1, s1, t1), (r4, i4, s4, t4) = outs
    assert (s4 == 0).all() and np.array_equal(s1, s4)
    np.testing.assert_allclose(r4, r1, atol=1e-7)
    np.testing.assert_allclose(t4, t1, atol=1e-6)


def test_team_work_queue_with_more_problems_than_resident_teams():
    """Teams forced on a batch larger than the resident teams (bmpc_set_team_waves(h, 4)): the teams pull problems from the work queue like the
    one-wave kernel's waves do; same solutions as one wave per problem, bitwise deterministic (which team solves which problem does not matter)."""
    import torch
    from boundmpc_amd import workload
    B = 700
    P, X, _ = workload.make_batch(B, seed=12)
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    one, team = _solvers()
    try:
        assert team.team_info(B)["waves"] == 4 and team.team_info(B)["resident_teams"] < B
        a = one.solve_batch(p, x0, out={}, want=("iters", "status"))
        b = team.solve_batch(p, x0, out={}, want=("iters", "status"))
        c = team.solve_batch(p, x0, out={}, want=("iters", "status"))
        torch.cuda.synchronize()
        assert int(b["status"].max()) == 0 and torch.equal(a["iters"], b["iters"]) and float((a["x"] - b["x"]).abs().max()) < 1e-9
        assert torch.equal(b["x"], c["x"])
    finally:
        one.close(); team.close()


def test_pair_kernel_in_a_captured_graph_and_with_a_dual_state():
    """The pair kernel behind the other entry points of a handle: a captured step (bmpc_graph_create: queue reset + pair kernel + restoration kernel)
    replayed over refreshed buffers equals the direct launches bit for bit; warm entry (dual state carried, iteration cap) equals the one-wave
    kernel's to round-off; the automatic choice picks pairs for 256 < B <= 512."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    B = 300
    s, one = BatchedOCPSolver(10, 4, 0.1), BatchedOCPSolver(10, 4, 0.1)
    one.set_team_waves(1)
    try:
        assert s.team_info(B)["waves"] == 2      # automatic
        p = torch.empty((B, 505), dtype=torch.float64, device="cuda"); x0 = torch.empty((B, 440), dtype=torch.float64, device="cuda")
        st_g, st_d, st_1 = s.new_state(B), s.new_state(B), one.new_state(B)
        graph = s.capture_step(p, x0, state=st_g, max_iter=4)
        for t, seed in enumerate((71, 72, 73)):
            P, X, _ = workload.make_batch(B, seed=seed)
            p.copy_(torch.tensor(P)); x0.copy_(torch.tensor(X))
            og = graph.launch(); torch.cuda.synchronize()
            xg, itg = og["x"].clone(), og["iters"].clone()
            pc, xc = p.clone(), x0.clone()
            od = s.solve_batch(pc, xc, state=st_d, max_iter=4)
            o1 = one.solve_batch(pc, xc, state=st_1, max_iter=4)
            torch.cuda.synchronize()
            assert torch.equal(xg, od["x"]) and torch.equal(itg, od["iters"]) and torch.equal(st_g, st_d)
            assert (od["status"] == 1).all() and torch.equal(od["iters"], o1["iters"])      # capped: four Newton steps each
            assert float((od["x"] - o1["x"]).abs().max()) < 1e-7 and float((st_d - st_1).abs().max() / st_1.abs().max()) < 1e-7
        graph.close()
    finally:
        s.close(); one.close()


@pytest.mark.parametrize("graph_waves,direct_waves", [(2, 1), (1, 2), (2, 4)])
def test_graph_replays_between_direct_launches_of_another_kernel_on_the_same_handle(graph_waves, direct_waves):
    """A captured step replayed on the handle's stream (requested on the null stream) between direct null-stream launches of ANOTHER kernel of the same
    handle -- they share the work-queue words and the workspace.  Until round 6 the queue was reset by a memset node, which the runtime did not order
    before the kernel node behind a cross-stream wait: the replayed pair kernel drew its indices from a queue nobody had reset (stale results, then a
    memory fault; profiles/r06_e_graph_memset_node.txt).  The reset is a kernel now."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    B = 200
    s, ref = BatchedOCPSolver(10, 4, 0.1), BatchedOCPSolver(10, 4, 0.1)
    ref.set_team_waves(1)
    try:
        p = torch.empty((B, 505), dtype=torch.float64, device="cuda"); x0 = torch.empty((B, 440), dtype=torch.float64, device="cuda")
        st_g, st_d, st_r = s.new_state(B), s.new_state(B), ref.new_state(B)
        s.set_team_waves(graph_waves)
        graph = s.capture_step(p, x0, state=st_g, max_iter=4)
        s.set_team_waves(direct_waves)
        for seed in (81, 82, 81):
            P, X, _ = workload.make_batch(B, seed=seed)
            p.copy_(torch.tensor(P)); x0.copy_(torch.tensor(X))
            xg = graph.launch()["x"].clone()
            od = s.solve_batch(p.clone(), x0.clone(), state=st_d, max_iter=4)
            o = ref.solve_batch(p.clone(), x0.clone(), out={}, state=st_r, max_iter=4)
            torch.cuda.synchronize()
            assert float((xg - o["x"]).abs().max()) < 1e-7 and float((od["x"] - o["x"]).abs().max()) < 1e-7
            assert float((st_g - st_r).abs().max() / st_r.abs().max()) < 1e-7
        graph.close()
    finally:
        s.close(); ref.close()
