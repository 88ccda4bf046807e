"""Parity tests proper: the HIP path through the C ABI (boundmpc_amd -> libboundmpc_hip.so) against the
oracle and the committed golden fixtures.  Need a real MI355X: `pytest -m gpu`.

Tolerances (fp64 everywhere).  The north star asks for <= 1e-4 rad RMS joint-trajectory error against the
reference solution; here the HIP path is held to 1e-7 rad RMS against the CPU oracle (same algorithm,
different summation orders / libm) and to 1e-4 rad RMS against the independent scipy solution."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
TOL_Q_RMS = 1e-7


@pytest.fixture(scope="module")
def solver():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from boundmpc_amd import BatchedOCPSolver
    s = BatchedOCPSolver(10, 4, 0.1)
    yield s
    s.close()


def _rms_q(a, b, N=10):
    d = (a - b).reshape(-1, N, 44)[:, :, 8:15]
    return float(np.sqrt(np.mean(d ** 2)))


def test_loaded_library_is_the_in_tree_hip_extension(solver):
    from boundmpc_amd import LIB_PATH
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(LIB_PATH) in maps
    # (the oracle may be mapped too: the TESTS load it as the checker; that the product never does is tests/test_cabi.py's job)
    info = solver.launch_info()
    assert info["grid"] >= 256 and info["lds_bytes"] <= 160 * 1024


@pytest.mark.parametrize("which", [1, 2])
def test_tick0_against_oracle_and_numpy_kkt(solver, which):
    from oracle import c_oracle, nlp
    d = np.load(os.path.join(G, f"g6_pack_exp{which}_tick0.npz"))
    p, x0 = d["p_f64"], d["x0_f64"]
    out = solver.solve_host(p, x0)
    ref = c_oracle.solve(p, x0, 10, 4, 0.1)
    assert out["status"][0] == 0 and abs(int(out["iters"][0]) - int(ref["iters"][0])) <= 1
    assert _rms_q(out["x"], ref["x"]) < TOL_Q_RMS
    np.testing.assert_allclose(out["x"], ref["x"], atol=1e-7)
    np.testing.assert_allclose(out["lam_g"], ref["lam_g"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out["lam_x"], ref["lam_x"], rtol=1e-6, atol=1e-6)
    # certificate from the independent numpy restatement
    f, g = nlp.nlp_eval(out["x"][0], p, 10, 4, 0.1)
    np.testing.assert_allclose(out["g"][0], g, atol=1e-11)
    assert abs(out["f"][0] - f) < 1e-9 * abs(f)
    g2 = g.reshape(10, 43)
    assert np.abs(g2[:, :36]).max() < 1e-6 and g2[:, 36:].max() < 1e-6
    gf, Jg = nlp.jac_g_complex_step(out["x"][0], p, 10, 4, 0.1)
    r = gf + Jg.T @ out["lam_g"][0] + out["lam_x"][0]
    assert np.abs(r).max() < 1e-4


@pytest.mark.parametrize("which", [1, 2])
def test_tick0_against_independent_scipy_solution(solver, which):
    d = np.load(os.path.join(G, f"g8_scipy_exp{which}_tick0.npz"))
    out = solver.solve_host(d["p"], d["x0"])
    assert _rms_q(out["x"], d["x"][None]) < 1e-6          # measured ~1e-8 rad; north-star tolerance 1e-4 rad RMS


@pytest.mark.parametrize("which", [1, 2])
def test_closed_loop_fixture_ticks(solver, which):
    """Every tick of the committed closed loop (warm starts, segment switches, the integrated-omega unwrap)."""
    d = np.load(os.path.join(G, f"g7_closedloop_exp{which}.npz"))
    out = solver.solve_host(d["p"], d["x0"])
    assert (out["status"] == 0).all()
    assert np.abs(out["iters"] - d["iters"]).max() <= 1
    assert _rms_q(out["x"], d["x"]) < TOL_Q_RMS
    assert np.abs(out["x"] - d["x"]).reshape(-1, 10, 44)[:, :, 8:].max() < 1e-5


def test_random_batch_1024_against_oracle_and_properties(solver):
    """BASELINE.json configs[1]: batch = 1024 random q0, N = 10.  Oracle comparison on every problem, plus
    size-independent properties: determinism, invariance under permutation of the batch, feasibility."""
    import torch
    from boundmpc_amd import workload
    from oracle import c_oracle
    P, X, _ = workload.make_batch(1024, seed=0)
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    o = solver.solve_batch(p, x0)
    torch.cuda.synchronize()
    x, st, it = o["x"].cpu().numpy(), o["status"].cpu().numpy(), o["iters"].cpu().numpy()
    ref = c_oracle.solve(P, X, 10, 4, 0.1)
    assert (st == ref["status"]).all() and (st == 0).mean() > 0.99
    ok = st == 0
    assert np.abs(it[ok] - ref["iters"][ok]).max() <= 2
    assert _rms_q(x[ok], ref["x"][ok]) < TOL_Q_RMS
    g = o["g"].cpu().numpy()[ok].reshape(-1, 10, 43)
    assert np.abs(g[:, :, :36]).max() < 1e-5 and g[:, :, 36:].max() < 1e-5
    assert (o["kkt"].cpu().numpy()[ok] <= 1e-8).all()
    # determinism
    o2 = solver.solve_batch(p, x0, out={})
    torch.cuda.synchronize()
    assert torch.equal(o2["x"], o["x"])
    # permutation of the batch (different wave <-> problem assignment) gives the same per-problem bits
    perm = torch.randperm(1024, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    o3 = solver.solve_batch(p[perm].contiguous(), x0[perm].contiguous(), out={})
    torch.cuda.synchronize()
    assert torch.equal(o3["x"], o["x"][perm])


def test_long_horizon_tight_tubes(solver):
    """BASELINE.json configs[3] (N = 30, tight bounds, batch 8192) at FULL size: size-independent properties on all 8192 problems
    (every solve ends converged or with a detected stall, within a bounded number of iterations; converged ones are feasible KKT
    points; bitwise determinism) plus the oracle on a 256-problem sample, problem by problem."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    P, X, _ = workload.make_batch(8192, seed=2, N=30, tight=True)
    s30 = BatchedOCPSolver(30, 4, 0.1)
    try:
        p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
        o = s30.solve_batch(p, x0)
        torch.cuda.synchronize()
        x, st, it, kkt = o["x"].cpu().numpy(), o["status"].cpu().numpy(), o["iters"].cpu().numpy(), o["kkt"].cpu().numpy()
        assert set(np.unique(st)) <= {0, 2, 3}, np.bincount(st)          # converged | stalled (local infeasibility) | numerical
        assert (st == 3).mean() <= 0.002
        ok = st == 0
        # Requirement of the round-3 / round-5 verdicts: >= 99.8 % converged, slowest problem <= 120 iterations.  Round 6: the converged fraction is met
        # -- 99.93 % -- by the SECOND ATTEMPT of the status-2 solves (most of them are feasible problems that converge from x0 on the barrier start of the
        # short horizons, profiles/r06_h_configs3_failures.txt); the iteration bound is not (a solve that takes both attempts: 120-223 iterations in sum).
        # Regression guard around those numbers (first attempt alone, three barrier restarts = round 5: 99.68 %, slowest 170, mean 32.8); the reference's iteration cap is 500 (BoundMPC.py:122)
        assert s30.get_second_attempt() == 100
        assert ok.mean() >= 0.999 and it.max() <= 280 and it.mean() <= 34.0, (ok.mean(), it.max(), it.mean())
        s30.set_second_attempt(0)
        o1 = s30.solve_batch(p, x0, out={}, want=("iters", "status")); torch.cuda.synchronize()
        st1, it1 = o1["status"].cpu().numpy(), o1["iters"].cpu().numpy()
        # the first attempt alone = round 5's solver (three barrier restarts: 99.68 %, slowest 170); with a second attempt behind it the solve gives up after two
        assert (st1 == 0).mean() >= 0.996 and it1.max() <= 220 and ((st1 == 0) & ~ok).sum() <= 3, ((st1 == 0).mean(), it1.max(), ((st1 == 0) & ~ok).sum())
        same_path = (st1 == 0) & (it1 == it)
        assert same_path.mean() >= 0.97 and torch.equal(o1["x"][torch.tensor(same_path, device="cuda")], o["x"][torch.tensor(same_path, device="cuda")])
        s30.set_second_attempt(100)
        assert (kkt[ok] <= 1e-8).all()
        g = o["g"].cpu().numpy()[ok].reshape(-1, 30, 43)
        assert np.abs(g[:, :, :36]).max() < 1e-6 and g[:, :, 36:].max() < 1e-6
        lbx, ubx, _, _ = s30.bounds()
        assert (x[ok] >= lbx - 1e-8).all() and (x[ok] <= ubx + 1e-8).all()
        # tubes really are tight here: an inequality row is active on most converged problems
        lam = o["lam_g"].cpu().numpy()[ok].reshape(-1, 30, 43)[:, :, 38:]
        assert (lam.max(axis=(1, 2)) > 1e-3).mean() > 0.9
        o2 = s30.solve_batch(p, x0, out={})
        torch.cuda.synchronize()
        assert torch.equal(o2["x"], o["x"]) and torch.equal(o2["status"], o["status"])
        # the second attempt problem by problem against the oracle's mirror of the rule: three feasible problems the first attempt gives up on, one that stays
        hard = np.array([695, 814, 4960, 1658])
        refh = c_oracle.solve(P[hard], X[hard], 30, 4, 0.1)
        assert list(st1[hard]) == [2, 2, 2, 2] and list(st[hard]) == [0, 0, 0, 2] and np.array_equal(refh["status"], st[hard]), (st1[hard], st[hard], refh["status"])
        assert np.abs(refh["iters"] - it[hard]).max() <= 25, (refh["iters"], it[hard])
        assert np.sqrt(np.mean(((x[hard[:3]] - refh["x"][:3]).reshape(-1, 30, 44)[:, :, 8:15]) ** 2, axis=(1, 2))).max() < 1e-6      # joint trajectories
        # oracle, per problem, on a sample spread over the batch
        idx = np.arange(0, 8192, 32)
        ref = c_oracle.solve(P[idx], X[idx], 30, 4, 0.1)
        same = ref["status"] == st[idx]
        assert same.mean() >= 0.99, same.mean()     # a solve whose last restart stalls at the threshold may fall on either side
        k = same & (ref["status"] == 0)
        # these are 40...120-iteration runs through nonconvex regions: round-off differences between the two implementations can
        # change an accepted step length somewhere, after which the iteration counts drift apart (and a problem with several
        # local minimisers may end in another one); most runs stay in lockstep
        dit = np.abs(it[idx][k] - ref["iters"][k])
        assert np.median(dit) == 0 and (dit <= 3).mean() >= 0.85, np.sort(dit)[-10:]
        per = np.sqrt(np.mean(((x[idx][k] - ref["x"][k]).reshape(-1, 30, 44)[:, :, 8:15]) ** 2, axis=(1, 2)))
        assert (per < 1e-6).mean() >= 0.9 and np.median(per) < 1e-7, (np.sort(per)[-5:], np.median(per))
    finally:
        s30.close()


def test_edge_cases(solver):
    import torch
    from boundmpc_amd import BatchedOCPSolver
    d = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))
    # empty batch
    e = solver.solve_batch(torch.empty((0, 505), dtype=torch.float64, device="cuda"), torch.empty((0, 440), dtype=torch.float64, device="cuda"))
    assert e["x"].shape == (0, 440)
    # ragged sizes (not multiples of the wave / grid size)
    for B in (1, 3, 65):
        o = solver.solve_host(d["p"][:B], d["x0"][:B])
        assert (o["status"] == 0).all() and _rms_q(o["x"], d["x"][:B]) < TOL_Q_RMS
    # iteration cap -> status 1 ("failure is data", BoundMPC.py:465-489), never an exception
    s3 = BatchedOCPSolver(10, 4, 0.1, max_iter=3)
    o = s3.solve_host(d["p"][:4], d["x0"][:4])
    assert (o["status"] == 1).all() and (o["iters"] == 3).all() and np.isfinite(o["x"]).all()
    s3.close()
    # the workspace of a handle is sized by its first batch and grows with later ones; captured graphs pin it
    from boundmpc_amd import BoundMPCHipError
    sg = BatchedOCPSolver(10, 4, 0.1)
    for B in (1, 70, 3):
        o = sg.solve_host(d["p"][:B], d["x0"][:B])
        assert (o["status"] == 0).all() and _rms_q(o["x"], d["x"][:B]) < TOL_Q_RMS
    p4, x4 = torch.tensor(d["p"][:4], device="cuda"), torch.tensor(d["x0"][:4], device="cuda")
    sg2 = BatchedOCPSolver(10, 4, 0.1)
    gr = sg2.capture_step(p4, x4)
    gr.launch(); torch.cuda.synchronize()
    assert _rms_q(gr.out["x"].cpu().numpy(), d["x"][:4]) < TOL_Q_RMS
    with pytest.raises(BoundMPCHipError):          # 2000 problems need more slabs than the graph's handle holds
        sg2.solve_batch(torch.zeros((2000, 505), dtype=torch.float64, device="cuda"), torch.zeros((2000, 440), dtype=torch.float64, device="cuda"))
    gr.close(); sg.close(); sg2.close()
    # wrong shapes / dtypes raise on the host side
    with pytest.raises(ValueError):
        solver.solve_batch(torch.zeros((2, 504), dtype=torch.float64, device="cuda"), torch.zeros((2, 440), dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        solver.solve_batch(torch.zeros((2, 505), dtype=torch.float32, device="cuda"), torch.zeros((2, 440), dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError):      # non-contiguous views are refused (no hidden temporaries on a foreign stream)
        solver.solve_batch(torch.zeros((2, 1010), dtype=torch.float64, device="cuda")[:, ::2], torch.zeros((2, 440), dtype=torch.float64, device="cuda"))


def test_bound_mpc_step_closed_loop_drop_in():
    """The drop-in object: boundmpc_amd.BoundMPC.step() driven like bound_mpc_node.py:292-372 drives the
    reference, HIP solver behind self.solver.  The closed loop must retrace the committed fixture."""
    from boundmpc_amd import workload
    from boundmpc_amd.bound_mpc import BoundMPC, integrate_joint
    from boundmpc_amd.robot_model import RobotModel
    d6 = np.load(os.path.join(G, "g6_pack_exp1_tick0.npz"))
    d7 = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))
    mk = lambda k: [np.array(v) for v in d6[k]]
    mpc = BoundMPC(mk("p_via"), mk("r_via"), [mk("p_lower"), mk("p_upper")], [mk("r_lower"), mk("r_upper")], mk("bp1_in"), mk("br1_in"),
                   list(d6["s_in"]), list(d6["e_p_min_in"]), list(d6["e_r_min_in"]), list(d6["e_p_max_in"]), list(d6["e_r_max_in"]),
                   p0=d6["p0fk"].copy(), params=workload.Params(weights=d6["weights_f64"], build=True))
    rm = RobotModel()
    q, dq, ddq, jerk, v = d6["q0"].copy(), np.zeros(7), np.zeros(7), np.zeros(7), np.zeros(6)
    x_phi_d = np.array([mpc.phi_max[0], 0, 0])
    for i in range(25):
        p_lie, _, _ = rm.forward_kinematics(q, dq)
        traj, _, _, t_mpc, iters = mpc.step(q, dq, ddq, p_lie, v, x_phi_d, jerk)
        assert mpc.error_count == 0 and abs(iters - d7["iters"][i]) <= 1
        np.testing.assert_allclose(traj["q"], d7["traj_q"][i], atol=1e-6)
        np.testing.assert_allclose(q, d7["q"][i], atol=1e-6)
        jm = np.concatenate((jerk[:, None], traj["dddq"][:, :2]), axis=1)
        q, dq, ddq, p_lie, v = integrate_joint(rm, jm, q, dq, ddq, mpc.dt)[:5]
        jerk = traj["dddq"][:, 0].copy()
    assert abs(mpc.phi_current[0] - d7["phi_current"][24]) < 1e-6


def _streams():
    """Two recorded closed-loop streams (experiment1 / experiment2 fixtures) as a batch of 2 problems per tick."""
    d1 = np.load(os.path.join(G, "g7_closedloop_exp1.npz")); d2 = np.load(os.path.join(G, "g7_closedloop_exp2.npz"))
    return (lambda t: np.stack([d1["p"][t], d2["p"][t]])), (lambda t: np.stack([d1["x0"][t], d2["x0"][t]])), (lambda t: np.stack([d1["x"][t], d2["x"][t]]))


@pytest.mark.parametrize("cap", [0, 2])
def test_warm_started_stream_against_oracle(solver, cap):
    """bmpc_solve_batch_warm: dual state carried (and shifted) across ticks; converged (cap 0) and real-time-iteration
    (2 Newton steps per tick) modes against the CPU oracle fed with the same sequence."""
    import torch
    from oracle import c_oracle
    P, X0, XF = _streams()
    T = 8
    st_d = solver.new_state(2)
    st_o = np.zeros((2, c_oracle.state_len(10)))
    for t in range(T):
        if t:
            solver.shift_state(st_d)
            nu = st_o[:, :570].reshape(2, 10, 57); nu[:, :-1] = nu[:, 1:].copy()
        o = solver.solve_batch(torch.tensor(P(t), device="cuda"), torch.tensor(X0(t), device="cuda"), state=st_d, max_iter=cap)
        ref = c_oracle.solve(P(t), X0(t), 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=cap) if cap else None, state=st_o)
        it = o["iters"].cpu().numpy()
        assert np.abs(it - ref["iters"]).max() <= (0 if cap else 1)
        x = o["x"].cpu().numpy()
        if cap:
            assert (o["status"].cpu().numpy() == 1).all()
            np.testing.assert_allclose(x, ref["x"], atol=1e-8)
            np.testing.assert_allclose(st_d.cpu().numpy()[:, :571], st_o[:, :571], rtol=1e-6, atol=1e-9)
        else:
            assert (o["status"].cpu().numpy() == 0).all()
            assert _rms_q(x, ref["x"]) < TOL_Q_RMS
            assert _rms_q(x, XF(t)) < 1e-6          # same minimiser as the cold-started fixture solve
        st_h = st_d.cpu().numpy()
        assert (st_h[:, 570] > 0).all() and (st_h[:, 571] == it).all()
    # a zeroed state is a cold start: identical bits to bmpc_solve_batch
    p, x0 = torch.tensor(P(0), device="cuda"), torch.tensor(X0(0), device="cuda")
    a = solver.solve_batch(p, x0)["x"].clone()
    b = solver.solve_batch(p, x0, state=solver.new_state(2))["x"]
    assert torch.equal(a, b)


def test_hip_graph_step_replays_and_matches_direct_launch(solver):
    """bmpc_graph_create/_launch: the captured {queue reset + kernel} step over fixed buffers, replayed tick after tick with
    refreshed contents, gives bit-identical results to direct launches of the same sequence."""
    import torch
    P, X0, _ = _streams()
    T = 5
    p = torch.empty((2, 505), dtype=torch.float64, device="cuda"); x0 = torch.empty((2, 440), dtype=torch.float64, device="cuda")
    st_g, st_d = solver.new_state(2), solver.new_state(2)
    graph = solver.capture_step(p, x0, state=st_g, max_iter=3)
    for t in range(T):
        if t:
            solver.shift_state(st_g); solver.shift_state(st_d)
        p.copy_(torch.tensor(P(t))); x0.copy_(torch.tensor(X0(t)))
        og = graph.launch()
        torch.cuda.synchronize()
        xg, itg = og["x"].clone(), og["iters"].clone()
        pc, xc = p.clone(), x0.clone()          # kept alive: the call is asynchronous and reads them on the launch stream
        od = solver.solve_batch(pc, xc, state=st_d, max_iter=3)
        assert torch.equal(xg, od["x"]) and torch.equal(itg, od["iters"]) and torch.equal(st_g, st_d)
    graph.close()
    with pytest.raises(ValueError):
        solver.solve_batch(p, x0, max_iter=3)                     # per-call cap needs a state buffer
    with pytest.raises(ValueError):
        solver.solve_batch(p, x0, state=torch.zeros((2, 5), dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("S,N", [(2, 5), (3, 12)])
def test_other_window_sizes_and_horizons(S, N):
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    P, X, _ = workload.make_batch(8, seed=4, N=N, S=S, workers=1)
    s = BatchedOCPSolver(N, S, 0.1)
    o = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda"))
    ref = c_oracle.solve(P, X, N, S, 0.1, nthreads=4)
    assert (o["status"].cpu().numpy() == 0).all()
    assert np.abs(o["iters"].cpu().numpy() - ref["iters"]).max() <= 1
    assert _rms_q(o["x"].cpu().numpy(), ref["x"], N) < TOL_Q_RMS
    s.close()


def test_plain_c_caller_of_the_abi(tmp_path):
    """The boundary is a C ABI: a gcc-built C program (tests/cabi/cabi_demo.c) links the in-tree library, solves experiment1
    tick 0 from host buffers and must print the oracle's solution."""
    import subprocess
    from boundmpc_amd import LIB_PATH
    from oracle import c_oracle
    d = np.load(os.path.join(G, "g6_pack_exp1_tick0.npz"))
    prob = tmp_path / "problem.bin"
    np.concatenate([d["p_f64"], d["x0_f64"]]).astype(np.float64).tofile(prob)
    exe = tmp_path / "cabi_demo"
    libdir = os.path.dirname(os.path.realpath(LIB_PATH))
    subprocess.check_call(["gcc", "-O1", "-o", str(exe), os.path.join(os.path.dirname(__file__), "cabi", "cabi_demo.c"),
                           "-L" + libdir, "-lboundmpc_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe), str(prob)], check=True, capture_output=True, text=True, timeout=120).stdout.split()
    ref = c_oracle.solve(d["p_f64"], d["x0_f64"], 10, 4, 0.1)
    assert int(out[0]) == 0 and abs(int(out[1]) - int(ref["iters"][0])) <= 1
    assert abs(float(out[2]) - ref["f"][0]) < 1e-8 * abs(ref["f"][0])
    np.testing.assert_allclose([float(v) for v in out[3:10]], ref["x"][0][8:15], atol=1e-8)


def test_gauss_newton_option_matches_oracle():
    """options.exact_hessian = 0 (Gauss-Newton Hessian) takes the other branch of the node-cost phases."""
    import torch
    from boundmpc_amd import BatchedOCPSolver
    from oracle import c_oracle
    d = np.load(os.path.join(G, "g6_pack_exp1_tick0.npz"))
    s = BatchedOCPSolver(10, 4, 0.1, exact_hessian=False)
    out = s.solve_host(d["p_f64"], d["x0_f64"])
    ref = c_oracle.solve(d["p_f64"], d["x0_f64"], 10, 4, 0.1, c_oracle.default_opts(exact_hessian=0))
    assert out["status"][0] == 0 and abs(int(out["iters"][0]) - int(ref["iters"][0])) <= 1
    assert _rms_q(out["x"], ref["x"]) < TOL_Q_RMS
    s.close()


@pytest.mark.gpu
def test_infeasible_problem_ends_as_status_2_like_the_oracle(solver):
    """phi_max < 0 <= phi: no feasible point.  Status 2 (stalled at a point of local infeasibility) at the oracle's iteration, finite outputs,
    and the feasible problems of the same batch are not disturbed."""
    from oracle import c_oracle
    d = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))
    p, x0 = d["p"][:6].copy(), d["x0"][:6].copy()
    p[1, 460] = p[4, 460] = -1.0
    o, ref = solver.solve_host(p, x0), c_oracle.solve(p, x0, 10, 4, 0.1)
    assert list(o["status"]) == [0, 2, 0, 0, 2, 0] == list(ref["status"])
    assert np.abs(o["iters"] - ref["iters"]).max() <= 1 and np.isfinite(o["x"]).all()
    ok = o["status"] == 0
    assert _rms_q(o["x"][ok], d["x"][:6][ok]) < TOL_Q_RMS


@pytest.mark.gpu
def test_two_handles_on_two_streams_overlap_and_stay_exact(solver):
    """Independent batches in flight (bench.py --inflight 2): two handles, one HIP stream each, launches interleaved without host
    synchronisation.  Every result is bit-identical to the same batch solved alone, and the kept event pairs give every launch's time."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    P, X, _ = workload.make_batch(512, seed=5, N=10)
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    halves = [(p[:256].contiguous(), x0[:256].contiguous()), (p[256:].contiguous(), x0[256:].contiguous())]
    alone = [solver.solve_batch(a, b)["x"].clone() for a, b in halves]
    torch.cuda.synchronize()
    hs = [BatchedOCPSolver(10, 4, 0.1), BatchedOCPSolver(10, 4, 0.1)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for h in hs:
        h.set_timing(6)
    outs = [{}, {}]
    for rep in range(6):
        for j in (0, 1):
            streams[j].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[j]):
                hs[j].solve_batch(*halves[j], out=outs[j])
    torch.cuda.synchronize()
    for j in (0, 1):
        assert torch.equal(outs[j]["x"], alone[j]) and (outs[j]["status"] == 0).all()
        times = [hs[j].kernel_ms(i) for i in range(6)]
        assert all(0.5 < t < 50.0 for t in times), times
        with pytest.raises(Exception):
            hs[j].kernel_ms(6)                  # only the last six launches are kept
        hs[j].close()


@pytest.mark.gpu
def test_second_seed_against_oracle():
    """A second synthetic batch (seed 1, 512 problems): same statuses, iteration counts within one, joint trajectories within 1e-6 rad
    per problem of the CPU oracle (tests/gpu_soak.py runs the long version of this check)."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    P, X, _ = workload.make_batch(512, seed=1)
    s = BatchedOCPSolver(10, 4, 0.1)
    out = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda"))
    ref = c_oracle.solve(P, X, 10, 4, 0.1)
    st, it, x = out["status"].cpu().numpy(), out["iters"].cpu().numpy(), out["x"].cpu().numpy()
    s.close()
    assert (st == ref["status"]).all() and (st == 0).all()
    assert np.abs(it - ref["iters"]).max() <= 1
    d = (x - ref["x"]).reshape(512, 10, 44)[:, :, 8:15]
    assert np.sqrt((d ** 2).mean(axis=(1, 2))).max() < 1e-6


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    """bench.py (driver contract): ONE JSON line with the metric keys, the roofline object and, at N=1, the cpu_baseline object."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "128", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["value"] > 0 and "workload" in d["config"] and d["config"]["solved_fraction"] == 1.0
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    # counter / flop files are tied to the kernel text by a hash: a file of another kernel text must show as stale, never as a number
    sys.path.insert(0, root)
    import bench
    for name, key in (("pmc_current.json", "traffic_note"), ("flops_current.json", "flops_source")):
        rec = json.load(open(os.path.join(root, "profiles", name)))
        assert "kernel_hash" in rec, name
        if rec["kernel_hash"] != bench.kernel_text_hash():
            where = d["roofline"] if key == "traffic_note" else d["roofline_fp64"]
            assert "stale" in where[key], (name, where[key])
            if key == "traffic_note":
                assert d["roofline"]["traffic"] is None
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-12
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    # compute-side figure priced with the exact operation count of the instrumented oracle, not with a model
    rf = d["roofline_fp64"]
    assert rf["flops_per_iteration"] > 1e6 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "riccati (factorise + forward)" in rf["flops_per_iteration_by_phase"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["gpu_vs_cpu_sample_joint_rms_rad"] < 1e-6
    # `achieved` is priced with the kernel's own operation count; the dense oracle's count (several times larger) is reported beside it
    assert "flops_source" in rf
    if "stale" not in rf["flops_source"]:      # (a count of another kernel text is never used: the oracle's count stands in, and says so)
        assert 1e6 < rf["flops_per_iteration"] < rf["oracle_dense_flops_per_iteration"] and rf["frac"] < rf["oracle_dense_frac"]
    # informational leg at the reference's Ipopt tolerance: fewer iterations, same problems solved, close to the 1e-8 solutions
    rt = d["at_reference_tolerance"]
    assert rt["tol"] == 1e-5 and rt["solved_fraction"] == 1.0 and rt["mean_iters"] < d["config"]["mean_iters"] and rt["joint_rms_vs_tol_1e-8_rad"] < 1e-3


# ---------------------------------------------------------------------------------------------------------------
# G9: f and g of the HIP kernel at fixed points against the values the reference's own builder produced there
# (tests/golden/make_g9.py).  A handle with max_iter = 0 evaluates the NLP at x0 and returns (status 1, x = x0).
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("key,N,S,h", [("n10", 10, 4, 0.1), ("n30", 30, 4, 0.1), ("n3s2", 3, 2, 0.05), ("n5s3", 5, 3, 0.05), ("n4s5", 4, 5, 0.05)])
def test_g9_kernel_f_and_g_equal_the_reference_nlp(key, N, S, h):
    """(every kernel shape that exists for the handle: one wave per problem, pairs, teams -- the evaluation is the same wave program's in each)"""
    from boundmpc_amd import BatchedOCPSolver
    d = np.load(os.path.join(G, "g9_nlp.npz"))
    X, P, F, Gg = d[key + "_x"], d[key + "_p"], d[key + "_f"], d[key + "_g"]
    outs = []
    for waves in (1, 2, 4):
        if S > 4 or (waves == 2 and N > 11) or (waves == 4 and N > 10):
            if waves != 1:
                continue      # no pair / team instantiation for this handle
        s = BatchedOCPSolver(N, S, h, max_iter=0)
        try:
            s.set_team_waves(waves)
            assert s.team_info(len(P))["waves"] == waves
            outs.append(s.solve_host(P, X))
            lbx, ubx, lbg, ubg = s.bounds()
        finally:
            s.close()
    out = outs[0]
    for o2 in outs[1:]:      # the same numbers from every shape (the node phase of teams / pairs splits lifted residuals and references over two waves: same expressions)
        np.testing.assert_array_equal(o2["g"], out["g"]); np.testing.assert_array_equal(o2["f"], out["f"])
    assert (out["status"] == 1).all() and (out["iters"] == 0).all()
    np.testing.assert_array_equal(out["x"], X)
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))
    assert rel(out["f"], F) <= 1e-12, rel(out["f"], F)
    assert rel(out["g"], Gg) <= 1e-11, rel(out["g"], Gg)
    for mine, name in ((lbx, "lbx"), (ubx, "ubx"), (lbg, "lbg"), (ubg, "ubg")):
        np.testing.assert_array_equal(mine, d[f"{key}_{name}"])


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2]: batch = 65536 over 8 GPUs.  One GPU here: rank 3's shard (8192 problems) of the seed-1 batch.
# ---------------------------------------------------------------------------------------------------------------
def test_configs2_shard_8192_properties_and_oracle_sample(solver):
    import torch
    from boundmpc_amd import workload
    from boundmpc_amd.distributed import shard_range
    from oracle import c_oracle
    lo, hi = shard_range(65536, 3, 8)
    assert hi - lo == 8192
    P, X, _ = workload.make_batch(65536, seed=1, rows=(lo, hi))
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    o = solver.solve_batch(p, x0)
    torch.cuda.synchronize()
    x, st, it, kkt = o["x"].cpu().numpy(), o["status"].cpu().numpy(), o["iters"].cpu().numpy(), o["kkt"].cpu().numpy()
    ok = st == 0
    assert ok.mean() >= 0.999, (ok.mean(), np.bincount(st))
    # size-independent properties at full size: KKT error, feasibility of the reference-form constraints and of the bounds
    assert (kkt[ok] <= 1e-8).all()
    g = o["g"].cpu().numpy()[ok].reshape(-1, 10, 43)
    assert np.abs(g[:, :, :36]).max() < 1e-6 and g[:, :, 36:].max() < 1e-6
    lbx, ubx, _, _ = solver.bounds()
    assert (x[ok] >= lbx - 1e-8).all() and (x[ok] <= ubx + 1e-8).all()
    # every problem started on its own path: first-stage joints stay near q0 (cheap sanity of the per-row independence)
    assert np.abs(x[ok].reshape(-1, 10, 44)[:, 0, 8:15] - X[ok].reshape(-1, 10, 44)[:, 0, 8:15]).max() < 0.2
    # determinism at this size
    o2 = solver.solve_batch(p, x0, out={})
    torch.cuda.synchronize()
    assert torch.equal(o2["x"], o["x"]) and torch.equal(o2["iters"], o["iters"])
    # oracle on a 256-problem sample spread over the shard
    idx = np.arange(0, 8192, 32)
    ref = c_oracle.solve(P[idx], X[idx], 10, 4, 0.1)
    assert (ref["status"] == st[idx]).all()
    k = ref["status"] == 0
    assert np.abs(it[idx][k] - ref["iters"][k]).max() <= 2
    assert _rms_q(x[idx][k], ref["x"][k]) < TOL_Q_RMS


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a plain interpreter (no torch.distributed.run): the script starts its two ranks itself.
    One-GPU box: --rehearse-one-gpu puts both ranks on cuda:0 and gathers over gloo (a rehearsal of the plumbing, not a number)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--steps", "2", "--warmup", "1",
                        "--batch", "128", "--workers", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 256 and rec["value"] > 0
    assert rec["config"]["solved_fraction"] == 1.0
    # every rank solves its own shard of ONE global batch: the job's problems are all distinct, and the line carries the
    # per-rank kernel times and the check that the gathered tensor holds each rank's solution at its place
    assert rec["config"]["distinct_problems"] == 256
    assert len(rec["per_rank"]["kernel_ms"]) == 2 and rec["per_rank"]["gathered_rows_equal_local_solution"] is True
    assert rec["per_rank"]["kernel_ms_min"] <= rec["per_rank"]["kernel_ms_max"] and rec["same_batch_ms_per_step"] > 0


def test_closed_loop_ticks_against_independent_scipy_solutions(solver):
    """HIP solver against the independent SLSQP solutions of 15 warm-started closed-loop ticks (fixture g8_scipy_ticks: segment
    switches, omega unwrap, active +-0.01 tube, phi_max active; see tests/test_oracle_golden.py)."""
    d = np.load(os.path.join(G, "g8_scipy_ticks.npz"))
    out = solver.solve_host(d["p"], d["x0"])
    assert (out["status"] == 0).all()
    dq = (out["x"] - d["x"]).reshape(-1, 10, 44)[:, :, 8:15]
    rms = np.sqrt(np.mean(dq ** 2, axis=(1, 2)))
    assert rms.max() < 5e-6 and np.median(rms) < 2e-7, rms
    assert np.abs(out["f"] - d["f"]).max() < 1e-7


def test_benchmark_batches_against_independent_slsqp_solutions():
    """HIP kernels against the independent SLSQP solutions of samples of the benchmark batches (fixture g8_scipy_batch: configs[1] seed 0 cold
    starts; tight N = 20 / N = 30 from a start near the oracle's solution) -- tests/test_oracle_golden.py holds the criteria."""
    import torch
    from boundmpc_amd import BatchedOCPSolver
    from tests.test_oracle_golden import check_against_slsqp_batch

    def solve(P, X0, N):
        s = BatchedOCPSolver(N, 4, 0.1)
        try:
            s.set_team_waves(1)      # the kernel of the benchmark batches (one wave per problem); teams are compared with it in tests/test_gpu_team.py
            o = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X0, device="cuda"), out={}, want=("f", "status"))
            return {k: v.cpu().numpy() for k, v in o.items()}
        finally:
            s.close()
    rep = check_against_slsqp_batch(solve, "HIP kernel")
    assert rep["c1"]["same_minimiser"] >= 60


def test_kkt_certificate_with_the_references_own_derivatives(solver):
    """HIP solutions of recorded closed-loop ticks (cold start, segment switch, end of path, experiment 2's tight tube) are KKT points of
    the REFERENCE's NLP: stationarity with the gradient / Jacobian obtained by a complex step through the reference's own builder
    (fixture G9) and the kernel's multipliers."""
    from tests.test_oracle_golden import _g9_solution_cases, _kkt_residual_with_reference_derivatives
    cases = _g9_solution_cases()
    assert len(cases) >= 8
    for i, j, which, tick in cases:
        d = np.load(os.path.join(G, f"g7_closedloop_exp{which}.npz"))
        out = solver.solve_host(d["p"][tick], d["x0"][tick])
        assert out["status"][0] == 0
        r = _kkt_residual_with_reference_derivatives(i, j, out["x"][0], out["lam_g"][0], out["lam_x"][0])
        assert r < 2e-4, (which, tick, r)
        assert (out["lam_g"][0].reshape(10, 43)[:, 36:] >= 0).all()

def test_horizon_limits_of_the_handle():
    """bmpc_create accepts N = 1..40 (BoundMPC.py:35 makes the horizon a parameter), and so do the stream entry points since round 4 (their
    closed loops at N = 36: tests/test_gpu_stream.py); malformed calls are refused before anything is launched."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, BoundMPCHipError
    with pytest.raises(BoundMPCHipError):
        BatchedOCPSolver(41, 4, 0.1)
    with pytest.raises(BoundMPCHipError):
        BatchedOCPSolver(10, 7, 0.1)      # nr_segs: 2..6 (BoundMPC.py:62 makes it a parameter)
    s = BatchedOCPSolver(40, 4, 0.1)
    try:
        from boundmpc_amd import workload
        P, X, _ = workload.make_batch(8, seed=3, N=40)
        o = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda"))
        assert int((o["status"] == 0).sum()) == 8
        import ctypes
        buf = torch.zeros(64, dtype=torch.float64, device="cuda"); vp = ctypes.c_void_p(buf.data_ptr())
        rc = s._lib.bmpc_stream_pack(s._h, 1, vp, 4, vp, vp, vp, vp, None, None)      # a path table shorter than the window (S + 1 entries): refused before anything is launched
        assert rc == 1, rc      # BMPC_ERR_ARG
        rc = s._lib.bmpc_stream_pack(s._h, 1, None, 5, vp, vp, vp, vp, None, None)    # missing buffer: refused
        assert rc == 1, rc
        lens = [ctypes.c_int() for _ in range(4)]
        assert s._lib.bmpc_stream_lengths(s._h, *[ctypes.byref(v) for v in lens]) == 0 and lens[1].value == 32 + 56 * 40 + 2
    finally:
        s.close()

@pytest.mark.parametrize("N,S", [(10, 5), (10, 6), (20, 6)])
def test_five_and_six_path_segments_against_the_oracle(N, S):
    """nr_segs > 4: the parameter vector no longer fits the 512-double LDS window; its tail lives in the iterate area and the iterate in
    the workspace (bmpc_wave.inl make_poff_lds).  256 random problems against the oracle, problem by problem."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    B = 256
    P, X, _ = workload.make_batch(B, seed=41 + S, N=N, S=S)
    assert P.shape[1] == 141 + 91 * S
    s = BatchedOCPSolver(N, S, 0.1)
    try:
        o = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda"))
        st, it, x = o["status"].cpu().numpy(), o["iters"].cpu().numpy(), o["x"].cpu().numpy()
    finally:
        s.close()
    ref = c_oracle.solve(P, X, N, S, 0.1, nthreads=16)
    assert (st == ref["status"]).all() and (st == 0).all()
    d = (x - ref["x"]).reshape(-1, N, 44)[:, :, 8:15]
    per = np.sqrt((d ** 2).mean(axis=(1, 2)))
    assert (per > 1e-6).sum() <= (1 if N > 11 else 0), float(per.max())
    assert np.sqrt((d[per <= 1e-6] ** 2).mean()) < 1e-7
    assert (np.abs(it - ref["iters"]) <= 2).mean() >= 0.97


@pytest.mark.gpu
def test_first_failures_of_the_closed_loops_g13b_on_the_gpu():
    """Fixture g13b: every first failing tick of the 256 closed loops of configs[4] as round 4 ran them (38 problems: 28 infeasible for SLSQP, 10
    feasible).  Round 5, restoration phase (include/boundmpc_hip.h bmpc_set_restoration), at the handle's DEFAULTS, on every launch shape -- one wave
    per problem, teams and pairs; the batch kernels hand the jammed problems to the restoration kernel (bmpc_resto.hip): the 28 locally infeasible
    problems end as status 2 within 60 iterations, 8 of the 10 feasible ones converge within 135, every status equals the oracle's, the objectives
    of the converged ones agree to 1e-9 relative.  With the phase switched off the kernels stall on all 38 like round 4's."""
    import torch
    from boundmpc_amd import BatchedOCPSolver
    from oracle import c_oracle
    d = np.load(os.path.join(G, "g13b_first_failures_256_streams.npz"))
    p, x0 = torch.tensor(d["p"], device="cuda"), torch.tensor(d["x0"], device="cuda")
    feas = (d["slsqp_eq"] < 1e-8) & (d["slsqp_ineq"] < 1e-8) & (d["slsqp_bounds"] < 1e-8)
    ref = c_oracle.solve(d["p"], d["x0"], 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=500, start_rollout=0), nthreads=8)
    for waves in (1, 4, 2):      # one wave per problem, teams, pairs
        s = BatchedOCPSolver(10, 4, 0.1, start_rollout=False); s.set_team_waves(waves)      # (x0 as given: in the loop these ticks are warm solves)
        o = s.solve_batch(p, x0); st, it, f = o["status"].cpu().numpy(), o["iters"].cpu().numpy(), o["f"].cpu().numpy()
        di = np.abs(it - ref["iters"])
        assert np.array_equal(st, ref["status"]) and di[st == 0].max() <= 8 and (di <= 8).sum() >= 36, (waves, st, it)      # (the two that fail after three phases may leave at different counts)
        assert (st[~feas] == 2).all() and it[~feas].max() <= 60
        conv = feas & (st == 0)
        assert conv.sum() >= 8 and it[conv].max() <= 135
        assert np.abs(f[conv] - ref["f"][conv]).max() < 1e-9 * np.abs(ref["f"][conv]).max()
        dq = (o["x"].cpu().numpy()[conv] - ref["x"][conv]).reshape(-1, 10, 44)[:, :, 8:15]
        assert np.sqrt((dq ** 2).mean(axis=(1, 2))).max() < 1e-5      # (ill-conditioned problems: multipliers of 1e6)
        s.set_restoration(False)
        o = s.solve_batch(p, x0); st = o["status"].cpu().numpy()
        assert (st != 0).all() and (st == d["oracle_status"]).mean() >= 0.9      # (a stalled solve may end as status 2 in one and 3 in the other)
        s.close()
    # the hand-over must not depend on the caller's optional outputs: status / iters NULL (handle-owned stand-ins)
    s = BatchedOCPSolver(10, 4, 0.1, start_rollout=False)
    x = torch.empty((38, 440), device="cuda", dtype=torch.float64)
    from boundmpc_amd import _lib
    _lib.check(s._lib.bmpc_solve_batch(s._h, 38, p.data_ptr(), x0.data_ptr(), x.data_ptr(), None, None, None, None, None, None, None, None), "bmpc_solve_batch")
    torch.cuda.synchronize()
    o = s.solve_batch(p, x0)
    assert torch.equal(x, o["x"])
    s.close()


# ---- the node's control loop (bound_mpc_node.py:48-83,292-401) over the HIP solver ----
@pytest.mark.gpu
@pytest.mark.parametrize("which", [1, 2])
def test_node_loop_over_the_hip_solver_retraces_g7(which):
    """NodeLoop (boundmpc_amd/node_loop.py: reset + step of the reference's MPCNode without rclpy) with the DEFAULT solver -- the GPU library behind
    NlpSolverShim, what a maintainer gets by swapping the import -- runs the whole closed loop of the reference's experiment until the goal
    (phi_max - phi <= 0.01) and retraces fixture G7 (the reference's own BoundMPC.step() driven tick by tick, tests/golden/make_golden.py): plant
    state to 1e-6 rad on every tick, iteration counts within 1, no fallback tick, and the per-tick dict carries every field of the MPCData message
    the node fills (bound_mpc_node.py:169-289)."""
    from boundmpc_amd import workload
    from boundmpc_amd.node_loop import NodeLoop
    d6 = np.load(os.path.join(G, f"g6_pack_exp{which}_tick0.npz")); d7 = np.load(os.path.join(G, f"g7_closedloop_exp{which}.npz"))
    mk = lambda k: [np.array(v) for v in d6[k]]
    published = []
    loop = NodeLoop(mk("p_via"), mk("r_via"), [mk("p_lower"), mk("p_upper")], [mk("r_lower"), mk("r_upper")], mk("bp1_in"), mk("br1_in"), list(d6["s_in"]),
                    list(d6["e_p_min_in"]), list(d6["e_r_min_in"]), list(d6["e_p_max_in"]), list(d6["e_r_max_in"]), p0=d6["p0fk"].copy(), q0=d7["q"][0],
                    params=workload.Params(weights=d6["weights_f64"], real_time=True, build=True), solver=None, publish=published.append)
    assert type(loop.mpc.solver).__name__ == "NlpSolverShim"
    T = len(d7["q"])
    worst = 0.0
    for t in range(T - 1):
        worst = max(worst, float(np.abs(loop.q - d7["q"][t]).max()), float(np.abs(loop.dq - d7["dq"][t]).max()) * 0.1)
        out = loop.step()
        assert out is not None and loop.mpc.error_count == int(d7["error_count"][t]) == 0
        assert abs(published[-1]["iterations"] - int(d7["iters"][t])) <= 1
        np.testing.assert_allclose(out[0]["q"], d7["traj_q"][t][:, :out[0]["q"].shape[1]], atol=1e-6)
    assert worst <= 1e-6, worst
    assert abs(loop.mpc.phi_current[0] - d7["phi_current"][T - 2]) < 1e-6 and len(loop.t_switch) == int(d7["sector"][T - 2])
    fields = {"stamp", "sector", "phi_switch_vector", "t_comp", "t_loop", "t_overhead", "iterations", "t_switch", "phi_switch", "fails", "phi", "dphi", "ddphi",
              "dddphi", "phi_max", "p", "v", "a", "q", "dq", "ddq", "dddq"}
    assert set(published[-1].keys()) == fields and len(published) == T - 1 and sum(published[-1]["fails"]) == 0
    loop.mpc.batched.close()


@pytest.mark.gpu
def test_bad_warm_starts_are_rescued_by_the_restoration_phase_on_the_gpu():
    """256 feasible tight N = 10 problems started from a warm start with noise 0.3 on every variable (far off its own dynamics): with the restoration
    phase (rollout of the iterate's jerks, elastic feasibility problem, feasible point handed back to the main phase) >= 97 % converge at the
    handle's defaults, on both launch shapes, with the oracle's statuses; with the phase switched off (round 4) a quarter does."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    P, X, _ = workload.make_batch(256, seed=50, N=10, tight=True)
    X2 = X + np.random.default_rng(5).normal(size=X.shape) * 0.3
    # (start_rollout off: x0 as given -- with the default these starts are rolled out before the first iteration and never need the phase; last lines)
    ref = c_oracle.solve(P, X2, 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=300, start_rollout=0), nthreads=8)
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X2, device="cuda")
    for waves in (1, 4):
        s = BatchedOCPSolver(10, 4, 0.1, max_iter=300, start_rollout=False); s.set_team_waves(waves)
        o = s.solve_batch(p, x0); st, it = o["status"].cpu().numpy(), o["iters"].cpu().numpy()
        assert (st == 0).mean() >= 0.97 and (st == ref["status"]).mean() >= 0.99 and np.abs(it - ref["iters"])[st == ref["status"]].max() <= 10, (waves, np.bincount(st), np.abs(it - ref["iters"]).max())
        ok = (st == 0) & (ref["status"] == 0)
        dq = (o["x"].cpu().numpy()[ok] - ref["x"][ok]).reshape(-1, 10, 44)[:, :, 8:15]
        assert np.sqrt((dq ** 2).mean(axis=(1, 2))).max() < 1e-5
        s.set_restoration(False)
        o = s.solve_batch(p, x0)
        assert (o["status"].cpu().numpy() == 0).mean() <= 0.4
        # the handle's defaults: the cold start that is not a trajectory is rolled out first -- all converge, with or without the phase, in the oracle's iterations
        s.set_start_rollout(True)
        ref1 = c_oracle.solve(P, X2, 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=300, restoration=0), nthreads=8)
        o = s.solve_batch(p, x0); st, it = o["status"].cpu().numpy(), o["iters"].cpu().numpy()
        assert (st == 0).all() and (ref1["status"] == 0).all() and it.mean() <= 20 and np.abs(it - ref1["iters"]).max() <= 3, (waves, np.bincount(st), it.mean())
        dq = (o["x"].cpu().numpy() - ref1["x"]).reshape(-1, 10, 44)[:, :, 8:15]
        assert np.sqrt((dq ** 2).mean(axis=(1, 2))).max() < 1e-5
        s.close()


@pytest.mark.gpu
def test_long_horizon_restoration_as_the_last_resort_behind_the_restarts():
    """configs[3] (N = 30, tight tubes, seed 2): problems 690 and 695 of the batch are two of those whose first attempt ends as status 2 after its two
    barrier restarts.  With the restoration phase switched on for this long-horizon handle (and the second attempt off) the phase follows the last
    restart (never a jam): both run on -- 690 for more than a hundred iterations -- and end as the oracle says; the other problems of the window are
    untouched (the phase is behind the restarts).  Mode 1 on a long horizon runs the whole batch in the restoration instantiation (round 6)."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    P, X, _ = workload.make_batch(8192, seed=2, N=30, tight=True, rows=(688, 696))
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    s = BatchedOCPSolver(30, 4, 0.1)
    assert s.get_restoration()["mode"] == 2      # default of long horizons: after a numerical breakdown only
    dflt = s.solve_batch(p, x0, out={}); st_d = dflt["status"].cpu().numpy()
    assert s.get_second_attempt() == 100 and (st_d == 0).all()      # round 6: with the second attempt of the defaults both converge (test_long_horizon_tight_tubes)
    s.set_second_attempt(0)                                          # below: the first attempt alone, and what the restoration phase makes of it
    off = s.solve_batch(p, x0); st_off, it_off = off["status"].cpu().numpy(), off["iters"].cpu().numpy()
    assert st_off[2] == 2 and st_off[7] == 2 and (np.delete(st_off, [2, 7]) == 0).all()
    s.set_restoration(True)
    on = s.solve_batch(p, x0); st_on, it_on = on["status"].cpu().numpy(), on["iters"].cpu().numpy()
    ref = c_oracle.solve(P, X, 30, 4, 0.1, opts=c_oracle.default_opts(mu_init=3.0, slack_push=0.1, stall_window=20, restoration=1), nthreads=8)
    assert np.array_equal(st_on, ref["status"]) and it_on[2] > it_off[2] + 100 and it_on[7] > it_off[7] and np.abs(it_on - ref["iters"]).max() <= 25, (st_on, it_on, ref["iters"])
    keep = np.delete(np.arange(8), [2, 7])
    assert np.array_equal(it_on[keep], it_off[keep]) and torch.equal(on["x"][torch.tensor(keep, device="cuda")], off["x"][torch.tensor(keep, device="cuda")])
    s.close()


@pytest.mark.gpu
def test_numerical_breakdowns_of_far_off_starts_go_to_the_restoration_phase_on_the_gpu():
    """64 loose N = 20 problems started with noise 0.3 on every variable of the warm start.  Mode 0 (round 4): a fifth ends as status 3 (dual residual
    beyond 1e12).  The default of long horizons (mode 2: restoration after a numerical breakdown only) hands those to the restoration kernel -- rollout,
    feasibility problem, main phase again -- and all 64 converge, with the oracle's statuses; the restart count of the long horizon travels with the
    hand-over."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    P, X, _ = workload.make_batch(64, seed=7, N=20)
    X2 = X + np.random.default_rng(3).normal(size=X.shape) * 0.3
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X2, device="cuda")
    ref = c_oracle.solve(P, X2, 20, 4, 0.1, opts=c_oracle.default_opts(mu_init=3.0, slack_push=0.1, stall_window=20, restoration=2, start_rollout=0, retry_cap=100), nthreads=8)
    s = BatchedOCPSolver(20, 4, 0.1, start_rollout=False)      # (x0 as given, so that the main phase does break down; the default rolls these starts out first)
    o = s.solve_batch(p, x0); st = o["status"].cpu().numpy()
    assert (st == 0).all() and np.array_equal(st, ref["status"]) and float(o["kkt"].max()) <= 1e-8
    s.set_restoration(0)
    o0 = s.solve_batch(p, x0); st0 = o0["status"].cpu().numpy()
    assert (st0 == 3).sum() >= 8 and (st0 == 0).sum() <= 56
    s.set_start_rollout(True); s.set_restoration(2)      # the handle's defaults: rolled out first (one problem still breaks down and goes to the phase), a third of the iterations
    ref1 = c_oracle.solve(P, X2, 20, 4, 0.1, nthreads=8)
    o1 = s.solve_batch(p, x0); st1, it1 = o1["status"].cpu().numpy(), o1["iters"].cpu().numpy()
    assert (st1 == 0).all() and np.array_equal(st1, ref1["status"]) and it1.mean() <= 0.5 * o["iters"].float().mean().item() and np.abs(it1 - ref1["iters"]).max() <= 8
    s.close()


@pytest.mark.gpu
def test_the_solver_does_not_depend_on_the_reference_warm_start():
    """128 feasible N = 10 problems (loose tubes, seed 60; tests/gpu_robustness.py is the full battery) started from all zeros, from uniform(-1, 1) noise in
    place of the reference's cold start and from that cold start + Gaussian noise 1.0 on every variable.  At the handle's defaults (a cold start that
    is not a trajectory is rolled out before the first iteration; restoration phase behind it) ALL converge, from zeros in the iterations of the
    reference's cold start, to the solutions of the oracle with the oracle's statuses problem by problem.  With x0 taken as given the restoration
    phase alone rescues >= 94 % / 99 % of them in about three times the iterations; with neither (round 4's solver) at most a quarter converges from
    zeros and none from the noisy starts."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    P, X, _ = workload.make_batch(128, seed=60, N=10)
    rng = np.random.default_rng(11)
    starts = {"zeros": (np.zeros_like(X), 0.94, 0.25), "uniform": (rng.uniform(-1, 1, X.shape), 0.99, 0.02), "noise 1.0": (X + rng.normal(size=X.shape), 0.99, 0.02)}
    s = BatchedOCPSolver(10, 4, 0.1)
    p = torch.tensor(P, device="cuda")
    cold = s.solve_batch(p, torch.tensor(X, device="cuda"))
    for name, (X0, lo, hi_off) in starts.items():
        x0 = torch.tensor(X0, device="cuda")
        for roll in (True, False):
            ref = c_oracle.solve(P, X0, 10, 4, 0.1, opts=c_oracle.default_opts(start_rollout=int(roll)), nthreads=8)
            s.set_restoration(1); s.set_start_rollout(roll)
            o = s.solve_batch(p, x0); st, it = o["status"].cpu().numpy(), o["iters"].cpu().numpy()
            assert (st == 0).mean() >= (1.0 if roll else lo) and np.array_equal(st, ref["status"]) and np.abs(it - ref["iters"]).max() <= 6, (name, roll, np.bincount(st))
            ok = st == 0
            dq = (o["x"].cpu().numpy()[ok] - ref["x"][ok]).reshape(-1, 10, 44)[:, :, 8:15]
            assert np.sqrt((dq ** 2).mean(axis=(1, 2))).max() < 1e-5, (name, roll)
            if roll:
                assert it.mean() <= 22, (name, it.mean())
                if name == "zeros":      # all zeros rolled out IS the reference's cold start (up to round-off: FK(q0) in place of the given p0)
                    assert np.abs(it - cold["iters"].cpu().numpy()).max() <= 1 and (o["x"] - cold["x"]).abs().reshape(-1, 10, 44)[:, :, 8:15].max().item() < 1e-6
        s.set_restoration(0)
        o0 = s.solve_batch(p, x0)
        assert (o0["status"].cpu().numpy() == 0).mean() <= hi_off, name
    s.close()


@pytest.mark.gpu
def test_far_off_cold_starts_of_other_sizes():
    """The rollout of a cold start that is not a trajectory on the other instantiations: the recorded starts of fixture G12 -- (N, S) = (5, 2), (8, 3),
    (6, 5), (12, 6): the iterate in the workspace for S > 4 / N > 11 -- and synthetic batches of N = 1, 2 and 40 stages, each with noise 0.3 and 1.0
    (0.5) on every variable: every problem converges, with the oracle's status and (within 2) its iterations."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    from oracle import c_oracle
    d = np.load(os.path.join(G, "g12_pack_other_sizes.npz"))
    cases = []
    for key, N, S in (("n5s2", 5, 2), ("n8s3", 8, 3), ("n6s5", 6, 5), ("n12s6", 12, 6)):
        P = np.where(np.isfinite(d[key + "_p"]), d[key + "_p"], 0.0)
        for nz in (0.3, 1.0):
            cases.append((N, S, float(d[key + "_dt"]), P, d[key + "_x0"] + np.random.default_rng(3).normal(size=d[key + "_x0"].shape) * nz))
    for N in (1, 2, 40):
        P, X, _ = workload.make_batch(32, seed=9, N=N)
        cases.append((N, 4, 0.1, P, X + np.random.default_rng(4).normal(size=X.shape) * 0.5))
    for N, S, dt, P, X0 in cases:
        ref = c_oracle.solve(P, X0, N, S, dt, nthreads=8)
        s = BatchedOCPSolver(N, S, dt)
        o = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X0, device="cuda")); st, it = o["status"].cpu().numpy(), o["iters"].cpu().numpy()
        assert (st == 0).all() and np.array_equal(st, ref["status"]) and np.abs(it - ref["iters"]).max() <= 2, (N, S, np.bincount(st), np.abs(it - ref["iters"]).max())
        s.close()

