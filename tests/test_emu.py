"""The wave program (boundmpc_amd/csrc/bmpc_wave.inl -- the text hipcc compiles for gfx950) executed by the
TEST-ONLY CPU lane emulator (tests/emu) against the C oracle.  Checks the kernel's indexing and phase structure
without a GPU; different lane execution orders must give bit-identical results (no intra-phase cross-lane
dependence).  The product never loads the emulator."""
import os

import numpy as np
import pytest

from boundmpc_amd import workload
from oracle import c_oracle
from tests.emu import emu

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("which", [1, 2])
def test_emu_matches_oracle_tick0_all_lane_orders(which):
    d = np.load(os.path.join(G, f"g6_pack_exp{which}_tick0.npz"))
    ref = c_oracle.solve(d["p_f64"], d["x0_f64"], 10, 4, 0.1)
    outs = [emu.solve(d["p_f64"], d["x0_f64"], 10, 4, 0.1, lane_order=o) for o in (0, 1, 2)]
    for o in outs:
        assert o["status"][0] == 0 and abs(int(o["iters"][0]) - int(ref["iters"][0])) <= 1
        # both sides stop at the first iterate with KKT error <= 1e-8: their last iterates agree to O(tol) (the weakly determined jerks least)
        np.testing.assert_allclose(o["x"], ref["x"], atol=2e-8)
        np.testing.assert_allclose(o["g"], ref["g"], atol=1e-9)      # g is evaluated at iterates that agree to 1e-10 (dg/dx = O(1..10))
        np.testing.assert_allclose(o["lam_g"], ref["lam_g"], atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(o["lam_x"], ref["lam_x"], atol=1e-8)
    for k in ("x", "g", "lam_g", "lam_x", "f"):
        np.testing.assert_array_equal(outs[0][k], outs[1][k])
        np.testing.assert_array_equal(outs[0][k], outs[2][k])


def test_emu_closed_loop_ticks():
    d = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))
    out = emu.solve(d["p"], d["x0"], 10, 4, 0.1, nthreads=4)
    assert (out["status"] == 0).all()
    assert np.abs(out["iters"] - d["iters"]).max() <= 1
    rms = np.sqrt(np.mean((out["x"] - d["x"]).reshape(-1, 10, 44)[:, :, 8:15] ** 2))
    assert rms < 1e-7   # a converged-at-threshold tick may differ by one Newton iteration (|dx| ~ 4e-7)
    d2 = np.load(os.path.join(G, "g7_closedloop_exp2.npz"))
    out2 = emu.solve(d2["p"], d2["x0"], 10, 4, 0.1, nthreads=4)
    assert (out2["status"] == 0).all()
    assert np.sqrt(np.mean((out2["x"] - d2["x"]).reshape(-1, 10, 44)[:, :, 8:15] ** 2)) < 1e-7


def test_emu_random_batch_and_long_horizon():
    P, X, _ = workload.make_batch(48, seed=7, workers=1)
    ref = c_oracle.solve(P, X, 10, 4, 0.1, nthreads=4)
    out = emu.solve(P, X, 10, 4, 0.1, nthreads=4)
    assert np.abs(out["iters"] - ref["iters"]).max() <= 1
    np.testing.assert_array_equal(out["status"], ref["status"])
    assert np.sqrt(np.mean((out["x"] - ref["x"]).reshape(-1, 10, 44)[:, :, 8:15] ** 2)) < 1e-7      # joints, rad RMS
    # N = 30, tight tubes (config 4 of BASELINE.json), small sample
    P, X, _ = workload.make_batch(6, seed=2, N=30, tight=True, workers=1)
    ref = c_oracle.solve(P, X, 30, 4, 0.1, nthreads=4)
    out = emu.solve(P, X, 30, 4, 0.1, nthreads=4)
    np.testing.assert_array_equal(out["status"], ref["status"])
    ok = ref["status"] == 0
    assert ok.any()
    assert np.abs(out["iters"] - ref["iters"]).max() <= 2
    dd = (out["x"][ok] - ref["x"][ok]).reshape(-1, 30, 44)
    assert np.sqrt(np.mean(dd[:, :, 8:15] ** 2)) < 1e-6      # joint trajectory (rad RMS)
    assert np.sqrt(np.mean(dd[:, :, :8] ** 2)) < 1e-3        # jerks are weakly determined (w_jerk = 1e-4) and O(35)


def test_emu_gauss_newton_and_iteration_cap():
    d = np.load(os.path.join(G, "g6_pack_exp1_tick0.npz"))
    o = emu.solve(d["p_f64"], d["x0_f64"], 10, 4, 0.1, emu.default_opts(max_iter=3))
    assert o["status"][0] == 1 and o["iters"][0] == 3
    ogn = emu.solve(d["p_f64"], d["x0_f64"], 10, 4, 0.1, emu.default_opts(exact_hessian=0))
    ref = c_oracle.solve(d["p_f64"], d["x0_f64"], 10, 4, 0.1, c_oracle.default_opts(exact_hessian=0))
    assert ogn["status"][0] == ref["status"][0] and abs(int(ogn["iters"][0]) - int(ref["iters"][0])) <= 1


def _replay(solve, d, ticks, N=10, **kw):
    """Warm-started replay of the first closed-loop ticks of a fixture: one stream, dual state carried and shifted."""
    st = np.zeros((1, c_oracle.state_len(N)))
    xs, its, sts = [], [], []
    for t in range(ticks):
        if t:
            nu = st[0, :N * 57].reshape(N, 57); nu[:-1] = nu[1:].copy()      # horizon advanced by one stage
        o = solve(d["p"][t], d["x0"][t], N, 4, 0.1, state=st, **kw)
        xs.append(o["x"][0].copy()); its.append(int(o["iters"][0])); sts.append(st.copy())
    return np.array(xs), np.array(its), np.array(sts)


def test_emu_warm_started_stream_matches_oracle():
    d = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))
    T = 8
    xo, io, so = _replay(c_oracle.solve, d, T)
    xe, ie, se = _replay(emu.solve, d, T)
    assert np.abs(ie - io).max() <= 1
    assert np.sqrt(np.mean((xe - xo).reshape(T, 10, 44)[:, :, 8:15] ** 2)) < 1e-7
    # a converged warm-started solve lands on the same minimiser as the cold-started fixture solve
    assert np.sqrt(np.mean((xo - d["x"][:T]).reshape(T, 10, 44)[:, :, 8:15] ** 2)) < 1e-6
    assert so[-1][0, -2] > 0 and se[-1][0, -1] == ie[-1]                     # mu stored, iteration count stored
    # the multipliers carried to the next tick agree where they matter (active rows)
    act = so[-1][0, :570] > 1e-3
    np.testing.assert_allclose(se[-1][0, :570][act], so[-1][0, :570][act], rtol=1e-5)


def test_emu_real_time_iteration_stream_matches_oracle():
    """Two Newton steps per tick, state and iterate carried over: status 1 every tick, emulator == oracle."""
    d = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))
    T = 6
    xo, io, so = _replay(c_oracle.solve, d, T, opts=c_oracle.default_opts(max_iter=2))
    xe, ie, se = _replay(emu.solve, d, T, opts=emu.default_opts(max_iter=2))
    assert (io == 2).all() and (ie == 2).all()
    np.testing.assert_allclose(xe, xo, atol=1e-8)
    np.testing.assert_allclose(se[:, 0, :571], so[:, 0, :571], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("S,N", [(2, 5), (3, 12)])
def test_emu_other_window_sizes_and_horizons(S, N):
    """nr_segs != 4 changes the parameter layout (n_p = 141 + 91 S); N > 11 takes the global-memory iterate path."""
    P, X, _ = workload.make_batch(4, seed=4, N=N, S=S, workers=1)
    assert P.shape[1] == 141 + 91 * S
    ref = c_oracle.solve(P, X, N, S, 0.1, nthreads=2)
    out = emu.solve(P, X, N, S, 0.1, nthreads=2)
    assert (ref["status"] == 0).all() and (out["status"] == 0).all()
    assert np.abs(out["iters"] - ref["iters"]).max() <= 1
    assert np.sqrt(np.mean((out["x"] - ref["x"]).reshape(-1, N, 44)[:, :, 8:15] ** 2)) < 1e-7


@pytest.mark.parametrize("N", [1, 2, 3, 10])
def test_emu_newton_direction_equals_oracle_dense_solve(N):
    """One Newton direction at a random interior point: the wave program's block-Riccati (gains, feed-forward, dZ) against the
    oracle's dense per-stage Riccati on the same QP -- isolates the linear algebra from the globalisation."""
    import ctypes
    from oracle import nlp
    L, CO = emu.lib(), c_oracle.lib()
    d = np.load(os.path.join(G, "g6_pack_exp1_tick0.npz"))
    p, x0f = d["p_f64"], d["x0_f64"]
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    names = "Z ZT T TT NUm LAM G GT HIN HT DZ DT DNU GH GVP RJ KIN REF KT KF RDY AES RLV SG TI SR size".split()
    rng = np.random.default_rng(3 + N)
    S, h = 4, 0.1
    offs = (ctypes.c_int * len(names))(); L.bmpc_emu_scr_offsets(N, offs); off = dict(zip(names, list(offs)))
    x = x0f[:44 * N] + rng.normal(size=44 * N) * 0.02
    x.reshape(N, 44)[:, 41] = np.linspace(0.3, 1.0, N)
    hin = nlp.internal_ineq(x, p, N, S)
    t = np.maximum(-hin, 1e-2); mu = 0.1; nu = mu / t * rng.uniform(0.5, 2, t.shape); delta = 10.0 if N < 10 else 1000.0
    scr = np.zeros(off["size"]); lds = np.zeros(L.bmpc_emu_lds_doubles())
    o = emu.default_opts()
    rc = L.bmpc_emu_newton(N, S, ctypes.c_double(h), ctypes.byref(o), P(p), P(x), P(t), P(nu), ctypes.c_double(mu), ctypes.c_double(delta), P(scr), P(lds))
    Qt = np.zeros((N, 35, 35)); qt = np.zeros((N, 35)); Xt = np.zeros((N, 35, 35)); A = np.zeros((N, 35, 43)); rd = np.zeros((N, 35))
    T = np.zeros((N, 44, 35)); rl = np.zeros((N, 44)); Kg = np.zeros((N, 8, 35)); kff = np.zeros((N, 8)); dZ = np.zeros(N * 44)
    rc2 = CO.bmpc_oracle_debug_qp(N, S, ctypes.c_double(h), P(p), P(x), P(t), P(nu), ctypes.c_double(mu), 1, ctypes.c_double(delta), P(Qt), P(qt),
                                  P(Xt), P(A), P(rd), P(T), P(rl), P(Kg), P(kff), P(dZ))
    assert rc == 0 and rc2 == 0
    KT = scr[off["KT"]:off["KT"] + N * 35 * 8].reshape(N, 8, 35)      # control-major since round 3
    np.testing.assert_allclose(KT, Kg, atol=1e-12)
    np.testing.assert_allclose(scr[off["KF"]:off["KF"] + 8 * N], kff.ravel(), atol=1e-12)
    np.testing.assert_allclose(scr[off["DZ"]:off["DZ"] + 44 * N], dZ, atol=1e-12)


def test_infeasible_problem_ends_as_status_2_in_oracle_and_emulator():
    """phi_max < 0 <= phi (casadi_ocp_formulation.py:307-310 against the bound phi >= 0, :145-147): no feasible point.  Round 5: the main phase
    jams, the restoration phase converges to a point whose violation is not zero, and the solve ends with status 2 at the same iteration in both
    builds ("failure is data", BoundMPC.py:465-489) -- after 9 and 26 iterations; without the restoration phase the stall test ends it (status 2 at
    a multiple of 20 >= 40); with both switched off the solve runs into the iteration cap (status 1)."""
    d = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))
    p, x0 = d["p"][:2].copy(), d["x0"][:2].copy()
    p[:, 460] = -1.0
    o, e = c_oracle.solve(p, x0, 10, 4, 0.1), emu.solve(p, x0, 10, 4, 0.1)
    assert (o["status"] == 2).all() and (e["status"] == 2).all()
    assert (o["iters"] == e["iters"]).all() and (o["iters"] <= 40).all()
    assert np.isfinite(o["x"]).all() and np.isfinite(e["x"]).all()
    o, e = c_oracle.solve(p, x0, 10, 4, 0.1, c_oracle.default_opts(restoration=0)), emu.solve(p, x0, 10, 4, 0.1, emu.default_opts(restoration=0))
    assert (o["status"] == 2).all() and (e["status"] == 2).all()
    assert (o["iters"] == e["iters"]).all() and (o["iters"] % 20 == 0).all() and (o["iters"] >= 40).all()
    off = c_oracle.solve(p, x0, 10, 4, 0.1, c_oracle.default_opts(stall_window=0, max_iter=90, restoration=0))
    assert (off["status"] == 1).all() and (off["iters"] == 90).all()


def test_no_solve_reads_what_it_has_not_written():
    """LDS and the workspace slab are reused from problem to problem on the GPU.  With both filled with NaN before every problem
    (BMPC_EMU_POISON) every output must stay bit-identical: cold solves, evaluation only (max_iter 0), capped Gauss-Newton ticks,
    warm-started ticks, the long-horizon instantiation."""
    from boundmpc_amd import workload
    d = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))

    def run():
        outs = []
        P, X, _ = workload.make_batch(8, seed=0, N=10)
        outs.append(emu.solve(P, X, 10, 4, 0.1, nthreads=4))
        outs.append(emu.solve(P, X, 10, 4, 0.1, emu.default_opts(max_iter=0), nthreads=4))
        outs.append(emu.solve(P, X, 10, 4, 0.1, emu.default_opts(exact_hessian=0, tol=1e-3, max_iter=4), nthreads=4))
        st = np.zeros((1, c_oracle.state_len(10)))
        for t in range(4):
            outs.append(emu.solve(d["p"][t:t + 1], d["x0"][t:t + 1], 10, 4, 0.1, emu.default_opts(tol=1e-3, max_iter=4, mu_warm=3e-2), state=st, nthreads=1))
            nu = st[0, :570].reshape(10, 57); nu[:-1] = nu[1:].copy()
        P3, X3, _ = workload.make_batch(2, seed=2, N=30, tight=True)
        outs.append(emu.solve(P3, X3, 30, 4, 0.1, nthreads=2))
        return outs
    plain = run()
    os.environ["BMPC_EMU_POISON"] = "1"
    try:
        poisoned = run()
    finally:
        del os.environ["BMPC_EMU_POISON"]
    for a, b in zip(plain, poisoned):
        for k in ("x", "g", "lam_g", "lam_x", "f", "kkt", "iters", "status"):
            assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_bounded_range_sincos_of_the_kinematics_against_libm():
    """bmpc_sincos (csrc/bmpc_wave.inl): two-constant Cody-Waite reduction + the classical kernels on [-pi/4, pi/4], used for the joint
    angles of the kinematic chain instead of the full-range library call.  Against libm: every quadrant, the reduction boundaries,
    and far outside the joint range."""
    import ctypes
    emu.build()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-3.3, 3.3, 20000), rng.uniform(-50, 50, 20000), np.arange(-8, 9) * np.pi / 4, np.arange(-8, 9) * np.pi / 4 + 1e-9,
                        np.array([0.0, 1e-300, -1e-300, 1e-9, 2.9670597283903604, -2.9670597283903604])])
    s, c = np.zeros_like(x), np.zeros_like(x)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    emu.lib().bmpc_emu_sincos(ctypes.c_int(len(x)), vp(x), vp(s), vp(c))
    assert np.abs(s - np.sin(x)).max() < 3e-16 and np.abs(c - np.cos(x)).max() < 3e-16
    assert np.abs(s * s + c * c - 1).max() < 5e-16

def test_emu_long_horizon_above_32_stages_equals_oracle():
    """N = 40: the kinematics points take two chunks of lanes and the multiplier staging four passes (bmpc_wave.inl wave_eval,
    wave_node_grad_wide); the lane program against the scalar oracle, problem by problem."""
    from boundmpc_amd import workload
    from oracle import c_oracle
    N = 40
    P, X, _ = workload.make_batch(3, seed=5, N=N)
    r = c_oracle.solve(P, X, N, 4, 0.1, opts=c_oracle.default_opts(mu_init=3.0, slack_push=0.1, stall_window=20), nthreads=3)
    e = emu.solve(P, X, N, 4, 0.1, opts=emu.default_opts(mu_init=3.0, slack_push=0.1, stall_window=20), nthreads=3)
    assert (r["status"] == 0).all() and (e["status"] == 0).all()
    assert (r["iters"] == e["iters"]).all()
    d = (e["x"] - r["x"]).reshape(-1, N, 44)[:, :, 8:15]
    assert np.abs(d).max() < 1e-8

def test_flop_counting_build_of_the_kernel_text_runs_the_same_iterations():
    """tests/emu/bmpc_emu_flops.cpp compiles the kernel text with a counting number type (profiles/flops_current.json comes from it):
    it must keep building with the kernel and take the same number of iterations as the plain emulator."""
    from boundmpc_amd import workload
    P, X, _ = workload.make_batch(2, seed=0)
    e = emu.solve(P, X, 10, 4, 0.1)
    c = emu.count_flops(P, X, 10, 4, 0.1)
    assert c["iterations"] == int(e["iters"].sum()) and c["converged"] == 2
    assert 1.0e6 < c["flops_per_iteration"] < 2.5e6


# ---- team variant of the wave program (NW cooperating waves per problem; tests/emu/bmpc_emu_team.cpp) ----
@pytest.mark.parametrize("nw", [4, 2, "pair"])
def test_team_program_matches_oracle_and_is_order_invariant(nw):
    """The team text (wide passes over 64 NW lanes, helper wave of the Riccati sweep, workspace rows in LDS; "pair": two waves with the workspace
    in the global slab and the helper's own staging, csrc/bmpc_pair.hip) against the CPU oracle, and bit-for-bit the same under forward / reverse /
    scrambled orders of the waves inside a wide phase and of the lanes inside a wave."""
    from boundmpc_amd import workload
    P, X, _ = workload.make_batch(1024, seed=0, rows=(0, 24), workers=1)
    ref = c_oracle.solve(P, X, 10, 4, 0.1, nthreads=4)
    one = emu.solve(P, X, 10, 4, 0.1, nthreads=4)
    os.environ["BMPC_EMU_POISON"] = "1"      # LDS and workspace of every problem start as NaN: a read of a word nobody wrote shows
    try:
        t = emu.solve_team(P, X, 10, 4, 0.1, nw=nw, nthreads=4)
    finally:
        del os.environ["BMPC_EMU_POISON"]
    assert (t["status"] == ref["status"]).all() and (t["status"] == 0).all()
    assert np.abs(t["iters"] - ref["iters"]).max() <= 1
    d = (t["x"] - ref["x"]).reshape(-1, 10, 44)[:, :, 8:15]
    assert np.sqrt((d ** 2).mean()) < 1e-7
    # the team changes only the order of a few sums behind discrete decisions: on this batch its iterates equal the one-wave program's
    assert np.array_equal(t["iters"], one["iters"]) and np.abs(t["x"] - one["x"]).max() < 1e-9
    for lo, wo in ((1, 1), (2, 2), (0, 2)):
        c = emu.solve_team(P, X, 10, 4, 0.1, nw=nw, lane_order=lo, wave_order=wo, nthreads=4)
        assert np.array_equal(c["x"], t["x"]) and np.array_equal(c["iters"], t["iters"]) and np.array_equal(c["lam_g"], t["lam_g"])


@pytest.mark.parametrize("nw", [4, "pair"])
def test_team_program_tight_tubes_short_horizons_and_warm_start(nw):
    """Teams / pairs on what stresses the Riccati sweep's retry path (tight tubes: indefinite stage blocks, regularisation: the sweep is abandoned at a
    stage barrier and started again), on other short horizons (pairs: 1, 2 and 11 too -- the barrier pairing of helper and sweep depends on N), and on
    the warm entry (dual state), each against the one-wave program, whose lockstep with the oracle the tests above hold."""
    from boundmpc_amd import workload
    P, X, _ = workload.make_batch(64, seed=9, N=10, tight=True, rows=(0, 24), workers=1)
    a, b = emu.solve(P, X, 10, 4, 0.1, nthreads=4), emu.solve_team(P, X, 10, 4, 0.1, nw=nw, nthreads=4)
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"]) and np.abs(a["x"] - b["x"]).max() < 1e-9
    for N in (3, 7) + ((1, 2, 11) if nw == "pair" else ()):
        P2, X2, _ = workload.make_batch(32, seed=3, N=N, rows=(0, 12), workers=1)
        a, b = emu.solve(P2, X2, N, 4, 0.1, nthreads=4), emu.solve_team(P2, X2, N, 4, 0.1, nw=nw, nthreads=4)
        assert (b["status"] == 0).all() and np.array_equal(a["iters"], b["iters"]) and np.abs(a["x"] - b["x"]).max() < 1e-9
    P3, X3, _ = workload.make_batch(32, seed=4, rows=(0, 8), workers=1)
    st_a, st_b = np.zeros((8, 572)), np.zeros((8, 572))
    for rep in range(2):      # second solve: warm start from the stored multipliers
        a = emu.solve(P3, X3, 10, 4, 0.1, nthreads=4, state=st_a); b = emu.solve_team(P3, X3, 10, 4, 0.1, nw=nw, nthreads=4, state=st_b)
        assert np.array_equal(a["iters"], b["iters"]) and np.abs(a["x"] - b["x"]).max() < 1e-9 and np.abs(st_a - st_b).max() < 1e-7


def test_restoration_phase_kernel_text_follows_the_oracle_on_g13b():
    """Fixture g13b (38 first failing closed-loop ticks: 28 locally infeasible, 10 feasible).  The kernel text with the restoration phase -- (a) as
    the batch kernels run it: main phase in a kernel without the phase, internal status 4, continuation by the restoration kernel from the iterate
    (Problem::resto_from); (b) with the phase inside the kernel, as the fused closed-loop ticks run it; (c) the team text -- against the oracle:
    every status equal, iterations within 8 (on the solves that do not go through three restoration phases), and (a) == (b) bit for bit (entering the restoration phase discards everything but the iterate)."""
    # (start_rollout = 0: in the loop these ticks are warm solves and are never rolled out; solved cold here)
    d = np.load(os.path.join(G, "g13b_first_failures_256_streams.npz"))
    ref = c_oracle.solve(d["p"], d["x0"], 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=500, start_rollout=0), nthreads=4)
    a = emu.solve(d["p"], d["x0"], 10, 4, 0.1, opts=emu.default_opts(max_iter=500, start_rollout=0), nthreads=4)
    os.environ["BMPC_EMU_INKERNEL"] = "1"
    try:
        b = emu.solve(d["p"], d["x0"], 10, 4, 0.1, opts=emu.default_opts(max_iter=500, start_rollout=0), nthreads=4)
    finally:
        del os.environ["BMPC_EMU_INKERNEL"]
    c = emu.solve_team(d["p"], d["x0"], 10, 4, 0.1, opts=emu.default_opts(max_iter=500, start_rollout=0), nthreads=4)
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["status"], b["status"])
    for r in (a, c):
        di = np.abs(r["iters"] - ref["iters"])
        # (iterations: equal on 36 of the 38; the two that end as status 2 after their THIRD restoration phase -- a hundred iterations of an ill-conditioned
        #  solve -- may leave at different counts)
        assert np.array_equal(r["status"], ref["status"]) and di[ref["status"] == 0].max() <= 8 and (di <= 8).sum() >= 36
        ok = ref["status"] == 0
        assert ok.sum() == 8 and np.abs(r["f"][ok] - ref["f"][ok]).max() < 1e-7 * np.abs(ref["f"][ok]).max()
    assert (ref["status"] != 4).all() and (a["status"] != 4).all()      # the internal hand-over status never leaves the library


def test_bad_warm_starts_are_rescued_by_the_restoration_phase():
    """128 tight N = 10 problems of the synthetic generator, each FEASIBLE (its cold start converges in ~13 iterations), started from a warm start with
    Gaussian noise of 0.3 on every variable -- a trajectory far off its own dynamics.  Round 4's solver (restoration off) converges on a quarter of
    them and stalls on the rest; the restoration phase first rolls the states out from the measured state with the iterate's own jerks (equality
    residuals above 1e-2), repairs the inequality rows and hands a feasible point back: >= 97 % converge within the default budget, in ~42 iterations
    on average.  Oracle and kernel text (batch kernel -> restoration kernel hand-over) agree problem by problem."""
    from boundmpc_amd import workload
    P, X, _ = workload.make_batch(128, seed=50, N=10, tight=True, workers=4)
    X2 = X + np.random.default_rng(5).normal(size=X.shape) * 0.3
    # (start_rollout = 0: these starts would otherwise be rolled out before the first iteration and never need the phase -- see the last lines)
    off = c_oracle.solve(P, X2, 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=300, restoration=0, start_rollout=0), nthreads=4)
    o = c_oracle.solve(P, X2, 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=300, start_rollout=0), nthreads=4)
    e = emu.solve(P, X2, 10, 4, 0.1, opts=emu.default_opts(max_iter=300, start_rollout=0), nthreads=4)
    assert (off["status"] == 0).mean() <= 0.4
    assert (o["status"] == 0).mean() >= 0.97 and o["iters"][o["status"] == 0].mean() <= 50
    assert np.array_equal(o["status"], e["status"]) and np.abs(o["iters"] - e["iters"]).max() <= 8
    ok = o["status"] == 0
    cold = c_oracle.solve(P, X, 10, 4, 0.1, nthreads=4)
    same = np.abs(o["f"][ok] - cold["f"][ok]) < 1e-6 * np.abs(cold["f"][ok])
    assert same.mean() >= 0.9      # mostly the minimiser the reference's cold start reaches (the NLP is non-convex: a few end in a neighbouring one)
    dq = (o["x"][ok] - e["x"][ok]).reshape(-1, 10, 44)[:, :, 8:15]
    assert np.sqrt((dq ** 2).mean(axis=(1, 2))).max() < 1e-5
    # the defaults: a cold start that is not a trajectory is rolled out before the first iteration (START_ROLLOUT_TOL): every problem converges, in
    # a third of the iterations, whether the phase is on or not; oracle and kernel text take the same decision and the same iterations
    for kw in ({}, {"restoration": 0}):
        o1 = c_oracle.solve(P, X2, 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=300, **kw), nthreads=4)
        e1 = emu.solve(P, X2, 10, 4, 0.1, opts=emu.default_opts(max_iter=300, **kw), nthreads=4)
        assert (o1["status"] == 0).all() and o1["iters"].mean() <= 20 and np.array_equal(o1["status"], e1["status"]) and np.abs(o1["iters"] - e1["iters"]).max() <= 3, kw
        dq = (o1["x"] - e1["x"]).reshape(-1, 10, 44)[:, :, 8:15]
        assert np.sqrt((dq ** 2).mean(axis=(1, 2))).max() < 1e-5


def test_numerical_breakdowns_of_far_off_starts_go_to_the_restoration_phase():
    """A dual residual beyond 1e12 used to end a solve as status 3.  With the restoration phase available (mode 1, and mode 2 = the default of long
    horizons: after a numerical breakdown ONLY, so that configs[3] pays nothing) the solve goes there instead: 32 loose N = 20 problems started with
    noise 0.3 on every variable: all converge (mode 0: a quarter breaks down); the other (N, S) of fixture G12 with the same noise on their recorded
    warm starts: 12 of 12, 15 of 16, 10 of 10 (mode 0: none -- they stall).  Kernel text (hand-over, incl. the restart count of a long horizon) == oracle."""
    from boundmpc_amd import workload
    P, X, _ = workload.make_batch(32, seed=7, N=20, workers=4)
    X2 = X + np.random.default_rng(3).normal(size=X.shape) * 0.3
    long_h = dict(mu_init=3.0, slack_push=0.1, stall_window=20, start_rollout=0)      # (start_rollout = 0: x0 as given, so that the main phase does break down)
    o, e = c_oracle.solve(P, X2, 20, 4, 0.1, opts=c_oracle.default_opts(restoration=2, **long_h), nthreads=4), emu.solve(P, X2, 20, 4, 0.1, opts=emu.default_opts(restoration=2, **long_h), nthreads=4)
    off = c_oracle.solve(P, X2, 20, 4, 0.1, opts=c_oracle.default_opts(restoration=0, **long_h), nthreads=4)
    assert (o["status"] == 0).all() and np.array_equal(o["status"], e["status"]) and (off["status"] == 3).sum() >= 4
    o1, e1 = c_oracle.solve(P, X2, 20, 4, 0.1, nthreads=4), emu.solve(P, X2, 20, 4, 0.1, nthreads=4)      # the defaults: rolled out first, 23 instead of 75 iterations
    assert (o1["status"] == 0).all() and np.array_equal(o1["status"], e1["status"]) and o1["iters"].mean() <= 0.5 * o["iters"].mean() and np.abs(o1["iters"] - e1["iters"]).max() <= 6
    d = np.load(os.path.join(G, "g12_pack_other_sizes.npz"))
    for key, N, S, want in (("n5s2", 5, 2, 12), ("n8s3", 8, 3, 15), ("n6s5", 6, 5, 10)):
        Pk = np.where(np.isfinite(d[key + "_p"]), d[key + "_p"], 0.0); Xk = d[key + "_x0"]; dt = float(d[key + "_dt"])
        Xn = Xk + np.random.default_rng(3).normal(size=Xk.shape) * 0.3
        o = c_oracle.solve(Pk, Xn, N, S, dt, opts=c_oracle.default_opts(max_iter=300, start_rollout=0), nthreads=4)
        e = emu.solve(Pk, Xn, N, S, dt, opts=emu.default_opts(max_iter=300, start_rollout=0), nthreads=4)
        off = c_oracle.solve(Pk, Xn, N, S, dt, opts=c_oracle.default_opts(max_iter=300, restoration=0, start_rollout=0), nthreads=4)
        assert (o["status"] == 0).sum() >= want and np.array_equal(o["status"], e["status"]) and np.abs(o["iters"] - e["iters"]).max() <= 8, key
        assert (off["status"] == 0).sum() == 0, key


def test_fixed_barrier_level_iterates_follow_the_oracle():
    """The real-time iteration on one barrier level (BatchedOCPSolver(fixed_barrier=...), bench_stream.py rtfix-*): mu_init = mu_warm = tol * mu_min_fac,
    so the barrier never moves and `tol` never fires -- K iterations are K Newton steps on one barrier problem.  Kernel text and oracle produce the same
    iterate after K = 3 and 6 steps (status 1 = iteration cap), cold and from a carried dual state; by 6 steps the barrier problem is solved to 1e-3."""
    from boundmpc_amd import workload
    P, X, _ = workload.make_batch(16, seed=12, N=10, workers=1)
    kw = dict(tol=1e-3, mu_init=0.1, mu_warm=0.1, mu_min_fac=100.0)
    so, se = np.zeros((16, c_oracle.state_len(10))), np.zeros((16, c_oracle.state_len(10)))
    x_o = x_e = X
    for K in (6, 3, 3):      # a cold solve, then two warm continuations from the iterate and the dual state it left
        o = c_oracle.solve(P, x_o, 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=K, **kw), nthreads=4, state=so)
        e = emu.solve(P, x_e, 10, 4, 0.1, opts=emu.default_opts(max_iter=K, **kw), nthreads=4, state=se)
        assert (o["status"] == 1).all() and (e["status"] == 1).all() and (o["iters"] == K).all() and (e["iters"] == K).all()
        assert np.abs(o["x"] - e["x"]).reshape(-1, 10, 44)[:, :, 8:15].max() < 1e-7 and np.abs(so - se).max() < 1e-6 * (1 + np.abs(so).max())
        assert np.allclose(so[:, 570], 0.1) and np.allclose(se[:, 570], 0.1)      # the stored barrier level stays where it is
        x_o, x_e = o["x"], e["x"]
    assert np.abs(o["g"].reshape(-1, 10, 43)[:, :, :36]).max() < 1e-3      # twelve Newton steps on one level: the dynamics rows are met



def test_second_attempt_of_a_long_horizon_solve():
    """Round 6: a stateless long-horizon solve that ends with status 2 is run once more from x0 on the barrier start of the short horizons (kernel text:
    wave_solve_retry; oracle: bmpc_oracle_solve_warm).  Problem 695 of BASELINE configs[3] is one of the 22 feasible problems the first attempt gives up on
    (profiles/r06_h_configs3_failures.txt): status 2 after 120 iterations without the rule (with it: two restarts, 90 iterations, then the second attempt), converged with it, the iterations of both attempts added;
    problem 1658 stays at status 2 (the second attempt runs into its cap of 100: the verdict of the first is kept, not status 1)."""
    from boundmpc_amd import workload
    P = np.concatenate([workload.make_batch(8192, seed=2, N=30, tight=True, rows=(b, b + 1))[0] for b in (695, 1658)])
    X = np.concatenate([workload.make_batch(8192, seed=2, N=30, tight=True, rows=(b, b + 1))[1] for b in (695, 1658)])
    long_h = dict(mu_init=3.0, slack_push=0.1, stall_window=20, restoration=2)
    r0 = c_oracle.solve(P, X, 30, 4, 0.1, opts=c_oracle.default_opts(**long_h), nthreads=2)
    assert list(r0["status"]) == [2, 2] and list(r0["iters"]) == [120, 160]             # (three barrier restarts, then the verdict)
    r1 = c_oracle.solve(P, X, 30, 4, 0.1, nthreads=2)                                     # the defaults of a long horizon: second attempt of 100 iterations
    e1 = emu.solve(P, X, 30, 4, 0.1, nthreads=2)
    assert list(r1["status"]) == [0, 2] and list(e1["status"]) == [0, 2]
    assert 90 < r1["iters"][0] <= 190 and r1["iters"][1] == 220 and e1["iters"][1] == 220 and abs(int(e1["iters"][0]) - int(r1["iters"][0])) <= 10
    assert np.abs(e1["x"][0] - r1["x"][0]).max() < 1e-4
    f, g = c_oracle.eval_fg(P[0], r1["x"][0], 30, 4, 0.1)                                   # the recovered point is feasible (reference-form rows)
    g = np.asarray(g).reshape(30, 43)
    assert np.abs(g[:, :36]).max() < 1e-8 and g[:, 36:].max() < 1e-8
