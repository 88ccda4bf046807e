"""The numpy oracle (oracle/nlp.py) against golden vectors produced by the reference's own
numeric code (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import nlp

G = os.path.join(os.path.dirname(__file__), "golden")


def test_g1_kinematics():
    d = np.load(os.path.join(G, "g1_kinematics.npz"))
    for i in range(d["q"].shape[0]):
        q, dq = d["q"][i], d["dq"][i]
        np.testing.assert_allclose(nlp.fk_pos(q), d["fk_pos"][i], atol=2e-15, rtol=0)
        J = nlp.jacobian(q)
        np.testing.assert_allclose(J, d["jacobian_fk"][i], atol=5e-15, rtol=0)
        np.testing.assert_allclose(nlp.velocity_ee(q, dq), d["velocity_ee"][i], atol=2e-14, rtol=0)
        np.testing.assert_allclose(nlp.omega_ee(q, dq), d["omega_ee"][i], atol=2e-14, rtol=0)
        A, O, p, R = nlp.chain(q)
        np.testing.assert_allclose(R, d["hom"][i][:3, :3], atol=5e-15, rtol=0)
        np.testing.assert_allclose(p, d["hom"][i][:3, 3], atol=2e-15, rtol=0)
    lim = d["limits"]
    np.testing.assert_array_equal(lim[1], nlp.Q_LIM)
    np.testing.assert_array_equal(lim[0], -nlp.Q_LIM)
    np.testing.assert_array_equal(lim[3], nlp.DQ_LIM)
    assert d["u_lim"][1] == nlp.U_LIM and d["u_lim"][0] == -nlp.U_LIM


def test_g2_integrator():
    d = np.load(os.path.join(G, "g2_integrator.npz"))
    h = float(d["h"])
    for i in range(d["jm2"].shape[0]):
        x, dx, ddx = nlp.integrate_chain(d["q0"][i], d["dq0"][i], d["ddq0"][i], d["jm2"][i][:, 0], d["jm2"][i][:, 1], h)
        np.testing.assert_allclose(x, d["ang2"][i], atol=1e-15, rtol=1e-15)
        np.testing.assert_allclose(dx, d["vel2"][i], atol=1e-15, rtol=1e-15)
        np.testing.assert_allclose(ddx, d["acc2"][i], atol=1e-15, rtol=1e-15)
    for i in range(d["jm11"].shape[0]):
        for k in range(10):
            x, dx, ddx = nlp.integrate_jerk_matrix(d["jm11"][i], k, d["q0"][i], d["dq0"][i], d["ddq0"][i], h)
            np.testing.assert_allclose(x, d["ang11"][i, k], atol=1e-13, rtol=1e-13)
            np.testing.assert_allclose(dx, d["vel11"][i, k], atol=1e-13, rtol=1e-13)
            np.testing.assert_allclose(ddx, d["acc11"][i, k], atol=1e-13, rtol=1e-13)
    for i in range(d["jm3"].shape[0]):
        x, dx, ddx = nlp.integrate_chain(d["q0"][i], d["dq0"][i], d["ddq0"][i], d["jm3"][i][:, 0], d["jm3"][i][:, 1], h)
        np.testing.assert_allclose(x, d["ang3"][i], atol=1e-15, rtol=1e-15)
        np.testing.assert_allclose(dx, d["vel3"][i], atol=1e-15, rtol=1e-15)
        np.testing.assert_allclose(ddx, d["acc3"][i], atol=1e-15, rtol=1e-15)


def test_p_layout_sizes():
    for S in (2, 3, 4, 6):
        assert nlp.p_layout(S)["_size"] == nlp.n_p(S)
    lay = nlp.p_layout(4)
    # offsets listed in SURVEY.md 8(a1)
    for name, off in (("q0", 0), ("p0", 24), ("v0", 30), ("iw_ref0", 36), ("dtau_init", 39), ("dtau_init_par", 42),
                      ("x_phi_d", 78), ("jerk_cur", 81), ("jerk_phi_cur", 88), ("phi_switch", 89),
                      ("jac_dtau_r_T", 94), ("jac_dtau_l_T", 103), ("p_ref", 112), ("dp_ref", 136),
                      ("dp_normed_ref", 160), ("bp1", 172), ("br2", 208), ("a4", 220), ("a0", 400),
                      ("weights", 445), ("phi_max", 460), ("dphi_max", 461), ("v1", 462), ("qd", 498)):
        assert lay[name][0] == off, name


@pytest.mark.parametrize("which", [1, 2])
def test_g6_cold_start_and_feasibility(which):
    """x0 captured from the reference step() equals the restated cold start, and the NLP
    restatement is consistent at it: all 36 dynamics equalities vanish at the cold start
    (q = q0, p = p0, everything else zero -- BoundMPC.py:316-321) and inequalities hold."""
    d = np.load(os.path.join(G, f"g6_pack_exp{which}_tick0.npz"))
    p, x0 = d["p_f64"], d["x0_f64"]
    np.testing.assert_array_equal(nlp.cold_start(d["q0"], d["p0fk"], 10), x0)
    f, g = nlp.nlp_eval(x0, p, 10, 4, float(d["dt_f64"]))
    g = g.reshape(10, 43)
    assert np.abs(g[:, :36]).max() < 1e-14
    assert (g[:, 36:] <= 1e-12).all()
    assert np.isfinite(f)


# ---------------------------------------------------------------------------------------------
# C oracle (oracle/bmpc_oracle.c) against the numpy restatement and by complex-step derivatives
# ---------------------------------------------------------------------------------------------
from oracle import c_oracle  # noqa: E402


def _perturbed_point(which, N=10, seed=1, scale=0.03):
    d = np.load(os.path.join(G, f"g6_pack_exp{which}_tick0.npz"))
    p, x0 = d["p_f64"], d["x0_f64"]
    rng = np.random.default_rng(seed)
    x = x0 + rng.normal(size=x0.shape) * scale
    x.reshape(N, 44)[:, 41] = np.abs(x.reshape(N, 44)[:, 41]) + np.linspace(0.1, 2.5, N)   # spread phi over segments
    return p, x, rng


@pytest.mark.parametrize("which", [1, 2])
def test_c_oracle_values_match_numpy(which):
    p, x, _ = _perturbed_point(which)
    f, g = nlp.nlp_eval(x, p, 10, 4, 0.1)
    fc, gc = c_oracle.eval_fg(p, x, 10, 4, 0.1)
    assert abs(f - fc) <= 1e-12 * abs(f)
    np.testing.assert_allclose(gc, g, atol=1e-13)


def test_c_oracle_kinematic_derivatives():
    rng = np.random.default_rng(5)
    for _ in range(4):
        q, dq = rng.uniform(-1.5, 1.5, 7), rng.uniform(-1, 1, 7)
        mp, mv, mw = rng.normal(size=3), rng.normal(size=3), rng.normal(size=3)
        pos, v, J, D, W = c_oracle.kin(q, dq, mp, mv, mw)
        Jn = nlp.jacobian(q)
        np.testing.assert_allclose(J, Jn, atol=1e-14)
        Dn = np.zeros((6, 7))
        for i in range(7):
            qc = q.astype(complex); qc[i] += 1e-30j
            Dn[:, i] = (nlp.jacobian(qc) @ dq).imag / 1e-30
        np.testing.assert_allclose(D, Dn, atol=1e-13)

        def S(y):
            Jy = nlp.jacobian(y[:7])
            return mp @ nlp.fk_pos(y[:7]) + mv @ (Jy[:3] @ y[7:]) + mw @ (Jy[3:] @ y[7:])

        def grad(y):
            g = np.zeros(14)
            for i in range(14):
                yc = y.astype(complex); yc[i] += 1e-30j
                g[i] = S(yc).imag / 1e-30
            return g
        y = np.concatenate([q, dq])
        H = np.zeros((14, 14))
        for i in range(14):
            e = np.zeros(14); e[i] = 1e-5
            H[:, i] = (grad(y + e) - grad(y - e)) / 2e-5
        np.testing.assert_allclose(W, H, atol=2e-9)


def test_c_oracle_adjoint_is_lagrangian_gradient():
    """Multipliers from the adjoint sweep zero the Lagrangian gradient w.r.t. every state variable; the jerk part
    equals Rj -- checked against a complex-step gradient of the numpy Lagrangian (first derivatives of f, g and of
    the internal inequality rows are therefore all correct)."""
    N, S, h = 10, 4, 0.1
    p, x, rng = _perturbed_point(1)
    nu = rng.uniform(0.1, 2, N * 57)
    lam, rj, _ = c_oracle.adjoint(p, x, nu, N, S, h)

    def L(xc):
        f, g = nlp.nlp_eval(xc, p, N, S, h)
        return f + lam @ g.reshape(N, 43)[:, :36].reshape(-1) + nu @ nlp.internal_ineq(xc, p, N, S)
    gl = np.zeros(x.size)
    xc = x.astype(complex)
    for i in range(x.size):
        xc[i] += 1e-30j; gl[i] = L(xc).imag / 1e-30; xc[i] = x[i]
    gl = gl.reshape(N, 44)
    scale = max(1.0, np.abs(lam).max())
    assert np.abs(gl[:, 8:]).max() < 1e-11 * scale
    np.testing.assert_allclose(gl[:, :8], rj.reshape(N, 8), atol=1e-11 * scale)


@pytest.mark.parametrize("which", [1, 2])
def test_c_oracle_solution_is_kkt_point(which):
    """Solve tick 0 from the reference's cold start and certify the result with quantities computed by the
    independent numpy restatement: equality residuals, inequality feasibility, complementarity and the
    Lagrangian gradient (complex step) with the reported multipliers lam_g / lam_x in CasADi's sign convention."""
    N, S, h = 10, 4, 0.1
    d = np.load(os.path.join(G, f"g6_pack_exp{which}_tick0.npz"))
    p, x0 = d["p_f64"], d["x0_f64"]
    out = c_oracle.solve(p, x0, N, S, h, c_oracle.default_opts(tol=1e-8))
    assert out["status"][0] == 0 and out["iters"][0] < 60
    x, lam_g, lam_x = out["x"][0], out["lam_g"][0], out["lam_x"][0]
    f, g = nlp.nlp_eval(x, p, N, S, h)
    np.testing.assert_allclose(out["g"][0], g, atol=1e-12)
    assert abs(out["f"][0] - f) < 1e-9 * abs(f)
    g2 = g.reshape(N, 43)
    assert np.abs(g2[:, :36]).max() < 1e-8
    assert g2[:, 36:].max() < 1e-8
    lbx, ubx, _, _ = nlp.bounds(N)
    assert (x >= lbx - 1e-9).all() and (x <= ubx + 1e-9).all()
    lg = lam_g.reshape(N, 43)
    assert (lg[:, 36:] >= 0).all()
    assert np.abs(lg[:, 36:] * g2[:, 36:]).max() < 1e-6          # complementarity of the inequality rows
    gf, Jg = nlp.jac_g_complex_step(x, p, N, S, h)
    r = gf + Jg.T @ lam_g + lam_x
    assert np.abs(r).max() < 2e-6 * max(1.0, np.abs(lam_g).max() / 100)


def test_c_oracle_reproduces_closed_loop_fixture():
    """The committed closed-loop fixture (reference host code + oracle solver) is reproducible."""
    d = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))
    idx = np.arange(0, len(d["x"]), 7)
    out = c_oracle.solve(d["p"][idx], d["x0"][idx], 10, 4, 0.1)
    assert (out["status"] == 0).all()
    assert np.abs(out["iters"] - d["iters"][idx]).max() <= 1
    # the fixture was produced by another build of the same source: a tick that converges at the threshold may take one
    # Newton step more or less, which moves the weakly determined jerks by ~1e-6 and the joints by ~1e-8
    dd = (out["x"] - d["x"][idx]).reshape(-1, 10, 44)
    assert np.sqrt(np.mean(dd[:, :, 8:15] ** 2)) < 1e-7 and np.abs(dd[:, :, :8]).max() < 1e-4


@pytest.mark.parametrize("which", [1, 2])
def test_scipy_independent_solution(which):
    """An independent NLP method (scipy SLSQP on the numpy restatement, reference-form constraints, cold start of the
    reference) lands on the same minimiser.  Fixture g8 is produced by oracle/solve_scipy.py (hours of CPU)."""
    d = np.load(os.path.join(G, f"g8_scipy_exp{which}_tick0.npz"))
    out = c_oracle.solve(d["p"], d["x0"], 10, 4, 0.1, c_oracle.default_opts(tol=1e-8))
    z, zs = out["x"][0].reshape(10, 44), d["x"].reshape(10, 44)
    rms_q = np.sqrt(np.mean((z[:, 8:15] - zs[:, 8:15]) ** 2))
    assert rms_q < 1e-6, rms_q                # measured 2.5e-8 / 5.4e-9 rad; the north-star tolerance is 1e-4 rad RMS
    assert abs(out["f"][0] - float(d["f"])) < 1e-9 * abs(float(d["f"]))


# ---------------------------------------------------------------------------------------------
# G9: the NLP itself -- f(x,p), g(x,p), bounds, parameter order and exact derivatives produced by the reference's own,
# unmodified setup_optimization_problem (casadi_ocp_formulation.py:9-391) run with numbers in place of symbols
# (tests/golden/make_g9.py, ref_nlp.py, numeric_sx.py).  Pins the composites reference_function / error_function /
# objective_function / decomp_function / integration_function and the constraint assembly, not only the leaves.
# ---------------------------------------------------------------------------------------------
G9 = np.load(os.path.join(G, "g9_nlp.npz"))
G9_SETS = (("n10", 10, 4, 0.1), ("n30", 30, 4, 0.1), ("n3s2", 3, 2, 0.05), ("n5s3", 5, 3, 0.05), ("n4s5", 4, 5, 0.05))


def _close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    assert np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))), float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


@pytest.mark.parametrize("key,N,S,h", G9_SETS)
def test_g9_numpy_restatement_equals_reference_nlp(key, N, S, h):
    X, P, F, Gg = G9[key + "_x"], G9[key + "_p"], G9[key + "_f"], G9[key + "_g"]
    assert len(X) >= 4
    for i in range(len(X)):
        f, g = nlp.nlp_eval(X[i], P[i], N, S, h)
        _close(f, F[i], 1e-13)
        _close(g, Gg[i], 1e-12)
    lbx, ubx, lbg, ubg = nlp.bounds(N)
    for mine, name in ((lbx, "lbx"), (ubx, "ubx"), (lbg, "lbg"), (ubg, "ubg")):
        np.testing.assert_array_equal(mine, G9[f"{key}_{name}"])


def test_g9_cases_cover_every_segment_and_the_past_the_end_row():
    X, P = G9["n10_x"], G9["n10_p"]
    seen = set()
    for x, p in zip(X, P):
        sw = p[89:94]
        for ph in x.reshape(10, 44)[:, 41]:
            seen.add(int(np.searchsorted(sw[1:], ph, side="right")))
    assert seen == {0, 1, 2, 3, 4}                 # 4 = phi >= phi_switch[S]: row S of a4..a0 is read
    tags = [str(t) for t in G9["n10_tag"]]
    assert sum(t.startswith("exp1") for t in tags) >= 30 and sum(t.startswith("exp2") for t in tags) >= 30
    # sigmoid of the objective (casadi_ocp_formulation.py:237) in its transition zone on some stage of some case
    mid = 0
    for x, p in zip(X, P):
        s = 1 / (1 + np.exp(-np.clip(100 * (x.reshape(10, 44)[:, 41] - (p[460] - 0.02)), -700, 700)))
        mid += int(np.any((s > 0.05) & (s < 0.95)))
    assert mid >= 5


@pytest.mark.parametrize("key,S", [("n10", 4), ("n3s2", 2), ("n5s3", 3), ("n4s5", 5)])
def test_g9_parameter_order_is_the_references(key, S):
    """The symbol behind every entry of p, as learned from the reference's own `params = ca.vertcat(...)`
    (casadi_ocp_formulation.py:361-376), against the offsets of nlp.p_layout."""
    names = [str(n) for n in G9[key + "_p_names"]]
    assert len(names) == nlp.n_p(S)
    lay = nlp.p_layout(S)
    # reference symbol name -> (layout name, CasADi shape)
    table = [("q_0", "q0", (7, 1)), ("dq_0", "dq0", (7, 1)), ("ddq_0", "ddq0", (7, 1)), ("phi_0", "phi0", (1, 1)),
             ("dphi_0", "dphi0", (1, 1)), ("ddphi_0", "ddphi0", (1, 1)), ("p_0", "p0", (6, 1)), ("v_0", "v0", (6, 1)),
             ("i_omega_ref_0", "iw_ref0", (3, 1)), ("initial lie space error", "dtau_init", (3, 1)),
             ("initial lie space error par", "dtau_init_par", (3, S)), ("initial lie space error orth1", "dtau_init_orth1", (3, S)),
             ("initial lie space error orth2", "dtau_init_orth2", (3, S)), ("x phi_desired", "x_phi_d", (3, 1)),
             ("jerk_current", "jerk_cur", (1, 7)), ("jerk_phi_current", "jerk_phi_cur", (1, 1)),
             ("path parameter switch", "phi_switch", (S + 1, 1)), ("right jacobian at initial error", "jac_dtau_r_T", (3, 3)),
             ("left jacobian at initial error", "jac_dtau_l_T", (3, 3)), ("linear ref position", "p_ref", (S, 6)),
             ("linear ref velocity", "dp_ref", (S, 6)), ("norm of orientation reference", "dp_normed_ref", (S, 3)),
             ("orthogonal error basis 1", "bp1", (S, 3)), ("orthogonal error basis 2", "bp2", (S, 3)),
             ("orthogonal error basis 1r", "br1", (S, 3)), ("orthogonal error basis 2r", "br2", (S, 3)),
             ("parameter 4 error function", "a4", (S + 1, 9)), ("parameter 3 error function", "a3", (S + 1, 9)),
             ("parameter 2 error function", "a2", (S + 1, 9)), ("parameter 1 error function", "a1", (S + 1, 9)),
             ("parameter 0 error function", "a0", (S + 1, 9)), ("cost weights", "weights", (15, 1)),
             ("max path parameter", "phi_max", (1, 1)), ("max path parameter", "dphi_max", (1, 1)),
             ("v1", "v1", (S, 3)), ("v2", "v2", (S, 3)), ("v3", "v3", (S, 3)), ("q desired", "qd", (7, 1))]
    k = 0
    for ref_name, mine, (n, m) in table:
        assert lay[mine][0] == k, (mine, lay[mine][0], k)
        # column-major flattening of the (n, m) symbol (the jerk row is transposed first, a 1 x 7 either way)
        for c in range(m):
            for r in range(n):
                assert names[k] == "%s[%d,%d]" % (ref_name, r, c), (k, names[k], ref_name, r, c)
                k += 1
        assert int(np.prod(lay[mine][1])) == n * m
    assert k == nlp.n_p(S)
    if key == "n10":
        xn = [str(n) for n in G9["n10_x_names"]]
        stage = (["u_%d[%d,0]" % (0, i) for i in range(8)] + ["q_1[%d,0]" % i for i in range(7)] + ["dq_1[%d,0]" % i for i in range(7)]
                 + ["ddq_1[%d,0]" % i for i in range(7)] + ["i_omega_1[%d,0]" % i for i in range(6)] + ["v_1[%d,0]" % i for i in range(6)]
                 + ["phi_1[0,0]", "dphi_1[0,0]", "ddphi_1[0,0]"])
        assert xn[:44] == stage and xn[44] == "u_1[0,0]" and xn[44 + 8] == "q_2[0,0]" and len(xn) == 440


def test_g9_exact_derivatives_of_the_reference():
    """Gradient of f and Jacobian of g by a complex step THROUGH THE REFERENCE'S CODE, against the same of the numpy
    restatement (which in turn certifies the C oracle's analytic derivatives in the tests above)."""
    for j, i in enumerate(G9["n10_deriv_case"]):
        x, p = G9["n10_x"][i], G9["n10_p"][i]
        gf, Jg = nlp.jac_g_complex_step(x, p, 10, 4, 0.1)
        _close(gf, G9["n10_grad_f"][j], 1e-11)
        _close(Jg, G9["n10_jac_g"][j], 1e-11)
        # structure the solver relies on: stage k's constraints touch stages k-1 and k only
        J = G9["n10_jac_g"][j].reshape(10, 43, 10, 44)
        for k in range(10):
            for l in range(10):
                if l > k or l < k - 1:
                    assert not J[k, :, l, :].any()


@pytest.mark.parametrize("key,N,S,h", [s for s in G9_SETS if s[2] <= 4])
def test_g9_c_oracle_equals_reference_nlp(key, N, S, h):
    X, P, F, Gg = G9[key + "_x"], G9[key + "_p"], G9[key + "_f"], G9[key + "_g"]
    for i in range(len(X)):
        f, g = c_oracle.eval_fg(P[i], X[i], N, S, h)
        _close(f, F[i], 1e-12)
        _close(g, Gg[i], 1e-11)


def _g9_solution_cases():
    """(fixture index, derivative slot, experiment, tick) of the recorded solutions that carry reference derivatives."""
    tags = [str(t) for t in G9["n10_tag"]]
    out = []
    for j, i in enumerate(G9["n10_deriv_case"]):
        t = tags[int(i)]
        if t.endswith("_sol"):
            which, tick = int(t[3]), int(t.split("_")[1][4:])
            out.append((int(i), j, which, tick))
    return out


def _kkt_residual_with_reference_derivatives(i, j, x, lam_g, lam_x):
    assert np.abs(x - G9["n10_x"][i]).max() < 2e-5      # same point as the fixture (its derivatives apply; curvature <= ~1e3)
    return np.abs(G9["n10_grad_f"][j] + G9["n10_jac_g"][j].T @ lam_g + lam_x).max()


def test_g9_kkt_certificate_with_the_references_own_derivatives():
    """Solutions of recorded closed-loop ticks are KKT points of the REFERENCE's NLP: stationarity evaluated with the gradient and
    Jacobian that came out of the reference's code (complex step through setup_optimization_problem, fixture G9) and the multipliers
    of the solver in CasADi's sign convention; feasibility with the reference's g.  Ticks: cold start, around a segment switch, end of
    path with phi_max active, experiment 2 with its +-0.01 tube."""
    cases = _g9_solution_cases()
    assert len(cases) >= 8
    for i, j, which, tick in cases:
        d = np.load(os.path.join(G, f"g7_closedloop_exp{which}.npz"))
        out = c_oracle.solve(d["p"][tick], d["x0"][tick], 10, 4, 0.1, c_oracle.default_opts(tol=1e-8))
        assert out["status"][0] == 0
        r = _kkt_residual_with_reference_derivatives(i, j, out["x"][0], out["lam_g"][0], out["lam_x"][0])
        assert r < 2e-4, (which, tick, r)       # |x - x_fixture| <= 1e-6 times the curvature, plus the solver's own 1e-8-scaled residual
        g = G9["n10_g"][i].reshape(10, 43)
        assert np.abs(g[:, :36]).max() < 1e-7 and g[:, 36:].max() < 1e-7
        assert (out["lam_g"][0].reshape(10, 43)[:, 36:] >= 0).all()


def test_scipy_independent_solutions_of_closed_loop_ticks():
    """Fixture g8_scipy_ticks (oracle/solve_scipy.py --ticks): scipy SLSQP on the numpy restatement, reference-form constraints, from
    the x0 the reference's step() handed to the solver, for 15 warm-started closed-loop ticks -- segment switch inside the horizon
    (exp1 45/46, exp2 13/29/40), the integrated-omega unwrap ticks (exp1 49, exp2 47), experiment 2's +-0.01 tube active (10, 12),
    phi_max active (exp1 143).  The solver lands on the same minimisers (SLSQP stops at its own ~1e-6 accuracy on some)."""
    d = np.load(os.path.join(G, "g8_scipy_ticks.npz"))
    assert len(d["tick"]) == 15
    out = c_oracle.solve(d["p"], d["x0"], 10, 4, 0.1, c_oracle.default_opts(tol=1e-8))
    assert (out["status"] == 0).all()
    dq = (out["x"] - d["x"]).reshape(-1, 10, 44)[:, :, 8:15]
    rms = np.sqrt(np.mean(dq ** 2, axis=(1, 2)))
    assert rms.max() < 5e-6 and np.median(rms) < 2e-7, rms       # north-star tolerance: 1e-4 rad RMS
    assert np.abs(out["f"] - d["f"]).max() < 1e-7


def _g8_batch_groups():
    d = np.load(os.path.join(G, "g8_scipy_batch.npz"))
    return d, sorted({k.split("_")[0] for k in d.files})


def check_against_slsqp_batch(solve, label):
    """Shared by the CPU (oracle) and GPU (kernel) tests: fixture g8_scipy_batch (oracle/solve_scipy_batch.py) holds INDEPENDENT SLSQP solutions of
    samples of the benchmark batches -- configs[1] seed 0 from the reference's cold start; the tight N = 20 / N = 30 batches from a start in the
    basin of the oracle's solution (dense SLSQP needs hours per cold start there).  Per problem: where SLSQP and the solver end at the same
    objective (to 1e-6) the joint trajectories must agree to 5e-6 rad RMS; the problems where they do NOT are reported, not hidden -- this
    non-convex NLP has several local minimisers for some random q0 -- and must stay a small minority; a problem on which SLSQP found a
    LOWER objective is counted separately."""
    d, groups = _g8_batch_groups()
    assert "c1" in groups
    report = {}
    for gname in groups:
        N = int(d[f"{gname}_def"][0])
        P, X0 = d[f"{gname}_p"], d[f"{gname}_x0"]
        out = solve(P, X0, N)
        ok = out["status"] == 0
        fs, fo = d[f"{gname}_f"], out["f"]
        feas = (d[f"{gname}_eq"] < 1e-8) & (d[f"{gname}_ineq"] < 1e-8)      # SLSQP ended at a feasible point (it stops on its iteration limit, not on a KKT test)
        same = ok & feas & (np.abs(fs - fo) <= 1e-6)      # absolute: f is ~2e3 on the cold batches, a neighbouring minimiser differs by ~1e-3
        dq = (out["x"] - d[f"{gname}_x"]).reshape(len(P), N, 44)[:, :, 8:15]
        rms = np.sqrt((dq ** 2).mean(axis=(1, 2)))
        lower = ok & feas & (fs < fo - 1e-6)
        report[gname] = dict(n=len(P), converged=int(ok.sum()), same_minimiser=int(same.sum()), slsqp_lower=int(lower.sum()),
                             rms_median=float(np.median(rms[same])) if same.any() else None, rms_max=float(rms[same].max()) if same.any() else None)
        # cold starts (SLSQP ran its 1000 iterations): 5e-6 rad; starts in the basin (SLSQP stops after ~25 iterations of a 880 / 1320-variable
        # dense QP sequence when its line search cannot improve, exit 8: its own accuracy there is ~1e-5 rad): the north-star tolerance 1e-4
        near = int(d[f"{gname}_def"][4]) == 1
        assert rms[same].max() < (1e-4 if near else 5e-6), (label, gname, rms[same].max())
        assert same.sum() >= 0.9 * ok.sum(), (label, gname, report[gname])
        if int(d[f"{gname}_def"][4]) == 0:      # cold starts: the solver itself must have converged on every problem of the sample
            assert ok.all(), (label, gname)
    print(f"\n{label} vs independent SLSQP solutions of the benchmark batches:", report)
    return report


def test_oracle_against_independent_slsqp_solutions_of_the_benchmark_batches():
    rep = check_against_slsqp_batch(lambda P, X0, N: c_oracle.solve(P, X0, N, 4, 0.1, nthreads=4), "CPU oracle")
    # configs[1] sample (64 problems, cold start): 61 at SLSQP's minimiser, on 2 SLSQP found a lower one (1e-3 relative), on 1 a higher one
    assert rep["c1"]["n"] == 64 and rep["c1"]["same_minimiser"] >= 60 and rep["c1"]["slsqp_lower"] <= 3


def test_hard_closed_loop_ticks_g13():
    """Fixture g13 (tests/cpu_closed_loop.py): the ticks on which the closed loops of BASELINE configs[4] failed in round 4 -- found by replaying 64 of
    the 256 streams on the CPU -- with SLSQP's verdict from the same start: infeasible on eight of the twelve (locally infeasible NLPs: the stream
    has drifted to where the tubes around the near-pi rotation of the path's third segment cannot be met from its state), feasible on four.
    * without the restoration phase the oracle reproduces the statuses the fixture recorded (stalls on ten);
    * with it (round 5, the default) the verdicts are SLSQP's: exactly the four feasible ticks converge, to SLSQP's minimiser and objective, and the
      eight infeasible ones end as status 2 (restoration converged to a non-zero violation, or its budget) within 60 iterations."""
    # (start_rollout = 0 throughout: in the loop these ticks are warm solves -- a dual state travels with the plan -- and are never rolled out; solved cold here)
    d = np.load(os.path.join(G, "g13_hard_ticks.npz"))
    old = c_oracle.solve(d["p"], d["x0"], 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=100, restoration=0, start_rollout=0), nthreads=4)
    assert np.array_equal(old["status"], d["oracle_status"])
    slsqp_feasible = (d["slsqp_eq"] < 1e-6) & (d["slsqp_ineq"] < 1e-6)
    assert slsqp_feasible.sum() == 4
    out = c_oracle.solve(d["p"], d["x0"], 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=500, start_rollout=0), nthreads=4)
    conv = out["status"] == 0
    assert np.array_equal(conv, slsqp_feasible) and out["iters"][conv].max() <= 120
    assert (out["status"][~conv] == 2).all() and out["iters"][~conv].max() <= 60
    dq = (out["x"][conv] - d["slsqp_x"][conv]).reshape(-1, 10, 44)[:, :, 8:15]
    assert np.sqrt((dq ** 2).mean(axis=(1, 2))).max() < 5e-6      # same minimiser as SLSQP
    assert np.abs((out["f"][conv] - d["slsqp_f"][conv]) / d["slsqp_f"][conv]).max() < 1e-8


def test_first_failures_of_all_256_closed_loops_g13b():
    """Fixture g13b (tests/cpu_closed_loop.py --streams 256 --first-only): EVERY first failing tick of the 256 closed loops of BASELINE configs[4]
    over 130 ticks as round 4 ran them (38 ticks on 31 streams), with SLSQP's end point from the same start: infeasible on 28, feasible on 10.
    Round 4 (no restoration phase): the oracle stalls on all 38; with the stall test off and Ipopt's iteration limit it converges on 8 of the 10
    feasible ones in 63-340 iterations and on none of the 28.
    Round 5 (restoration phase, the default -- oracle/bmpc_oracle.c solve_one): AT THE DEFAULTS
    * the 28 locally infeasible problems end as status 2 within 60 iterations (22-54), no numerical breakdowns;
    * 8 of the 10 feasible ones converge within 130 iterations (64-128), to SLSQP's objective on seven (SLSQP higher by 9e-5 on one: the oracle's
      point is the better minimiser); the other two -- the two on which the patient solver of round 4 failed as well -- end as status 2 after the
      third restoration phase (123 / 129 iterations)."""
    # (start_rollout = 0 throughout: in the loop these ticks are warm solves and are never rolled out; solved cold here)
    d = np.load(os.path.join(G, "g13b_first_failures_256_streams.npz"))
    assert len(d["stream"]) == 38 and len(set(d["stream"].tolist())) == 31
    old = c_oracle.solve(d["p"], d["x0"], 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=100, restoration=0, start_rollout=0), nthreads=4)
    assert np.array_equal(old["status"], d["oracle_status"]) and (d["oracle_status"] != 0).all()
    feas = (d["slsqp_eq"] < 1e-8) & (d["slsqp_ineq"] < 1e-8) & (d["slsqp_bounds"] < 1e-8)
    assert feas.sum() == 10 and (d["slsqp_exit"][feas] == 0).sum() == 1
    out = c_oracle.solve(d["p"], d["x0"], 10, 4, 0.1, opts=c_oracle.default_opts(max_iter=500, start_rollout=0), nthreads=4)
    st, it = out["status"], out["iters"]
    assert (st[~feas] == 2).all() and it[~feas].max() <= 60
    conv = feas & (st == 0)
    assert conv.sum() >= 8 and it[conv].max() <= 130 and (st[feas & ~conv] == 2).all() and it[feas].max() <= 135
    rel = (out["f"][conv] - d["slsqp_f"][conv]) / d["slsqp_f"][conv]
    assert (np.abs(rel) < 1e-6).sum() >= 7 and rel.max() < 1e-6 and rel.min() > -1e-3      # never worse than SLSQP's point
    patient = c_oracle.default_opts(max_iter=500, stall_window=0, restoration=0, start_rollout=0)      # round 4's patient handle, for the record
    a = c_oracle.solve(d["p"][feas], d["x0"][feas], 10, 4, 0.1, opts=patient, nthreads=4)
    assert (a["status"] == 0).sum() >= 8 and a["iters"][a["status"] == 0].min() >= 60


def test_oracle_converges_from_starts_far_from_the_reference_warm_start():
    """128 feasible N = 10 problems from all zeros / uniform(-1, 1) / the cold start + noise 1.0 on every variable.  Defaults: a cold start that is not a
    trajectory (integrator-chain residual above START_ROLLOUT_TOL) is rolled out with its own jerks before the first iteration -- all converge, from
    zeros like the reference's cold start (its rollout IS that cold start up to round-off).  With x0 taken as given (start_rollout = 0) the restoration phase
    alone rescues 94-100 % in about three times the iterations; with neither (round 4's algorithm) three quarters of the first and all of the others
    stall.  (GPU counterpart: test_the_solver_does_not_depend_on_the_reference_warm_start; the full battery: tests/gpu_robustness.py.)"""
    from boundmpc_amd import workload
    P, X, _ = workload.make_batch(128, seed=60, N=10)
    rng = np.random.default_rng(11)
    cold = c_oracle.solve(P, X, 10, 4, 0.1, nthreads=8)
    for name, X0, lo, hi_off in (("zeros", np.zeros_like(X), 0.94, 0.25), ("uniform", rng.uniform(-1, 1, X.shape), 0.99, 0.02), ("noise 1.0", X + rng.normal(size=X.shape), 0.99, 0.02)):
        o = c_oracle.solve(P, X0, 10, 4, 0.1, nthreads=8)
        assert (o["status"] == 0).all() and o["iters"].mean() <= 22 and o["kkt"].max() <= 1e-8, (name, np.bincount(o["status"]))
        if name == "zeros":
            assert np.abs(o["iters"] - cold["iters"]).max() <= 1 and np.abs(o["x"] - cold["x"]).reshape(-1, 10, 44)[:, :, 8:15].max() < 1e-6      # (FK(q0) in place of the given p0: round-off)
        o1 = c_oracle.solve(P, X0, 10, 4, 0.1, opts=c_oracle.default_opts(start_rollout=0), nthreads=8)
        assert (o1["status"] == 0).mean() >= lo and o1["iters"].max() <= 100 and o1["iters"].mean() >= 1.5 * o["iters"].mean(), (name, np.bincount(o1["status"]))
        o0 = c_oracle.solve(P, X0, 10, 4, 0.1, opts=c_oracle.default_opts(restoration=0, start_rollout=0), nthreads=8)
        assert (o0["status"] == 0).mean() <= hi_off, name


def test_orientation_tube_exits_are_reproduced_by_slsqp_driven_closed_loops_g14():
    """Fixture g14 (tests/golden/make_g14.py): stretches of four closed loops of BASELINE configs[4] driven tick by tick by an INDEPENDENT solver
    (scipy's SLSQP on the reference's constraint form) and, from the same snapshot, by the oracle.  The plant of the two streams that leave the
    ORIENTATION tube in the oracle's loop leaves it under SLSQP as well -- same ticks, same excess -- while every applied plan satisfied its tube rows:
    the NLP constrains the orientation through the per-tick linearisation of the error split (mpc_utils_casadi.py:6-67, casadi_ocp_formulation.py:
    316-349), the measured state is judged by the exact split (stream.tube_excess_of_state).  A property of the formulation, not of the solver."""
    from boundmpc_amd import stream as bstream
    d = np.load(os.path.join(G, "g14_slsqp_closed_loops.npz"))
    assert d["oracle_streams"] == 32 and d["oracle_streams_leaving"] >= 16 and 0.03 < d["oracle_fraction_outside"] < 0.08      # the 130-tick oracle loops the stretches were cut from
    for b, leaves, first in zip(d["streams"], d["leaves"], d["first_exit"] - d["start_tick"]):
        for solver in ("oracle", "slsqp"):      # the stored excess rows are what tube_excess_of_state gives for the stored packed problems
            ex_p, ex_r = bstream.tube_excess_of_state(d[f"{solver}_p_{b}"])
            np.testing.assert_allclose(ex_r, d[f"{solver}_ex_r_{b}"], atol=1e-12); np.testing.assert_allclose(ex_p, d[f"{solver}_ex_p_{b}"], atol=1e-12)
        eo, es = d[f"oracle_ex_r_{b}"].max(axis=1), d[f"slsqp_ex_r_{b}"].max(axis=1)
        dq = np.abs(d[f"oracle_q_{b}"] - d[f"slsqp_q_{b}"]).max(axis=1)
        ex_pos = d[f"slsqp_ex_p_{b}"].max()
        assert ex_pos <= 1e-9      # the POSITION tube is never left (its rows are exact in the NLP)
        if leaves:
            # up to and including the first sample outside the tube the two plants are the same plant (SLSQP stops at its iteration limit on
            # most ticks: ~1e-7 rad), and the excess of that sample is the same
            assert dq[:first + 1].max() < 5e-6
            np.testing.assert_allclose(es[:first + 2], eo[:first + 2], atol=2e-4)
            assert es[first] > 0 and es.max() > 0.01 and (es > 0).sum() >= 3
            assert d[f"slsqp_applied_{b}"][:first].all()      # every plan ahead of the exit was a solution SLSQP itself calls feasible (violations < 1e-6) and was applied
            assert d[f"slsqp_eq_{b}"][:first].max() < 1e-6 and d[f"slsqp_iq_{b}"][:first].max() < 1e-6
        else:
            assert es.max() < 0 and eo.max() < 0 and dq.max() < 1e-3
            np.testing.assert_allclose(es, eo, atol=2e-4)
