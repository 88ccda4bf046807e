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
