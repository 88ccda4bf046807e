#!/usr/bin/env python3
"""Fixture G9: the NLP itself, as the reference's own builder evaluates it.

Runs the reference's UNMODIFIED `setup_optimization_problem` (casadi_ocp_formulation.py:9-391 -> reference_function,
error_function, objective_function, decomp_function, integration_function, RobotModel.fk_pos/velocity_ee/omega_ee,
calcAngle/Velocity/Acceleration) with numbers in place of symbols (ref_nlp.py / numeric_sx.py) and records, per case,
(x, p) -> f, g; once per (N, S): lbx, ubx, lbg, ubg and the reference symbol behind every entry of x and p (the
505-offset order of casadi_ocp_formulation.py:361-376); and for a few cases the exact gradient of f and Jacobian of g
obtained by a complex step through the reference's code.

Cases (N=10,S=4 unless noted):
  physical   (x, p) of recorded closed-loop ticks of experiment 1 / 2 (p assembled by the reference's own step(),
             fixture G7): solutions, their warm starts, and perturbed points; ticks with a segment switch inside the
             horizon, with the sigmoid of the objective active (end of path), with experiment 2's +-0.01 tube
  random     every symbol random (phi_switch increasing), phi_k spread over every segment and beyond phi_switch[S]
  N=30       random + physical-parameter cases for the long horizon;  S=2,3,5 and N=3: random cases

Row S of a4..a0: the reference leaves it uninitialised (np.empty, BoundMPC.py:235-240) and reads it once phi >= phi_switch[S].
The build defines it as a copy of row S-1; the cases here carry that copy in p, so both sides read the same numbers.

Run in the build container only:  python tests/golden/make_g9.py
"""
import os
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from ref_nlp import RefNlp  # noqa: E402

NZ = 44


def p_offsets(r):
    """offset of the first element of every reference symbol in p, from the learned layout."""
    off = {}
    for k, (i, rr, cc) in enumerate(r.p_map):
        off.setdefault(i, k)
    return {r.sym[i][0] + "#%d" % i: k for i, k in off.items()}


def fill_row_S(r, p):
    """a4..a0 row S := row S-1 (see header)."""
    p = p.copy()
    pos = {(i, rr, cc): k for k, (i, rr, cc) in enumerate(r.p_map)}
    for i, (name, n, m) in enumerate(r.sym):
        if name.startswith("parameter ") and name.endswith("error function"):
            for c in range(m):
                p[pos[(i, n - 1, c)]] = p[pos[(i, n - 2, c)]]
    return p


def sym_index(r, name, occurrence=0):
    return [i for i, s in enumerate(r.sym) if s[0] == name][occurrence]


def random_case(r, rng, past_end=True):
    """every symbol random; path parameters of the stages spread over all segments (and past the last switch)."""
    N, S = r.N, r.S
    p = rng.normal(size=r.n_p)
    x = rng.normal(size=r.n_x)
    pos = {(i, rr, cc): k for k, (i, rr, cc) in enumerate(r.p_map)}
    isw = sym_index(r, "path parameter switch")
    sw = np.concatenate([[0.0], np.cumsum(rng.uniform(0.3, 2.0, S))])
    for j in range(S + 1):
        p[pos[(isw, j, 0)]] = sw[j]
    top = sw[-1] + (0.6 if past_end else -1e-3)
    phis = np.sort(rng.uniform(0.0, top, N))
    # a few stages exactly on a switch value: `phi < phi_switch[i+1]` is strict
    if N >= 6:
        phis[N // 2] = sw[min(2, S)]
    z = x.reshape(N, NZ)
    z[:, 41] = phis
    iphimax = sym_index(r, "max path parameter", 0)
    p[pos[(iphimax, 0, 0)]] = phis[int(0.7 * N)] + 0.015        # sigmoid mid-range on some stages, 0/1 on others
    # unit tangents / bases are not required by the formulas; keep them O(1) random
    return fill_row_S(r, p), z.reshape(-1)


def main():
    rng = np.random.default_rng(9)
    out = {}
    g7 = {1: np.load(os.path.join(OUT, "g7_closedloop_exp1.npz")), 2: np.load(os.path.join(OUT, "g7_closedloop_exp2.npz"))}

    # ---------------- N=10, S=4 -------------------------------------------------------------------
    r = RefNlp(10, 4, 0.1)
    X, P, tag = [], [], []
    for which, d in g7.items():
        T = d["x"].shape[0]
        sect = d["sector"]
        switch_ticks = [int(t) for t in np.nonzero(np.diff(sect))[0]]
        ticks = sorted(set([0, 1, 2, 7, T // 3, T // 2, T - 12, T - 6, T - 3, T - 1]
                           + switch_ticks + [t + 1 for t in switch_ticks if t + 1 < T]
                           + [max(t - 4, 0) for t in switch_ticks]))
        for t in ticks:
            p = fill_row_S(r, d["p"][t])
            for kind, x in (("sol", d["x"][t]), ("warm", d["x0"][t]),
                            ("pert", d["x"][t] + 0.05 * rng.normal(size=440))):
                X.append(x); P.append(p); tag.append(f"exp{which}_tick{t}_{kind}")
    for i in range(24):
        p, x = random_case(r, rng, past_end=(i % 3 != 0))
        X.append(x); P.append(p); tag.append(f"random{i}")
    X, P = np.array(X), np.array(P)
    F = np.zeros(len(X)); G = np.zeros((len(X), r.n_g))
    for i in range(len(X)):
        F[i], G[i] = r(X[i], P[i])
    assert np.isfinite(F).all() and np.isfinite(G).all()
    out.update(n10_x=X, n10_p=P, n10_f=F, n10_g=G, n10_tag=np.array(tag),
               n10_lbx=r.lbx, n10_ubx=r.ubx, n10_lbg=r.lbg, n10_ubg=r.ubg,
               n10_x_names=np.array(r.names("x")), n10_p_names=np.array(r.names("p")),
               n10_g_names=np.array(r.g_names))
    # exact derivatives through the reference's code at three points (a solution incl. switch, a perturbed point, random)
    dsel = [tag.index("exp1_tick0_sol"), tag.index("exp2_tick7_pert"), tag.index("random1")]
    # recorded SOLUTIONS whose KKT conditions the tests certify with these derivatives: ticks around a segment switch, the end of the
    # path (phi_max active, sigmoid on), experiment 2 with its +-0.01 tube
    for want in ("exp1_tick1_sol", "exp1_tick45_sol", "exp1_tick46_sol", "exp1_tick%d_sol" % (len(g7[1]["x"]) - 12), "exp2_tick0_sol", "exp2_tick13_sol",
                 "exp2_tick14_sol", "exp2_tick%d_sol" % (len(g7[2]["x"]) // 2)):
        if want in tag and tag.index(want) not in dsel:
            dsel.append(tag.index(want))
    switch_sol = [i for i, t in enumerate(tag) if t.endswith("_sol") and t.startswith("exp1")
                  and len(set(np.searchsorted(P[i][89:94], X[i].reshape(10, 44)[:, 41], side="right"))) > 1]
    if switch_sol:
        dsel.append(switch_sol[0])
    GF, JG = [], []
    for i in dsel:
        gf, jg = r.grad_jac(X[i], P[i])
        GF.append(gf); JG.append(jg)
    out.update(n10_deriv_case=np.array(dsel), n10_grad_f=np.array(GF), n10_jac_g=np.array(JG))
    print("N=10: %d cases, %d with derivatives; tags e.g. %s" % (len(X), len(dsel), tag[:3]))

    # ---------------- N=30, S=4 -------------------------------------------------------------------
    r30 = RefNlp(30, 4, 0.1)
    assert r30.names("p") == r.names("p")
    X, P, tag = [], [], []
    for i in range(6):
        p, x = random_case(r30, rng, past_end=(i % 2 == 0))
        X.append(x); P.append(p); tag.append(f"random{i}")
    for which, d in g7.items():           # physical parameters, horizon extended with a perturbed cold start
        for t in (0, d["x"].shape[0] // 2):
            p = fill_row_S(r30, d["p"][t])
            z = np.zeros((30, NZ))
            z[:, 8:15] = d["q"][t]; z[:, 29:35] = d["p_lie"][t]
            z[:10] = d["x"][t].reshape(10, NZ); z[10:] = z[9]
            x = z.reshape(-1) + 0.02 * rng.normal(size=30 * NZ)
            X.append(x); P.append(p); tag.append(f"exp{which}_tick{t}_ext")
    X, P = np.array(X), np.array(P)
    F = np.zeros(len(X)); G = np.zeros((len(X), r30.n_g))
    for i in range(len(X)):
        F[i], G[i] = r30(X[i], P[i])
    assert np.isfinite(F).all() and np.isfinite(G).all()
    out.update(n30_x=X, n30_p=P, n30_f=F, n30_g=G, n30_tag=np.array(tag),
               n30_lbx=r30.lbx, n30_ubx=r30.ubx, n30_lbg=r30.lbg, n30_ubg=r30.ubg)
    print("N=30: %d cases" % len(X))

    # ---------------- other (N, S) ------------------------------------------------------------------
    for (N, S) in ((3, 2), (5, 3), (4, 5)):
        rr = RefNlp(N, S, 0.05)
        X, P = [], []
        for i in range(4):
            p, x = random_case(rr, rng, past_end=(i % 2 == 0))
            X.append(x); P.append(p)
        X, P = np.array(X), np.array(P)
        F = np.zeros(len(X)); G = np.zeros((len(X), rr.n_g))
        for i in range(len(X)):
            F[i], G[i] = rr(X[i], P[i])
        assert np.isfinite(F).all() and np.isfinite(G).all()
        k = f"n{N}s{S}"
        out.update({k + "_x": X, k + "_p": P, k + "_f": F, k + "_g": G, k + "_dt": 0.05,
                    k + "_p_names": np.array(rr.names("p")), k + "_lbx": rr.lbx, k + "_ubx": rr.ubx,
                    k + "_lbg": rr.lbg, k + "_ubg": rr.ubg})
        print(f"N={N},S={S}: {len(X)} cases, n_p={rr.n_p}")
    np.savez_compressed(os.path.join(OUT, "g9_nlp.npz"), **out)
    print("written", os.path.join(OUT, "g9_nlp.npz"), os.path.getsize(os.path.join(OUT, "g9_nlp.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
