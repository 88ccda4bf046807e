"""ORACLE (test infrastructure only): an INDEPENDENT solve of the reference NLP.

scipy's SLSQP (a dense SQP, unrelated to the build's interior-point/Riccati algorithm) is
run on the numpy restatement of the NLP in the reference's own form -- squared tube
constraints, lbg <= g <= ubg, lbx <= x <= ubx, cold start of BoundMPC.py:316-321 -- with
complex-step derivatives.  Its solution certifies that the build's solver lands on the same
local minimiser an independent NLP method finds (Ipopt itself cannot be run here;
SURVEY.md 8c).  Slow (minutes); used only to create tests/golden/g8_scipy_exp*_tick0.npz.
"""
import os
import sys
import time

import numpy as np
from scipy.optimize import minimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nlp  # noqa: E402


def solve(p, x0, N=10, S=4, h=0.1, maxiter=400, ftol=1e-14, verbose=True):
    lbx, ubx, lbg, ubg = nlp.bounds(N)
    cache = {}

    def ev(x):
        key = x.tobytes()
        if key not in cache:
            cache.clear()
            f, g = nlp.nlp_eval(x, p, N, S, h)
            gf, Jg = nlp.jac_g_complex_step(x, p, N, S, h)
            cache[key] = (f, g, gf, Jg)
        return cache[key]
    eq = np.where(lbg == ubg)[0]
    iq = np.where(lbg != ubg)[0]
    cons = [dict(type="eq", fun=lambda x: ev(x)[1][eq], jac=lambda x: ev(x)[3][eq]),
            dict(type="ineq", fun=lambda x: -ev(x)[1][iq], jac=lambda x: -ev(x)[3][iq])]
    t0 = time.time()
    res = minimize(lambda x: ev(x)[0], x0, jac=lambda x: ev(x)[2], bounds=list(zip(lbx, ubx)), constraints=cons,
                   method="SLSQP", options=dict(maxiter=maxiter, ftol=ftol, disp=verbose))
    if verbose:
        print("SLSQP:", res.message, "iters", res.nit, "time %.1fs" % (time.time() - t0))
    return res


if __name__ == "__main__":
    G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    for which in (1, 2):
        d = np.load(os.path.join(G, f"g6_pack_exp{which}_tick0.npz"))
        res = solve(d["p_f64"], d["x0_f64"])
        f, g = nlp.nlp_eval(res.x, d["p_f64"], 10, 4, 0.1)
        np.savez_compressed(os.path.join(G, f"g8_scipy_exp{which}_tick0.npz"), x=res.x, f=f, g=g, nit=res.nit,
                            success=res.success, p=d["p_f64"], x0=d["x0_f64"])
        print("exp", which, "f", f, "max eq viol", np.abs(g.reshape(10, 43)[:, :36]).max(),
              "max ineq", g.reshape(10, 43)[:, 36:].max())
