"""ORACLE (test infrastructure only): an INDEPENDENT solve of the reference NLP.

scipy's SLSQP (a dense SQP, unrelated to the build's interior-point/Riccati algorithm) is
run on the numpy restatement of the NLP in the reference's own form -- squared tube
constraints, lbg <= g <= ubg, lbx <= x <= ubx, cold start of BoundMPC.py:316-321 -- with
complex-step derivatives.  Its solution certifies that the build's solver lands on the same
local minimiser an independent NLP method finds (Ipopt itself cannot be run here;
SURVEY.md 8c).  Slow (minutes to hours); used only to create tests/golden/g8_scipy_*.npz:
  python oracle/solve_scipy.py                 tick 0 of both experiments (cold start)
  python oracle/solve_scipy.py --ticks         warm-started closed-loop ticks of fixture G7 (segment switch inside the horizon,
                                               the integrated-omega unwrap tick, active asymmetric +-0.01 tube, phi_max active),
                                               each from the x0 the reference's step() handed to the solver; one process per tick
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
            gf, Jg = nlp.jac_g_banded_complex_step(x, p, N, S, h)
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


TICKS = {1: [1, 20, 37, 45, 46, 49, 99, 143], 2: [1, 10, 12, 13, 29, 40, 47]}


def _one_tick(job):
    which, t, G = job
    d = np.load(os.path.join(G, f"g7_closedloop_exp{which}.npz"))
    p, x0 = d["p"][t], d["x0"][t]
    t0 = time.time()
    res = solve(p, x0, maxiter=600, verbose=False)
    f, g = nlp.nlp_eval(res.x, p, 10, 4, 0.1)
    print(f"exp{which} tick {t}: {res.message} nit {res.nit} {time.time() - t0:.0f}s f {f:.10g} "
          f"eq {np.abs(g.reshape(10, 43)[:, :36]).max():.1e} ineq {g.reshape(10, 43)[:, 36:].max():.1e}", flush=True)
    return which, t, res.x, f, res.nit, bool(res.success), p, x0


def solve_ticks(G, nproc=7):
    import multiprocessing as mp
    jobs = [(w, t, G) for w in TICKS for t in TICKS[w]]
    with mp.Pool(nproc) as pool:
        out = pool.map(_one_tick, jobs, chunksize=1)
    np.savez_compressed(os.path.join(G, "g8_scipy_ticks.npz"), exp=np.array([o[0] for o in out]), tick=np.array([o[1] for o in out]),
                        x=np.array([o[2] for o in out]), f=np.array([o[3] for o in out]), nit=np.array([o[4] for o in out]),
                        success=np.array([o[5] for o in out]), p=np.array([o[6] for o in out]), x0=np.array([o[7] for o in out]))


if __name__ == "__main__":
    G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    if "--ticks" in sys.argv:
        solve_ticks(G)
        sys.exit(0)
    for which in (1, 2):
        d = np.load(os.path.join(G, f"g6_pack_exp{which}_tick0.npz"))
        res = solve(d["p_f64"], d["x0_f64"])
        f, g = nlp.nlp_eval(res.x, d["p_f64"], 10, 4, 0.1)
        np.savez_compressed(os.path.join(G, f"g8_scipy_exp{which}_tick0.npz"), x=res.x, f=f, g=g, nit=res.nit,
                            success=res.success, p=d["p_f64"], x0=d["x0_f64"])
        print("exp", which, "f", f, "max eq viol", np.abs(g.reshape(10, 43)[:, :36]).max(),
              "max ineq", g.reshape(10, 43)[:, 36:].max())
