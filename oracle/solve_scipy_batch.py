"""ORACLE (test infrastructure only): INDEPENDENT solutions of the BENCHMARK workloads.

oracle/solve_scipy.py pins the two experiment paths; the benchmark batches (random q0, boundmpc_amd.workload.make_batch --
SURVEY.md 8d configs 2-4) had no checker other than the oracle, which shares the product's algorithm.  This script runs scipy's
SLSQP (dense SQP on oracle/nlp.py in the reference's own constraint form, complex-step derivatives; nothing in common with the
interior-point / Riccati solver) from the SAME cold start on a sample of those batches and stores, per problem, SLSQP's solution,
objective, constraint violation and verdict -> tests/golden/g8_scipy_batch.npz.

  python oracle/solve_scipy_batch.py [--procs 6] [--only c1|c3|n20]     # resumable: finished problems are kept in gpurun_out/-free
                                                                         # partial files under /tmp/g8_batch/

Groups (group id, N, tight, seed, rows of the seeded batch):
  c1   N=10 loose, seed 0 (BASELINE configs[1]): problems 0..63 of the 1024
  c3   N=30 tight, seed 2 (configs[3]):          problems 0..7 of the 8192 (start "near", see GROUPS)
  n20  N=20 tight, seed 26(the tight soak of tests/test_gpu_soak.py): problems 0..15 of 512 (start "near")
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nlp, c_oracle  # noqa: E402
from scipy.optimize import minimize  # noqa: E402


def jac_g_banded_fd(x, p, N, S=4, h=0.1, eps=1e-6):
    """dg/dx by central differences of the C restatement's g (oracle/bmpc_oracle.c bmpc_oracle_eval, pinned by fixture G9), two evaluations
    per colour: stage k's constraints depend on z_{k-1}, z_k only, so entry j of every second stage is perturbed together (88 colours, as
    nlp.jac_g_banded_complex_step).  Agrees with the complex-step Jacobian through oracle/nlp.py to 3e-10 (checked when the fixture was made);
    1.1 s -> 13 ms per Jacobian at N = 10, which is what makes a batch of SLSQP solves affordable."""
    n = x.size
    Jg = np.zeros((43 * N, n))
    for par in (0, 1):
        ks = np.arange(par, N, 2)
        for j in range(44):
            xp = x.copy(); xm = x.copy(); xp[ks * 44 + j] += eps; xm[ks * 44 + j] -= eps
            d = (c_oracle.eval_fg(p, xp, N, S, h)[1] - c_oracle.eval_fg(p, xm, N, S, h)[1]).reshape(N, 43) / (2 * eps)
            for k in ks:
                Jg[k * 43:(k + 1) * 43, k * 44 + j] = d[k]
                if k + 1 < N:
                    Jg[(k + 1) * 43:(k + 2) * 43, k * 44 + j] = d[k + 1]
    return Jg


def solve(p, x0, N=10, S=4, h=0.1, maxiter=800, ftol=1e-14):
    """SLSQP on the reference's NLP form (lbx <= x <= ubx, lbg <= g <= ubg with the squared tube rows), f and g from the C restatement,
    grad f analytic (bmpc_oracle_adjoint with zero multipliers; equal to the complex step through nlp.py to 1e-14), dg/dx by
    central differences.  The SOLVER is independent of the product's algorithm; the function evaluations are the oracle's."""
    lbx, ubx, lbg, ubg = nlp.bounds(N)
    cache = {}

    def ev(x):
        key = x.tobytes()
        if key not in cache:
            cache.clear()
            f, g = c_oracle.eval_fg(p, x, N, S, h)
            gf = c_oracle.adjoint(p, x, np.zeros(N * 57), N, S, h)[2]
            cache[key] = (f, g.copy(), gf, jac_g_banded_fd(x, p, N, S, h))
        return cache[key]
    eq = np.where(lbg == ubg)[0]
    iq = np.where(lbg != ubg)[0]
    cons = [dict(type="eq", fun=lambda x: ev(x)[1][eq], jac=lambda x: ev(x)[3][eq]),
            dict(type="ineq", fun=lambda x: -ev(x)[1][iq], jac=lambda x: -ev(x)[3][iq])]
    return minimize(lambda x: ev(x)[0], x0, jac=lambda x: ev(x)[2], bounds=list(zip(lbx, ubx)), constraints=cons,
                    method="SLSQP", options=dict(maxiter=maxiter, ftol=ftol, disp=False))

# start: "cold" = the reference's cold start (BoundMPC.py:316-321), the start the product gets; "near" = the oracle's solution plus 1e-3 of
# seeded noise on every variable (dense SLSQP costs O(n^3) per iteration: a cold start takes ~5 h per problem at N = 30, a start in the
# basin 20 min) -- "near" certifies that the point the oracle found is the local minimiser of its basin according to an independent
# method, not that an independent method reaches that basin from the cold start
GROUPS = {"c1": dict(N=10, tight=False, seed=0, B=1024, n=64, start="cold"),
          "n20": dict(N=20, tight=True, seed=26, B=512, n=16, start="near"),
          "c3": dict(N=30, tight=True, seed=2, B=8192, n=8, start="near")}
TMP = "/tmp/g8_batch"


def _job(job):
    name, i, p, x0, N = job
    out = os.path.join(TMP, f"{name}_{i:03d}.npz")
    if os.path.exists(out):
        return out
    t0 = time.time()
    res = solve(p, x0, N=N, maxiter=2500 if N > 10 else 1000)
    f, g = nlp.nlp_eval(res.x, p, N, 4, 0.1)
    g2 = g.reshape(N, 43)
    np.savez_compressed(out, x=res.x, f=f, nit=res.nit, success=bool(res.success), status=res.status,
                        eq=np.abs(g2[:, :36]).max(), ineq=g2[:, 36:].max(), secs=time.time() - t0)
    print(f"{name} {i}: {res.message} nit {res.nit} {time.time() - t0:.0f}s f {f:.10g} eq {np.abs(g2[:, :36]).max():.1e} "
          f"ineq {g2[:, 36:].max():.1e}", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--only", default="")
    ap.add_argument("--collect", action="store_true", help="only gather the finished partial files into the fixture")
    a = ap.parse_args()
    import multiprocessing as mp
    from boundmpc_amd import workload
    os.makedirs(TMP, exist_ok=True)
    jobs, meta = [], {}
    for name, gdef in GROUPS.items():
        if a.only and name not in a.only.split(","):
            continue
        P, X, _ = workload.make_batch(gdef["B"], seed=gdef["seed"], N=gdef["N"], tight=gdef["tight"], rows=(0, gdef["n"]), workers=1)
        meta[name] = (P, X, gdef)
        X0 = X
        if gdef["start"] == "near":
            ref = c_oracle.solve(P, X, gdef["N"], 4, 0.1, nthreads=4)
            X0 = ref["x"] + 1e-3 * np.random.default_rng(1000 + gdef["seed"]).standard_normal(X.shape)
            X0 = np.where((ref["status"] == 0)[:, None], X0, X)          # a problem the oracle did not solve is started cold
            meta[name] = (P, X0, gdef)
        jobs += [(name, i, P[i], X0[i], gdef["N"]) for i in range(gdef["n"])]
    if not a.collect:
        with mp.get_context("fork").Pool(a.procs) as pool:
            for _ in pool.imap_unordered(_job, jobs, chunksize=1):
                pass
    out = {}
    for name, (P, X, gdef) in meta.items():
        done = [i for i in range(gdef["n"]) if os.path.exists(os.path.join(TMP, f"{name}_{i:03d}.npz"))]
        if not done:
            continue
        d = [np.load(os.path.join(TMP, f"{name}_{i:03d}.npz")) for i in done]
        out[f"{name}_idx"] = np.array(done)
        out[f"{name}_p"] = P[done]; out[f"{name}_x0"] = X[done]
        for k in ("x", "f", "nit", "success", "status", "eq", "ineq", "secs"):
            out[f"{name}_{k}"] = np.array([e[k] for e in d])
        out[f"{name}_def"] = np.array([gdef["N"], int(gdef["tight"]), gdef["seed"], gdef["B"], int(gdef["start"] == "near")])
    dst = os.path.join(ROOT, "tests", "golden", "g8_scipy_batch.npz")
    if os.path.exists(dst):       # keep the groups a partial run did not touch
        old = np.load(dst)
        for k in old.files:
            if k.split("_")[0] not in meta:
                out[k] = old[k]
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items() if k.endswith("_f")})


if __name__ == "__main__":
    main()
