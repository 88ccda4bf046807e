"""Diagnostic (CPU, hours): scipy SLSQP (oracle/solve_scipy_batch.solve: dense SQP on the reference's constraint form, nothing in common with the product's
algorithm) started from the point where the l1 restoration phase stopped, for the problems of configs[3] that keep status 2 under every restoration cap
(gpurun_out/c3_failures.npz): does an independent method find a feasible point nearby?  Usage: python tests/c3_failures_slsqp.py [procs] [maxiter]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import nlp, solve_scipy_batch as ssb
D = np.load(os.path.join(ROOT, "gpurun_out", "c3_failures.npz"))
N = 30
lbx, ubx, lbg, ubg = nlp.bounds(N)
def job(j):
    p = D["p"][j]; x0 = np.clip(D["x_m1_cap400"][j], lbx, ubx)
    t0 = time.time()
    res = ssb.solve(p, x0, N=N, maxiter=int(sys.argv[2]) if len(sys.argv) > 2 else 150)
    f, g = nlp.nlp_eval(res.x, p, N, 4, 0.1)
    vg = np.maximum(np.maximum(lbg - g, g - ubg), 0.0).reshape(N, 43)
    line = f"problem {D['idx'][j]}: SLSQP status {res.status} ({res.message}) nit {res.nit} {time.time() - t0:.0f} s: f {f:.6g} eq-viol {vg[:, :36].max():.1e} ineq-viol {vg[:, 36:].max():.1e}; moved {np.abs(res.x - x0).max():.2e} from the start"
    print(line, flush=True)
    return line
if __name__ == "__main__":
    import multiprocessing as mp
    todo = [j for j in range(len(D["idx"])) if D["st_m1_cap400"][j] != 0]
    with mp.get_context("fork").Pool(int(sys.argv[1]) if len(sys.argv) > 1 else 6) as pool:
        lines = list(pool.imap_unordered(job, todo, chunksize=1))
    open(os.path.join(ROOT, "gpurun_out", "c3_failures_slsqp.txt"), "w").write("\n".join(sorted(lines)) + "\n")
