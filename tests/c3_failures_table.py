"""Diagnostic (CPU): table of the problems of configs[3] that end with status 2 at the handle's defaults -- what the full restoration phase makes of them
(gpurun_out/c3_failures.npz, tests/gpu_c3_failures.py) and an INDEPENDENT evaluation of the reference-form constraints (oracle/nlp.py, numpy) at every
final point: bounds lbx <= x <= ubx, lbg <= g <= ubg.  A point that satisfies them to 1e-8 is a certificate that the problem is feasible, however it
was found.  Usage: python tests/c3_failures_table.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import nlp
D = np.load(os.path.join(ROOT, "gpurun_out", "c3_failures.npz"))
N = 30
lbx, ubx, lbg, ubg = nlp.bounds(N)
def viol(x, p):
    f, g = nlp.nlp_eval(x, p, N, 4, 0.1)
    vg = np.maximum(np.maximum(lbg - g, g - ubg), 0.0); vx = np.maximum(np.maximum(lbx - x, x - ubx), 0.0)
    g2 = vg.reshape(N, 43)
    return f, g2[:, :36].max(), g2[:, 36:].max(), vx.max()
print("problem | defaults (mode 2): status its  eq-viol  ineq-viol | mode 1 cap 40: status its | cap 150: status its | cap 400: status its  f  eq-viol  ineq-viol  bound-viol | verdict")
nf = 0
for j, b in enumerate(D["idx"]):
    p = D["p"][j]
    _, e0, i0, _ = viol(D["x_default"][j], p)
    f4, e4, i4, b4 = viol(D["x_m1_cap400"][j], p)
    feas = max(e4, i4, b4) <= 1e-8 and D["st_m1_cap400"][j] == 0
    nf += feas
    print(f"{b:5d} | {D['st_default'][j]} {D['it_default'][j]:4d} {e0:.1e} {i0:.1e} | {D['st_m1_cap40'][j]} {D['it_m1_cap40'][j]:4d} | {D['st_m1_cap150'][j]} {D['it_m1_cap150'][j]:4d} | "
          f"{D['st_m1_cap400'][j]} {D['it_m1_cap400'][j]:4d} {f4:.6g} {e4:.1e} {i4:.1e} {b4:.1e} | "
          + ("FEASIBLE: a KKT point to 1e-8 whose constraints hold (independent evaluation)" if feas else f"no feasible point found: the l1 phase stops at violation eq {e4:.1e} / ineq {i4:.1e}"))
print(f"{nf} of {len(D['idx'])} are feasible problems (solver gap of the default restoration mode for N > 11); {len(D['idx']) - nf} stay without a feasible point at any cap")
