"""Diagnostic (GPU box): WHERE does a fixed-barrier closed loop leave the position tube?  256 streams x 130 ticks in the mode bench.py reports for
configs[4] (fixed barrier level 0.1, five Newton steps per tick, acceptance at 1e-2); for the worst plant samples: stream, tick, tube half width there,
the applied / replayed history of the stream around it and the plan's own first-stage position rows.  Usage: python tests/gpu_tube_excess_diag.py [feas_tol] [level] [cap] [position row cap]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from boundmpc_amd import BatchedOCPSolver, stream as bstream, workload  # noqa: E402

FT = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-2
MU = (sys.argv[2] if sys.argv[2] == "auto" else float(sys.argv[2])) if len(sys.argv) > 2 else 0.1
LC = float(os.environ.get("LEVEL_C", "0.1"))
CAP = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ROWCAP = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
B, T = 256, 131
q0s = workload.random_q0(256, seed=3)
mpcs, recs = [], []
for q0 in q0s:
    m, p0fk = workload.make_mpc(q0)
    mpcs.append(m)
    recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
slv = BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=30, fixed_barrier=MU, bound_margin=2e-3, level_c=LC)
slv.set_rt_feasibility_tol(FT)
slv.set_rt_position_row_cap(ROWCAP)
torch.cuda.set_stream(torch.cuda.Stream())
sb = bstream.StreamBatch(slv, mpcs)
sb.set_robot(np.stack(recs))
exp, app, gv, rows, wid, errc, phis = [], [], [], [], [], [], []
for t in range(T):
    if t == 0:
        sb.tick(max_iter=100, warm_dual=True, simulate=True)
    else:
        sb.tick_graph(max_iter=CAP, warm_dual=True, simulate=True, accept_capped=True)
    torch.cuda.synchronize()
    if t > 0:
        p = sb.p.cpu().numpy()
        ex_p, ex_r = bstream.tube_excess_of_state(p)
        l, w = bstream.tube_excess_of_state(p, rows=True)
        has_plan = (sb.state[:, bstream.SS["ERRCNT"]] < 10).cpu().numpy()
        exp.append(np.where(has_plan[:, None], ex_p, -np.inf).max(axis=1)); wid.append(np.abs(w[:, 1:3]).min(axis=1))
    app.append((sb.traj[:, -2] > 0.5).cpu().numpy()); gv.append(sb.traj[:, -1].cpu().numpy())
    rows.append(sb.g.reshape(B, 10, 43)[:, :, 39:41].max(dim=2).values.cpu().numpy())      # position rows (l^2 - w^2) of every stage of the iterate
    errc.append(sb.state[:, bstream.SS["ERRCNT"]].cpu().numpy().copy()); phis.append(sb.state[:, bstream.SS["PHI"]].cpu().numpy().copy())
exp = np.array(exp); app = np.array(app); gv = np.array(gv); rows = np.array(rows); wid = np.array(wid); errc = np.array(errc); phis = np.array(phis)
n = np.isfinite(exp).sum()
print(f"feas tol {FT:g}, level {MU} (c = {LC:g}), {CAP} steps, position row cap {ROWCAP:g}: {100 * (exp > 1e-6).sum() / n:.3f} % of {n} plant samples outside the position tube, largest excess {exp.max():.2e} m; "
      f"applied ticks {100 * app[1:].mean():.1f} %, streams with a plan at the end {100 * float((sb.state[:, bstream.SS['VALID']] > 0.5).double().mean()):.1f} %; mean phi at the end {float(sb.state[:, bstream.SS['PHI']].mean()):.3f}")
order = np.dstack(np.unravel_index(np.argsort(-exp, axis=None), exp.shape))[0][:6]
for ti, b in order:
    t = ti + 1      # exp[ti] is the plant state packed at tick ti + 1 = the state the plan applied at tick ti led to
    lo = max(t - 6, 0)
    print(f"  stream {b} state of tick {t}: excess {exp[ti, b]:.2e} m, narrowest half width there {wid[ti, b]:.3f} m, phi {phis[t, b]:.3f}; history ticks {lo}..{t}: "
          f"applied {app[lo:t + 1, b].astype(int).tolist()} errcnt {errc[lo:t + 1, b].astype(int).tolist()} g_viol {['%.1e' % v for v in gv[lo:t + 1, b]]}; "
          f"largest position row of the iterate per tick (m^2, over the stages) {['%.1e' % v for v in rows[lo:t + 1, b].max(axis=1)]}, first stage {['%.1e' % v for v in rows[lo:t + 1, b, 0]]}")
sb.close(); slv.close()
