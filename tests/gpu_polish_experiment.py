"""Diagnostic (GPU box), verdict round 5 item 6c: "a polish from the level's iterate toward tolerance for the plan that is applied, while the level's iterate
stays the carrier of the warm start".  256 streams x 130 ticks with the three-kernel tick (pack, solves, post; not fused: the tick times are upper
bounds): per tick KL Newton steps on the held level (the carrier: iterate + dual state), then P more steps FROM the carrier with the barrier walking down
(a second handle, a copy of the dual state); the polished plan is judged and applied, the carrier (variant "carrier") or the accepted plan (variant
"plan") is what the next tick shifts.  Reported against the loops solved to 1e-8: plans kept, streams within 1e-2 rad RMS, tube excess.
Usage: python tests/gpu_polish_experiment.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from boundmpc_amd import BatchedOCPSolver, stream as bstream, workload  # noqa: E402

B, T, N = 256, 131, 10
q0s = workload.random_q0(256, seed=3)
mpcs, recs = [], []
for q0 in q0s:
    m, p0fk = workload.make_mpc(q0)
    mpcs.append(m)
    recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
recs = np.stack(recs)
torch.cuda.set_stream(torch.cuda.Stream())
FLAG = 32 + 56 * N + 1      # ss_updated(N) + 1: "the last iterate was rejected: continue from xlast"


def loop(mode, KL=5, P=0, carrier=False, level="auto", pol_tol=1e-3):
    conv = mode == "converged"
    slv = BatchedOCPSolver(10, 4, 0.1, max_iter=100, stall_window=16) if conv else BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=30, fixed_barrier=level, bound_margin=2e-3)
    pol = BatchedOCPSolver(10, 4, 0.1, tol=pol_tol, max_iter=30, mu_init=0.1, mu_warm=1e-5, bound_margin=2e-3) if P else None
    for s in (slv, pol):
        if s is not None:
            s.set_rt_feasibility_tol(1e-2); s.set_rt_position_row_cap(1e-5); s.set_start_rollout(False)
    sb = bstream.StreamBatch(slv, mpcs); sb.set_robot(recs)
    Q, ms, tube = [], [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    outp = {}
    for t in range(T):
        if t == 0 or conv:
            sb.tick(max_iter=100 if t == 0 else 0, warm_dual=True, simulate=True, fused=False, accept_capped=not conv and t > 0)
        else:
            e0.record()
            if carrier:
                sb.state[:, FLAG] = 1.0      # always continue from sb.x (the carrier)
            sb.pack(True, None, continue_rejected=True)
            out = dict(x=sb.x, g=sb.g, iters=sb.iters, status=sb.status, kkt=sb.kkt)
            slv.solve_batch(sb.p, sb.x0, out=out, want=("g", "iters", "status", "kkt"), state=sb.dual, max_iter=KL)      # the carrier: sb.x, sb.dual
            if P:
                stp = sb.dual.clone()
                o = pol.solve_batch(sb.p, sb.x, out=outp, want=("g", "iters", "status", "kkt"), state=stp, max_iter=P)
                xc = sb.x.clone()
                sb.x.copy_(o["x"]); sb.g.copy_(o["g"]); sb.status.copy_(o["status"])
                sb.post(True, None, True)
                if carrier:
                    sb.x.copy_(xc)
            else:
                sb.post(True, None, True)
            e1.record(); e1.synchronize(); ms.append(e0.elapsed_time(e1))
        torch.cuda.synchronize()
        Q.append(sb.robot[:, :7].cpu().numpy().copy())
        if t > 0:
            has_plan = (sb.state[:, bstream.SS["ERRCNT"]] < 10).cpu().numpy()
            ex_p, ex_r = bstream.tube_excess_of_state(sb.p.cpu().numpy())
            tube.append(np.where(has_plan[:, None], ex_p, -np.inf).max())
    alive = float((sb.state[:, bstream.SS["VALID"]] > 0.5).double().mean()); phi = float(sb.state[:, bstream.SS["PHI"]].mean())
    sb.close(); slv.close()
    if pol is not None:
        pol.close()
    return np.array(Q), alive, (np.array(ms) if ms else np.zeros(1)), max(tube), phi


ref, alive, _, tb, phi = loop("converged")
print(f"converged loops: plans kept {100 * alive:.1f} %, mean phi {phi:.2f}, largest position excess {tb:.1e} m", flush=True)
for name, kw in (("level auto, 5 steps (the bench mode)", dict(KL=5)),
                 ("level 3 + polish 2, next tick from the plan", dict(KL=3, P=2)),
                 ("level 3 + polish 3, next tick from the plan", dict(KL=3, P=3)),
                 ("level 3 + polish 3, next tick from the carrier", dict(KL=3, P=3, carrier=True)),
                 ("level 4 + polish 4, next tick from the carrier", dict(KL=4, P=4, carrier=True)),
                 ("level 4 + polish 4, next tick from the plan", dict(KL=4, P=4)),
                 ("level 0.1 fixed 3 + polish 3, carrier", dict(KL=3, P=3, carrier=True, level=0.1))):
    Q, alive, ms, tb, phi = loop("rt", **kw)
    dev = np.sqrt(np.mean((Q - ref) ** 2, axis=(0, 2)))
    print(f"{name:52s}: plans kept {100 * alive:.1f} %, within 1e-2 rad RMS of the converged loops {100 * (dev <= 1e-2).mean():.1f} % (median {np.median(dev):.3f} rad), "
          f"mean phi {phi:.2f}, largest position excess {tb:.1e} m, three-kernel tick p50 {np.percentile(ms, 50):.2f} / p99 {np.percentile(ms, 99):.2f} ms", flush=True)
