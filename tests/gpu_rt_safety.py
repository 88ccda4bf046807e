"""Diagnostic (GPU box): closed-loop safety of the real-time tick modes (BASELINE configs[4]).  256 streams x T ticks, one captured tick per
launch on an explicit stream.  For every (Hessian, iteration cap K, acceptance threshold) the loops are compared with the loops solved
to 1e-8 every tick: tick latency, share of ticks whose iterate was applied, streams that ran out of plan, largest joint deviation,
joint-limit violations of the plant state, share of streams within 1e-2 rad RMS.  Usage: python tests/gpu_rt_safety.py [T] [modes...]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, workload, stream as bstream
from boundmpc_amd.robot_model import RobotModel

T = int(sys.argv[1]) if len(sys.argv) > 1 else 130
B = 256
q0s = workload.random_q0(B, seed=3)
mpcs, recs = [], []
for q0 in q0s:
    m, p0fk = workload.make_mpc(q0)
    mpcs.append(m)
    recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
recs = np.stack(recs)
qlim = np.array(RobotModel().q_lim_upper)
torch.cuda.set_stream(torch.cuda.Stream())


def run(name, slv, cap, warm, capped, tol):
    if tol is not None:
        slv.set_rt_feasibility_tol(tol)
    sb = bstream.StreamBatch(slv, mpcs); sb.set_robot(recs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    Q, ms, ok, valid, viol, its = [], [], [], [], [], []
    for t in range(T):
        if t == 0:
            sb.tick(max_iter=100, warm_dual=True, simulate=True)
            if not warm:
                sb.dual.zero_()
        else:
            e0.record(); sb.tick_graph(max_iter=cap, warm_dual=warm, simulate=True, accept_capped=capped); e1.record(); e1.synchronize()
            ms.append(e0.elapsed_time(e1))
            ok.append((sb.traj[:, -2] > 0.5).double().mean().item()); valid.append((sb.state[:, bstream.SS["VALID"]] > 0.5).double().mean().item())
            viol.append(sb.traj[:, -1].cpu().numpy()); its.append(sb.iters.double().mean().item())
        Q.append(sb.robot[:, :7].clone())
    Q = torch.stack(Q).cpu().numpy()
    phi = sb.state[:, bstream.SS["PHI"]].cpu().numpy()
    sb.close()
    return dict(name=name, Q=Q, ms=np.array(ms), ok=np.array(ok), valid=np.array(valid), viol=np.array(viol), its=np.array(its), phi=phi)


conv = run("converged", BatchedOCPSolver(10, 4, 0.1, max_iter=100), 0, False, False, None)
print(json.dumps(dict(mode="converged", tick_ms_p50=float(np.percentile(conv["ms"], 50)), tick_ms_p99=float(np.percentile(conv["ms"], 99)),
                      applied=float(conv["ok"].mean()), limit_violations=int((np.abs(conv["Q"]) > qlim + 1e-9).sum()))), flush=True)
want = sys.argv[2:]
for hess, ex in (("gn", False), ("exact", True)):
    for cap in (4, 3, 5):
        for tol in (None, 1e-4, 1e-3, 1e-2, 1e-1, 1e9):
            name = f"{hess}-cap{cap}-tol{tol}"
            if want and not any(w in name for w in want):
                continue
            slv = BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=cap, mu_warm=3e-2, exact_hessian=ex)
            r = run(name, slv, 0, True, True, tol)
            slv.close()
            dev = r["Q"] - conv["Q"]
            per = np.sqrt(np.mean(dev ** 2, axis=(0, 2)))
            v = r["viol"].ravel()
            print(json.dumps(dict(mode=name, tick_ms_p50=round(float(np.percentile(r["ms"], 50)), 3), tick_ms_p99=round(float(np.percentile(r["ms"], 99)), 3),
                                  tick_ms_max=round(float(r["ms"].max()), 3), applied=round(float(r["ok"].mean()), 4), valid_min=round(float(r["valid"].min()), 4),
                                  mean_iters=round(float(r["its"].mean()), 2),
                                  g_viol_p50=float(np.percentile(v, 50)), g_viol_p90=float(np.percentile(v, 90)), g_viol_p99=float(np.percentile(v, 99)),
                                  median_stream_rms=float(np.median(per)), within_1e2=round(float((per <= 1e-2).mean()), 3), max_joint_dev=round(float(np.abs(dev).max()), 3),
                                  limit_violations=int((np.abs(r["Q"]) > qlim + 1e-9).sum()), phi_dev_max=round(float(np.abs(r["phi"] - conv["phi"]).max()), 3))), flush=True)
