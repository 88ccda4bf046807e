"""Diagnostic (GPU box): which stream / tick of the 64-stream closed loops moves its path parameter backwards, and what the solver reported there."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, stream as bstream, workload
B, T = 64, 130
q0s = workload.random_q0(256, seed=3)[:B]
mpcs, recs = [], []
for q0 in q0s:
    m, p0fk = workload.make_mpc(q0); mpcs.append(m)
    recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
for resto in (True, False):
    slv = BatchedOCPSolver(10, 4, 0.1, max_iter=100, restoration=resto)
    sb = bstream.StreamBatch(slv, mpcs); sb.set_robot(np.stack(recs))
    side = torch.cuda.Stream(); rows = []
    with torch.cuda.stream(side):
        for t in range(T):
            if t == 0: sb.tick(max_iter=100, warm_dual=True, simulate=True)
            else: sb.tick_graph(warm_dual=True, simulate=True)
            side.synchronize()
            rows.append((sb.state[:, bstream.SS["PHI"]].cpu().numpy().copy(), sb.state[:, bstream.SS["ERRCNT"]].cpu().numpy().copy(), sb.status.cpu().numpy().copy(),
                         sb.iters.cpu().numpy().copy(), (sb.traj[:, -2] > 0.5).cpu().numpy().copy(), sb.traj[:, -1].cpu().numpy().copy()))
    phis = np.array([r[0] for r in rows]); d = np.diff(phis, axis=0)
    print(f"restoration {resto}: min dphi per tick {d.min():.4f}; streams with a plan at the end {(rows[-1][1] < 10).mean():.3f}; applied {np.mean([r[4].mean() for r in rows]):.3f}")
    for (t, b) in zip(*np.where(d < -0.02)):
        print(f"   tick {t + 1} stream {b}: phi {phis[t, b]:.4f} -> {phis[t + 1, b]:.4f}; status {rows[t + 1][2][b]} iters {rows[t + 1][3][b]} applied {rows[t + 1][4][b]} g_viol {rows[t + 1][5][b]:.2e} "
              f"errcnt before {rows[t][1][b]:.0f} after {rows[t + 1][1][b]:.0f}; previous tick: status {rows[t][2][b]} iters {rows[t][3][b]}")
    sb.close(); slv.close()
