"""Diagnostic (GPU box): the fixed barrier level on the REFERENCE's own two experiments through the stream API (duals and rejected iterates carried, first tick
solved out, K Newton steps per tick): which level still reaches the goal (phi_max - phi <= 0.01)?  Usage: python tests/gpu_fixed_level_experiments.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_stream as T
from boundmpc_amd import BatchedOCPSolver, stream as bstream
ms = T._mpcs()
def run(label, K, feas, ticks=300, **kw):
    slv = BatchedOCPSolver(10, 4, 0.1, **kw); slv.set_rt_feasibility_tol(feas)
    sb = bstream.StreamBatch(slv, [m for m, _ in ms]); sb.set_robot(np.stack([T._robot0(m, d) for m, d in ms]))
    done=[None,None]; app=[]; best=[9.0,9.0]; lv=[]
    for t in range(ticks):
        if t == 0: sb.tick(max_iter=100, warm_dual=True, simulate=True)
        else:
            if K: sb.tick_graph(max_iter=K, warm_dual=True, simulate=True, accept_capped=True)
            else: sb.tick_graph(warm_dual=True, simulate=True)
        torch.cuda.synchronize()
        st = sb.state.cpu().numpy(); app.append((sb.traj[:, -2] > 0.5).cpu().numpy().copy())
        lv.append(sb.dual[:, 570].cpu().numpy().copy())
        for b in range(2):
            best[b] = min(best[b], float(ms[b][0].phi_max[0] - st[b, bstream.SS["PHI"]]))
            if done[b] is None and ms[b][0].phi_max[0] - st[b, bstream.SS["PHI"]] <= 0.01: done[b]=t+1
        if all(d is not None for d in done): break
    st = sb.state.cpu().numpy()
    print(label, "ticks to the goal (exp1, exp2):", done, "applied fraction", np.mean(app,axis=0).round(3).tolist(), "phi", st[:, bstream.SS["PHI"]].round(3).tolist(), "of", [float(m.phi_max[0]) for m,_ in ms], "valid", st[:, bstream.SS["VALID"]].tolist(), "closest approach", [round(v, 4) for v in best], "level in the dual state at ticks 1, 50, 100, last", [np.round(lv[i], 4).tolist() for i in (1, min(50, len(lv) - 1), min(100, len(lv) - 1), len(lv) - 1)], flush=True)
    sb.close(); slv.close()
run("converged", 0, 1e-4)
for L in (0.1, 0.01):
    for K in (8,):
        run(f"level {L} K={K} rule 1e-2", K, 1e-2, tol=1e-3, max_iter=30, fixed_barrier=L)
run("level 0.1 K=8 rule 1e-4", 8, 1e-4, tol=1e-3, max_iter=30, fixed_barrier=0.1)
for c in (0.02, 0.05, 0.1):      # the level that sets itself: clamp(c (phi_max - phi), 0.01, 0.1) per stream, held inside the tick
    for K in (8,):
        run(f"level auto c={c} K={K} rule 1e-2", K, 1e-2, tol=1e-3, max_iter=30, fixed_barrier="auto", level_c=c)
