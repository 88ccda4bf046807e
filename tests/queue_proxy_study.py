"""Diagnostic (CPU): how well cheap quantities of the cold start predict the iteration count of a long-horizon solve, and what a work queue ordered by
them would gain (list scheduling on 1024 waves, time = iterations).  Inputs: gpurun_out/queue_proxy_iters.npz (tests/gpu_queue_proxy_data.py)."""
import os, sys, heapq
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from boundmpc_amd import workload
from oracle import c_oracle
D = np.load(os.path.join(ROOT, "gpurun_out", "queue_proxy_iters.npz"))
def makespan(it, order, W=1024):
    h = [0.0] * W; heapq.heapify(h)
    for b in order:
        t = heapq.heappop(h); heapq.heappush(h, t + it[b] + 1.0)      # +1: prologue / epilogue of a solve
    return max(h)
def spearman(a, b):
    ra = np.argsort(np.argsort(a)); rb = np.argsort(np.argsort(b)); return np.corrcoef(ra, rb)[0, 1]
for name, N, tight, seed, B in (("c3", 30, True, 2, 8192), ("n30s7", 30, True, 7, 4096), ("n20s26", 20, True, 26, 4096), ("n30loose", 30, False, 9, 4096)):
    P, X, _ = workload.make_batch(B, seed=seed, N=N, tight=tight)
    it = D[name + "_iters"].astype(float)
    f = np.zeros(B); feats = {}
    gm = np.zeros((B, N, 43))
    for b in range(B):
        fb, gb = c_oracle.eval_fg(P[b], X[b], N, 4, 0.1)
        f[b] = fb; gm[b] = np.asarray(gb).reshape(N, 43)
    eq = np.abs(gm[:, :, :36]); iq = gm[:, :, 36:]
    feats["f(x0)"] = f
    feats["max ineq row"] = iq.max(axis=(1, 2))
    feats["sum pos ineq rows"] = np.maximum(iq, 0).sum(axis=(1, 2))
    feats["sum pos tube rows (last 5)"] = np.maximum(iq[:, :, 2:], 0).sum(axis=(1, 2))
    feats["eq 1-norm"] = eq.sum(axis=(1, 2))
    feats["f + 1e3 viol"] = f + 1e3 * np.maximum(iq, 0).sum(axis=(1, 2))
    print(f"== {name}: B={B} N={N} iterations mean {it.mean():.1f} max {it.max():.0f}; balanced bound {it.sum() / 1024:.0f}, slowest {it.max():.0f}")
    print(f"   natural order makespan {makespan(it, range(B)):.0f}; oracle (true iterations, longest first) {makespan(it, np.argsort(-it)):.0f}")
    for k, v in feats.items():
        print(f"   {k:28s} spearman {spearman(v, it):+.3f}  makespan longest-expected-first {makespan(it, np.argsort(-v)):.0f}   (ascending: {makespan(it, np.argsort(v)):.0f})")
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"queue_proxy_feats_{name}.npz"), f=f, g=gm.astype(np.float32), it=it)
