"""Diagnostic (GPU box): per-stream in-kernel solve time inside real-time ticks (256 streams, warm duals, tol 1e-3, cap K): where does the
tick time of the slowest stream come from?  Usage: python tests/gpu_stream_latency.py [K]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, workload, stream as bstream

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B, T = 256, 40
q0s = workload.random_q0(B, seed=3)
mpcs, recs = [], []
for q0 in q0s:
    m, p0fk = workload.make_mpc(q0)
    mpcs.append(m)
    recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
slv = BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=K, mu_warm=3e-2)
lat = torch.zeros(B, dtype=torch.float64, device="cuda")
slv.set_latency_buffer(lat)
sb = bstream.StreamBatch(slv, mpcs); sb.set_robot(np.stack(recs))
rows = []
for t in range(T):
    if t == 0:
        sb.tick(max_iter=100, warm_dual=True, simulate=True)
        continue
    sb.tick_graph(max_iter=0, warm_dual=True, simulate=True, accept_capped=True)
    torch.cuda.synchronize()
    rows.append(np.stack([lat.cpu().numpy(), sb.iters.cpu().numpy().astype(float)], 1))
a = np.concatenate(rows)
print(f"cap {K}: {len(a)} solves; in-kernel time per solve: median {np.median(a[:,0]):.0f} us, p90 {np.percentile(a[:,0],90):.0f}, p99 {np.percentile(a[:,0],99):.0f}, max {a[:,0].max():.0f}")
for k in range(1, K + 1):
    m = a[:, 1] == k
    if m.sum():
        v = a[m, 0]
        print(f"  {k} iterations: {m.sum():6d} solves, median {np.median(v):.0f} us ({np.median(v)/k:.0f} per iteration incl. fixed part), p90 {np.percentile(v,90):.0f}, p99 {np.percentile(v,99):.0f}, max {v.max():.0f}")
per_tick_max = np.array([r[:, 0].max() for r in rows])
print(f"  slowest stream of a tick: median {np.median(per_tick_max):.0f} us, max {per_tick_max.max():.0f} us")
