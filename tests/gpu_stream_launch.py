"""Diagnostic (GPU box): tick latency of 256 real-time streams (warm duals, tol 1e-3, cap K) launched as ONE hipGraph against the same
four operations enqueued directly on the stream; HIP events around the launch and host clock around launch + synchronize.
Usage: python tests/gpu_stream_launch.py [K]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, workload, stream as bstream

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B, T = 256, 60
q0s = workload.random_q0(B, seed=3)
mpcs, recs = [], []
for q0 in q0s:
    m, p0fk = workload.make_mpc(q0)
    mpcs.append(m)
    recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
for how in ("graph", "direct"):
    slv = BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=K, mu_warm=3e-2)
    sb = bstream.StreamBatch(slv, mpcs); sb.set_robot(np.stack(recs))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dev_ms, host_ms = [], []
    for t in range(T):
        if t == 0:
            sb.tick(max_iter=100, warm_dual=True, simulate=True); torch.cuda.synchronize(); continue
        t0 = time.perf_counter()
        ev0.record()
        if how == "graph":
            sb.tick_graph(max_iter=0, warm_dual=True, simulate=True, accept_capped=True)
        else:
            sb.tick(max_iter=0, warm_dual=True, simulate=True, accept_capped=True)
        ev1.record(); ev1.synchronize()
        host_ms.append((time.perf_counter() - t0) * 1e3); dev_ms.append(ev0.elapsed_time(ev1))
    d, h = np.array(dev_ms[3:]), np.array(host_ms[3:])
    print(f"cap {K} {how:6s}: events p50 {np.median(d):.3f} p99 {np.percentile(d,99):.3f} ms | host clock p50 {np.median(h):.3f} p99 {np.percentile(h,99):.3f} ms | mean iters {sb.iters.double().mean().item():.2f}")
    sb.close(); slv.close()
