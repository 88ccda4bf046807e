#!/usr/bin/env python3
"""BASELINE.json configs[4]: warm-started receding-horizon stream, hipGraph-captured step, batch=256 parallel trajectories.

Not the driver's bench line (that is bench.py / configs[1]); this script measures the per-tick latency of the
solver on a stream.  Workload: 256 trajectories replaying the recorded experiment1 closed loop (tests/golden/
g7_closedloop_exp1.npz: p and x0 of 155 consecutive ticks, x0 = shifted previous solution as BoundMPC.py:372-375),
trajectory b started at tick (b mod 100), so a batch mixes all phases of the motion.  Host-side packing is not part of
the timed region (device-side packing is a later row); the timed region per tick is the replay of the
captured graph {work-queue reset, solver kernel}, measured with HIP events on the launch stream.

Modes: cold (dual cold start every tick = what the reference does with Ipopt), warm (dual state carried, solved to tol),
rti-K (K Newton steps per tick).  For rti-K the deviation from the converged solution of the same tick is reported."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--ticks", type=int, default=50)
    ap.add_argument("--tol", type=float, default=1e-8)
    args = ap.parse_args()
    import torch
    from boundmpc_amd import BatchedOCPSolver
    d = np.load(os.path.join(ROOT, "tests", "golden", "g7_closedloop_exp1.npz"))
    B, T = args.batch, args.ticks
    off = np.arange(B) % 100
    assert off.max() + T <= d["p"].shape[0]
    dev = torch.device("cuda:0")
    Pall = torch.tensor(d["p"], device=dev); Xall = torch.tensor(d["x0"], device=dev); Xfin = torch.tensor(d["x"], device=dev)
    idx = torch.tensor(off, device=dev)
    solver = BatchedOCPSolver(10, 4, 0.1, tol=args.tol)
    solver.set_timing(True)
    p = torch.empty((B, 505), dtype=torch.float64, device=dev); x0 = torch.empty((B, 440), dtype=torch.float64, device=dev)
    res = []
    for mode, cap, warm in (("cold", 0, False), ("warm", 0, True), ("rti-3", 3, True), ("rti-2", 2, True), ("rti-1", 1, True)):
        state = solver.new_state(B) if (warm or cap) else None
        graph = solver.capture_step(p, x0, state=state, max_iter=cap)
        ms, its, err = [], [], []
        for t in range(T):
            p.copy_(Pall[idx + t]); x0.copy_(Xall[idx + t])
            if state is not None:
                if not warm: state.zero_()
                elif t: solver.shift_state(state)
            torch.cuda.synchronize()
            out = graph.launch()
            ms.append(solver.last_kernel_ms())
            its.append(float(out["iters"].double().mean().item()))
            dq = (out["x"] - Xfin[idx + t]).view(B, 10, 44)[:, :, 8:15]
            err.append(float(torch.sqrt((dq ** 2).mean()).item()))
        graph.close()
        ms = np.array(ms[1:]); its = np.array(its[1:]); err = np.array(err[1:])      # tick 0 is the cold start of the stream
        res.append({"mode": mode, "tick_ms_p50": float(np.percentile(ms, 50)), "tick_ms_p99": float(np.percentile(ms, 99)),
                    "ticks_per_s": float(1e3 / ms.mean()), "solves_per_s": float(B * 1e3 / ms.mean()), "mean_iters": float(its.mean()),
                    "rms_joint_dev_vs_converged_rad": float(np.sqrt(np.mean(err ** 2)))})
    print(json.dumps({"metric": "per-tick solver latency, warm-started stream (BASELINE configs[4])", "batch": B, "ticks": T - 1, "tol": args.tol,
                      "workload": "256 replayed experiment1 closed-loop streams, staggered start, hipGraph-captured step", "results": res}))


if __name__ == "__main__":
    main()
