"""Diagnostic (GPU box): the drop-in behind the reference's closed loop in its REAL-TIME configuration -- NodeLoop (the reference node's reset + step) over
NlpSolverShim(BatchedOCPSolver(tol=1e-3, fixed_barrier=LEVEL, max_iter=K)): K Newton steps per call on one barrier level, the shim stateless (cold duals
every call, x0 as given), the reference's own acceptance rule in BoundMPC.step() (solver success OR summed violation of g below 1e-4, BoundMPC.py:460-465)
deciding -- against the same loop over the default handle (every call solved to 1e-8).  Both experiments of the reference, until the goal.
RESULT (round 5): this does NOT work -- the stateless call with the 1e-4 rule loses the plan (experiment 1 at tick 74 on the level 0.1, at 45 on 0.01) or
crawls (experiment 2 on 0.1); the fixed-level ticks are a mode of the STREAM API (dual state and rejected iterates carried, first tick solved out,
acceptance at 1e-2): tests/gpu_fixed_level_experiments.py.
Usage: python tests/gpu_realtime_dropin.py [LEVEL=0.1] [K=8]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boundmpc_amd import BatchedOCPSolver, NlpSolverShim, workload
from boundmpc_amd.node_loop import NodeLoop
LEVEL = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
G = os.path.join(ROOT, "tests", "golden")


def run(which, solver, label):
    d6 = np.load(os.path.join(G, f"g6_pack_exp{which}_tick0.npz")); d7 = np.load(os.path.join(G, f"g7_closedloop_exp{which}.npz"))
    mk = lambda k: [np.array(v) for v in d6[k]]
    pub = []
    loop = NodeLoop(mk("p_via"), mk("r_via"), [mk("p_lower"), mk("p_upper")], [mk("r_lower"), mk("r_upper")], mk("bp1_in"), mk("br1_in"), list(d6["s_in"]),
                    list(d6["e_p_min_in"]), list(d6["e_r_min_in"]), list(d6["e_p_max_in"]), list(d6["e_r_max_in"]), p0=d6["p0fk"].copy(), q0=d7["q"][0],
                    params=workload.Params(weights=d6["weights_f64"], real_time=True, build=True), solver=solver, publish=pub.append)
    raw = loop.mpc.solver
    lat, fails, ticks = [], 0, 0
    orig = raw.__call__
    for t in range(400):
        t0 = time.perf_counter(); out = loop.step(); lat.append((time.perf_counter() - t0) * 1e3)
        if out is None:
            print(f"  {label}: the loop lost its plan at tick {t}"); break
        ticks += 1; fails += int(loop.mpc.error_count > 0)
        if loop.mpc.phi_max[0] - loop.mpc.phi_current[0] <= 0.01:
            break
    its = [p["iterations"] for p in pub]; tc = np.array([p["t_comp"] for p in pub]) * 1e3
    print(f"  experiment {which}, {label}: goal reached after {ticks} ticks (fixture, solved to 1e-8: {len(d7['q']) - 1}); ticks that replayed the previous plan {fails}; "
          f"iterations per call mean {np.mean(its):.1f} max {max(its)}; solver call (t_comp of the node) p50 {np.percentile(tc, 50):.2f} / p99 {np.percentile(tc, 99):.2f} ms; "
          f"whole step() incl. the host-side packing p50 {np.percentile(lat, 50):.2f} ms; final phi {loop.mpc.phi_current[0]:.3f} of {loop.mpc.phi_max[0]:.3f}")


for which in (1, 2):
    run(which, None, "default handle (1e-8)")
    s = BatchedOCPSolver(10, 4, 0.1, tol=1e-3, fixed_barrier=LEVEL, max_iter=K)
    run(which, NlpSolverShim(s), f"barrier level {LEVEL:g}, {K} Newton steps per call")
    s.close()
