#!/usr/bin/env python3
"""INTEGRATION.md section B, executable (build container only: needs /root/reference, has no GPU).

The REFERENCE's own `BoundMPC` object (bound_mpc/BoundMPC/BoundMPC.py, unmodified) is constructed with the one-line change a
maintainer would make -- `setup_optimization_problem` hands back the product's `NlpSolverShim` instead of `ca.nlpsol(...)` -- and
its own step() / compute_return_data() drive a closed loop for 20 ticks the way bound_mpc_node.py:292-372 does.  The bound vectors
the shim is called with are the ones the reference's unmodified builder produces (ref_nlp.RefNlp); the shim refuses any other.

Without a GPU the shim's backend here is the CPU oracle wrapped in the `BatchedOCPSolver` interface (bounds / solve_host); on a
GPU box the same two lines take `boundmpc_amd.BatchedOCPSolver` (tests/test_gpu_parity.py::test_bound_mpc_step_closed_loop_drop_in
runs that loop on the device with the host mirror).  Prints the deviation from the committed closed-loop fixture G7."""
import os
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, OUT); sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402


class OracleBackend:
    """`BatchedOCPSolver` interface (bounds, solve_host) on the CPU oracle -- stands in for the GPU handle in this container only."""

    def __init__(self, N, S, dt):
        from oracle import nlp
        self.N, self.S, self.dt = N, S, dt
        self._b = nlp.bounds(N)

    def bounds(self):
        return self._b

    def solve_host(self, p, x0):
        from oracle import c_oracle
        return c_oracle.solve(p, x0, self.N, self.S, self.dt)


def main(ticks=20):
    mg._install_standins()
    from scipy.spatial.transform import Rotation as R
    from bound_mpc.RobotModel import RobotModel
    from bound_mpc.utils import get_default_path, get_default_weights, integrate_joint
    import bound_mpc.BoundMPC.BoundMPC as B
    from ref_nlp import RefNlp
    from boundmpc_amd.solver import NlpSolverShim
    ref = RefNlp(10, 4, 0.1)          # lbu, ubu, lbg, ubg exactly as the reference's builder returns them

    def setup_with_shim(N, nr_joints, nr_segs, dt, *a, **k):       # <- the maintainer's one-line change
        return NlpSolverShim(OracleBackend(N, nr_segs, dt)), ref.lbx.tolist(), ref.ubx.tolist(), ref.lbg.tolist(), ref.ubg.tolist(), ref.g_names
    B.setup_optimization_problem = setup_with_shim
    rm = RobotModel()
    for which in (1, 2):
        s = mg.experiment_setup(which, RobotModel, get_default_path, R)
        params = mg._Params(n=10, dt=0.1, weights=list(get_default_weights()), nr_segs=4, real_time=False)      # logging on
        cp = mg._cp
        mpc = B.BoundMPC(cp(s["p_via"]), cp(s["r_via"]), [cp(s["p_limits"][0]), cp(s["p_limits"][1])], [cp(s["r_limits"][0]), cp(s["r_limits"][1])],
                         cp(s["bp1"]), cp(s["br1"]), cp(s["s"]), cp(s["e_p_min"]), cp(s["e_r_min"]), cp(s["e_p_max"]), cp(s["e_r_max"]),
                         p0=np.copy(s["p0fk"]), params=params)
        d7 = np.load(os.path.join(OUT, f"g7_closedloop_exp{which}.npz"))
        q = s["q0"].copy(); dq = np.zeros(7); ddq = np.zeros(7); jerk = np.zeros(7); v = np.zeros(6)
        x_phi_d = np.array([mpc.phi_max[0], 0, 0])
        dev = 0.0
        for t in range(ticks):
            p_lie, _, _ = rm.forward_kinematics(q, dq)
            traj, ref_data, err_data, t_mpc, iters = mpc.step(q, dq, ddq, p_lie, v, x_phi_d, jerk)
            assert mpc.error_count == 0 and ref_data is not None and len(err_data["e_p"]) == 10
            dev = max(dev, np.abs(traj["q"] - d7["traj_q"][t]).max())
            jm = np.concatenate((jerk[:, None], traj["dddq"][:, :2]), axis=1)
            q, dq, ddq, p_lie, v = integrate_joint(rm, jm, q, dq, ddq, mpc.dt)[:5]
            jerk = traj["dddq"][:, 0].copy()
        print(f"experiment{which}: {ticks} ticks of the reference's BoundMPC.step() over NlpSolverShim: phi = {mpc.phi_current[0]:.6f} "
              f"(fixture {d7['phi_current'][ticks - 1]:.6f}), max |q - fixture| = {dev:.2e} rad, last iter_count {iters}")
        assert dev < 1e-6


if __name__ == "__main__":
    main()
