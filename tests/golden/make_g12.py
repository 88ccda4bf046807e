#!/usr/bin/env python3
"""Fixture G12: packing and post-processing for other horizons / window sizes than the experiments' (N=10, S=4).

The reference's own BoundMPC (stub solver, as for G6/G7) is constructed with (n, nr_segs) = (5, 2), (8, 3), (20, 4), (6, 5), (12, 6) on the
experiment-2 path (asymmetric tubes, mixed bases) and driven in closed loop for a few ticks, the CPU oracle standing where Ipopt would:
recorded per tick: the arguments of step(), (x0, p) handed to the solver, the solution used, traj_data and the advanced state.
Build container only:  python tests/golden/make_g12.py"""
import os
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT); sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
import make_golden as mg  # noqa: E402


def main():
    mg._install_standins()
    from scipy.spatial.transform import Rotation as R
    from bound_mpc.RobotModel import RobotModel
    from bound_mpc.utils import get_default_path, get_default_weights, integrate_joint
    import bound_mpc.BoundMPC.BoundMPC as B
    from oracle import c_oracle
    rm = RobotModel()
    w = get_default_weights()
    out = {}
    for (N, S, dt, ticks) in ((5, 2, 0.1, 12), (8, 3, 0.08, 16), (20, 4, 0.1, 8), (6, 5, 0.1, 10), (12, 6, 0.1, 6)):
        setup = mg.experiment_setup(2, RobotModel, get_default_path, R)
        mpc, stub = mg.make_mpc(B, setup, w, n=N, dt=dt, nr_segs=S)
        mask = mg.UNDEF_MASK(S)

        def answer(x0, p):
            pp = p.copy(); pp[~mask] = 0.0
            o = c_oracle.solve(pp, x0, N, S, dt)
            return o["x"][0], o["g"][0], int(o["status"][0]) == 0, int(o["iters"][0])
        stub.answer = answer
        q = setup["q0"].copy(); dq = np.zeros(7); ddq = np.zeros(7); jerk = np.zeros(7); v = np.zeros(6)
        x_phi_d = np.array([mpc.phi_max[0], 0, 0])
        rec = {k: [] for k in ("q", "dq", "ddq", "jerk", "p_lie", "v", "x0", "p", "x", "traj_q", "traj_p", "traj_a", "traj_phi", "phi_current",
                               "pr_ref", "iw_ref", "sector", "error_count")}
        for t in range(ticks):
            p_lie, _, _ = rm.forward_kinematics(q, dq)
            st = dict(q=q.copy(), dq=dq.copy(), ddq=ddq.copy(), jerk=jerk.copy(), p_lie=p_lie.copy(), v=v.copy())
            traj, _, _, _, _ = mpc.step(q, dq, ddq, p_lie, v, x_phi_d, jerk)
            x0, p = stub.calls[-1]
            p = p.copy(); p[~mask] = 0.0
            for k, v_ in st.items():
                rec[k].append(v_)
            rec["x0"].append(x0); rec["p"].append(p); rec["x"].append(np.array(mpc.prev_solution, dtype=float).copy())
            for k in ("q", "p", "a", "phi"):
                rec["traj_" + k].append(np.array(traj[k], dtype=float).copy())
            rec["phi_current"].append(mpc.phi_current[0]); rec["pr_ref"].append(np.array(mpc.pr_ref, dtype=float).copy())
            rec["iw_ref"].append(np.array(mpc.iw_ref, dtype=float).copy()); rec["sector"].append(mpc.ref_path.sector)
            rec["error_count"].append(mpc.error_count)
            jm = np.concatenate((jerk[:, None], traj["dddq"][:, :2]), axis=1)
            ns = integrate_joint(rm, jm, q, dq, ddq, mpc.dt)
            q, dq, ddq, p_lie, v = ns[0], ns[1], ns[2], ns[3], ns[4]
            jerk = traj["dddq"][:, 0].copy()
        key = f"n{N}s{S}_"
        out.update({key + k: np.array(v_) for k, v_ in rec.items()})
        out[key + "dt"] = dt; out[key + "mask"] = mask; out[key + "phi_max"] = mpc.phi_max[0]
        print(f"N={N} S={S}: {ticks} ticks, phi {rec['phi_current'][-1]:.4f}, error counts {sorted(set(rec['error_count']))}, n_p {len(p)}")
    out["weights"] = w
    np.savez_compressed(os.path.join(OUT, "g12_pack_other_sizes.npz"), **out)
    print("written", os.path.getsize(os.path.join(OUT, "g12_pack_other_sizes.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
