#!/usr/bin/env python3
"""Fixture G11: re-planning.  The reference's own BoundMPC is driven over the first ticks of the recorded experiment-1 loop (G7),
then `update()` (BoundMPC.py:163-217) switches it to a new path that starts at the current pose -- called the way the node's
trajectory callback calls it (bound_mpc_node.py:121-165) -- and three more ticks follow, whose warm start goes through the
re-projection branch of step() (BoundMPC.py:335-369; `self.updated` is never cleared).  Recorded: the arguments of update(), the
state it leaves, and (x0, p) + the advanced state of every later tick (solutions after the update come from the CPU oracle).
Build container only:  python tests/golden/make_g11.py"""
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
    d7 = np.load(os.path.join(OUT, "g7_closedloop_exp1.npz"))
    setup = mg.experiment_setup(1, RobotModel, get_default_path, R)
    w = get_default_weights()
    mpc, stub = mg.make_mpc(B, setup, w, dt=0.1)
    tick = [0]
    post = []

    def answer(x0, p):
        if tick[0] < T_UPD:
            return d7["x"][tick[0]], np.zeros(430), True, 1
        pp = p.copy(); pp[~mg.UNDEF_MASK()] = 0.0
        o = c_oracle.solve(pp, x0, 10, 4, 0.1)
        post.append((x0.copy(), pp, o["x"][0].copy(), int(o["status"][0])))
        return o["x"][0], o["g"][0], int(o["status"][0]) == 0, int(o["iters"][0])
    stub.answer = answer
    T_UPD = 12
    q = setup["q0"].copy(); dq = np.zeros(7); ddq = np.zeros(7); jerk = np.zeros(7); v = np.zeros(6)
    x_phi_d = np.array([mpc.phi_max[0], 0, 0])
    out = {}
    rec = {k: [] for k in ("q", "dq", "ddq", "jerk", "p_lie", "v", "phi_current", "dphi_current", "ddphi_current", "dddphi_current",
                           "pr_ref", "iw_ref", "sector", "traj_q", "traj_phi")}
    for t in range(T_UPD + 3):
        tick[0] = t
        p_lie, jac, djac = rm.forward_kinematics(q, dq)
        if t == T_UPD:
            # the node's callback (bound_mpc_node.py:121-165): first via point slightly ahead of the current pose, new via points
            # of the experiment-2 shape relative to it, measured Cartesian state handed over
            p0n = p_lie[:3] + 0.5 * mpc.dt * v[:3]
            r0 = R.from_rotvec(p_lie[3:])
            s2 = mg.experiment_setup(2, RobotModel, get_default_path, R)
            shift = p0n - s2["p_via"][0]
            p_via = [np.asarray(pv) + shift for pv in s2["p_via"]]
            r_via = [r0.as_matrix()] + [np.asarray(rv) for rv in s2["r_via"][1:]]
            a_cart = jac @ ddq + djac @ dq
            j_cart = jac @ jerk + 2 * djac @ ddq          # plausible measured jerk (value only matters as an input)
            args = dict(p_via=np.array(p_via), r_via=np.array(r_via), p_lower=np.array(s2["p_limits"][0]), p_upper=np.array(s2["p_limits"][1]),
                        r_lower=np.array(s2["r_limits"][0]), r_upper=np.array(s2["r_limits"][1]), bp1=np.array(s2["bp1"]), br1=np.array(s2["br1"]),
                        s=np.array(s2["s"]), e_p_min=np.array(s2["e_p_min"]), e_r_min=np.array(s2["e_r_min"]), e_p_max=np.array(s2["e_p_max"]),
                        e_r_max=np.array(s2["e_r_max"]), p=p_lie.copy(), v=v.copy(), a=a_cart, jerk=j_cart)
            cp = mg._cp
            mpc.update(cp(p_via), cp(r_via), [cp(s2["p_limits"][0]), cp(s2["p_limits"][1])], [cp(s2["r_limits"][0]), cp(s2["r_limits"][1])],
                       cp(s2["bp1"]), cp(s2["br1"]), cp(s2["s"]), cp(s2["e_p_min"]), cp(s2["e_r_min"]), cp(s2["e_p_max"]), cp(s2["e_r_max"]),
                       p_lie.copy(), v.copy(), a_cart.copy(), j_cart.copy(), p0=p_lie.copy(), params=mg._Params(weights=list(w)))
            x_phi_d = np.array([mpc.phi_max[0], 0, 0])
            out.update({"upd_" + k: v_ for k, v_ in args.items()})
            out.update(after_update_phi=np.array([mpc.phi_current[0], mpc.dphi_current[0], mpc.ddphi_current[0], mpc.dddphi_current[0]]),
                       after_update_pr_ref=np.array(mpc.pr_ref, dtype=float), after_update_iw_ref=np.array(mpc.iw_ref, dtype=float),
                       after_update_phi_max=mpc.phi_max[0])
        st = dict(q=q.copy(), dq=dq.copy(), ddq=ddq.copy(), jerk=jerk.copy(), p_lie=p_lie.copy(), v=v.copy())
        traj, _, _, _, _ = mpc.step(q, dq, ddq, p_lie, v, x_phi_d, jerk)
        if t >= T_UPD:
            for k, v_ in st.items():
                rec[k].append(v_)
            rec["phi_current"].append(mpc.phi_current[0]); rec["dphi_current"].append(mpc.dphi_current[0])
            rec["ddphi_current"].append(mpc.ddphi_current[0]); rec["dddphi_current"].append(mpc.dddphi_current[0])
            rec["pr_ref"].append(np.array(mpc.pr_ref, dtype=float)); rec["iw_ref"].append(np.array(mpc.iw_ref, dtype=float))
            rec["sector"].append(mpc.ref_path.sector)
            rec["traj_q"].append(np.array(traj["q"])); rec["traj_phi"].append(np.array(traj["phi"]))
        jm = np.concatenate((jerk[:, None], traj["dddq"][:, :2]), axis=1)
        ns = integrate_joint(rm, jm, q, dq, ddq, mpc.dt)
        q, dq, ddq, p_lie, v = ns[0], ns[1], ns[2], ns[3], ns[4]
        jerk = traj["dddq"][:, 0].copy()
    out.update({k: np.array(v_) for k, v_ in rec.items()})
    out.update(t_update=T_UPD, x0=np.array([a[0] for a in post]), p=np.array([a[1] for a in post]), x=np.array([a[2] for a in post]),
               status=np.array([a[3] for a in post]), weights=w)
    np.savez_compressed(os.path.join(OUT, "g11_update.npz"), **out)
    print("post-update statuses", out["status"], "phi after update", out["after_update_phi"], "phi per tick", out["phi_current"])


if __name__ == "__main__":
    main()
