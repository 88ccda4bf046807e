#!/usr/bin/env python3
"""Fixture G10: the logging dictionaries ref_data / err_data of the reference's compute_return_data (BoundMPC.py:614-752).

The reference's own BoundMPC object (log on: params.real_time = False) is driven over the recorded closed loops of fixture G7 --
its step() packs, the stub solver answers with the recorded solution of that tick, its compute_return_data runs the numeric
branches of reference_function / error_function (bound_mpc_functions.py:43-202) -- and the two dictionaries of selected ticks are
stored, together with one tick at which the stub reports a failure (error_count = 1: shortened plan, BoundMPC.py:465-489).
The numeric branches lean on CasADi's DM conventions for what `ca.if_else` returns; numeric_sx.DM provides them (see its header).
Build container only:  python tests/golden/make_g10.py"""
import os
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import make_golden as mg  # noqa: E402

REF_KEYS = ("p", "dp", "ddp", "dp_normed", "r_par_bound", "bound_lower", "bound_upper", "e_p_off", "e_r_off", "bp1", "bp2", "br1", "br2",
            "v1", "v2", "v3")
ERR_KEYS = ("e_p", "de_p", "e_p_par", "e_p_orth", "de_p_par", "de_p_orth", "e_r", "de_r", "e_r_par", "e_r_orth1", "e_r_orth2")


def main():
    mg._install_standins()
    from scipy.spatial.transform import Rotation as R
    from bound_mpc.RobotModel import RobotModel
    from bound_mpc.utils import get_default_path, get_default_weights, integrate_joint
    import bound_mpc.BoundMPC.BoundMPC as B
    rm = RobotModel()
    out = {}
    for which in (1, 2):
        d7 = np.load(os.path.join(OUT, f"g7_closedloop_exp{which}.npz"))
        T = d7["x"].shape[0]
        setup = mg.experiment_setup(which, RobotModel, get_default_path, R)
        mpc, stub = mg.make_mpc(B, setup, get_default_weights(), dt=0.1)
        mpc.log = True
        fail_tick = 6 if which == 1 else 20
        tick = [0]

        def answer(x0, p):
            t = tick[0]
            np.testing.assert_allclose(p[mg.UNDEF_MASK()], d7["p"][t][mg.UNDEF_MASK()], atol=1e-9)   # the loop really retraces G7
            if t == fail_tick:
                return d7["x"][t], np.ones(430), False, 1
            return d7["x"][t], np.zeros(430), True, 1
        stub.answer = answer
        sect = d7["sector"]
        sw_ticks = [int(t) for t in np.nonzero(np.diff(sect))[0]]
        keep = sorted(set([0, 1, fail_tick, T // 2, T - 2] + sw_ticks + [t + 1 for t in sw_ticks] + [max(t - 3, 0) for t in sw_ticks]))
        q = setup["q0"].copy(); dq = np.zeros(7); ddq = np.zeros(7); jerk = np.zeros(7); v = np.zeros(6)
        x_phi_d = np.array([mpc.phi_max[0], 0, 0])
        rec = {}
        for t in range(T):
            tick[0] = t
            if t == fail_tick + 1:
                break          # after the forced failure the loop leaves the recorded one: stop there
            p_lie, _, _ = rm.forward_kinematics(q, dq)
            traj, ref, err, _, _ = mpc.step(q, dq, ddq, p_lie, v, x_phi_d, jerk)
            if t in keep:
                n = len(ref["p"])
                for k in REF_KEYS:
                    rec[f"t{t}_ref_{k}"] = np.array([np.asarray(ref[k][i], dtype=float).ravel() for i in range(n)])
                for k in ERR_KEYS:
                    rec[f"t{t}_err_{k}"] = np.array([np.asarray(err[k][i], dtype=float).ravel() for i in range(n)])
                rec[f"t{t}_error_count"] = mpc.error_count
            jm = np.concatenate((jerk[:, None], traj["dddq"][:, :2]), axis=1)
            ns = integrate_joint(rm, jm, q, dq, ddq, mpc.dt)
            q, dq, ddq, p_lie, v = ns[0], ns[1], ns[2], ns[3], ns[4]
            jerk = traj["dddq"][:, 0].copy()
        # second pass without the failure for the later ticks
        mpc, stub = mg.make_mpc(B, setup, get_default_weights(), dt=0.1)
        mpc.log = True
        stub.answer = lambda x0, p: (d7["x"][tick[0]], np.zeros(430), True, 1)
        q = setup["q0"].copy(); dq = np.zeros(7); ddq = np.zeros(7); jerk = np.zeros(7); v = np.zeros(6)
        for t in range(T):
            tick[0] = t
            p_lie, _, _ = rm.forward_kinematics(q, dq)
            traj, ref, err, _, _ = mpc.step(q, dq, ddq, p_lie, v, x_phi_d, jerk)
            if t in keep and t > fail_tick:
                n = len(ref["p"])
                for k in REF_KEYS:
                    rec[f"t{t}_ref_{k}"] = np.array([np.asarray(ref[k][i], dtype=float).ravel() for i in range(n)])
                for k in ERR_KEYS:
                    rec[f"t{t}_err_{k}"] = np.array([np.asarray(err[k][i], dtype=float).ravel() for i in range(n)])
                rec[f"t{t}_error_count"] = mpc.error_count
            jm = np.concatenate((jerk[:, None], traj["dddq"][:, :2]), axis=1)
            ns = integrate_joint(rm, jm, q, dq, ddq, mpc.dt)
            q, dq, ddq, p_lie, v = ns[0], ns[1], ns[2], ns[3], ns[4]
            jerk = traj["dddq"][:, 0].copy()
        ticks = sorted(int(k.split("_")[0][1:]) for k in rec if k.endswith("_error_count"))
        out.update({f"exp{which}_{k}": v_ for k, v_ in rec.items()})
        out[f"exp{which}_ticks"] = np.array(ticks)
        out[f"exp{which}_fail_tick"] = fail_tick
        print(f"exp{which}: ticks {ticks} (failure forced at {fail_tick})")
    np.savez_compressed(os.path.join(OUT, "g10_logging.npz"), **out)
    print("written", os.path.join(OUT, "g10_logging.npz"), os.path.getsize(os.path.join(OUT, "g10_logging.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
