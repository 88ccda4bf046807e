#!/usr/bin/env python3
"""Generate golden vectors from the reference's OWN numeric code (run in the build
container only; /root/reference does not exist on the GPU box).

What this does
--------------
The reference (Thieso/BoundMPC) is pure Python.  Its numeric leaves (Maple-generated
iiwa14 kinematics, the hat-function jerk integrator, the quartic tube coefficients,
ReferencePath, the SO(3) helpers and the whole pre-solve half of ``BoundMPC.step``)
run under plain numpy.  They only need *names* from three absent third-party
modules, which this script provides in a temporary directory, never in the repo:

* ``casadi``      -> sin/cos/acos/sqrt/exp/dot/sumsqr/vertcat/if_else/norm_2 mapped to
                     numpy, plus empty ``SX/MX/DM`` classes used by ``isinstance`` tests.
                     No CasADi symbolic functionality is emulated; anything touching
                     ``ca.SX.sym`` / ``ca.nlpsol`` is NOT run.
* ``sensor_msgs`` / ``bound_mpc_msg`` -> empty message classes (imported by
                     ``utils/util_functions.py:3-5`` but never used by the leaves).

The Ipopt solve itself cannot be executed here, so ``BoundMPC.setup_optimization_problem``
is replaced by a stub that records the (x0, p) the reference's ``step()`` hands to the
solver (fixture G6) and, when given a solution vector, lets the reference's own
``compute_return_data`` run on it (fixture G7).

Outputs: ``tests/golden/*.npz`` (data only: inputs and expected outputs).
Usage:   python tests/golden/make_golden.py [--g7-from FILE.npz]
"""
import os
import sys
import tempfile
import textwrap
import copy
import argparse

import numpy as np

REF = "/root/reference/bound_mpc"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_standins():
    """Temp-dir stand-ins for the absent casadi / ROS message modules: see ref_nlp.install_standins and numeric_sx.py."""
    sys.path.insert(0, OUT)
    import ref_nlp
    return ref_nlp.install_standins()


class _Params:
    def __init__(self, n=10, dt=0.1, weights=None, nr_segs=4, real_time=True):
        self.n = n
        self.dt = dt
        self.weights = weights
        self.build = True
        self.real_time = real_time
        self.nr_segs = nr_segs


class _Captured(Exception):
    pass


class _StubSolver:
    """Stands where ca.nlpsol(...) would be (BoundMPC.py:150,446-456)."""

    def __init__(self):
        self.calls = []
        self.answer = None  # callable(x0, p) -> x or None

    def generate_dependencies(self, *a, **k):
        pass

    def __call__(self, x0=None, lbx=None, ubx=None, lbg=None, ubg=None, p=None):
        x0 = np.array(x0, dtype=float).ravel()
        p = np.array(p, dtype=float).ravel()
        self.calls.append((x0.copy(), p.copy()))
        if self.answer is None:
            raise _Captured()
        ans = self.answer(x0, p)
        ng = len(lbg)
        g = np.zeros((ng, 1))
        self._ok, self._it = True, 0
        if isinstance(ans, tuple):
            ans, g, self._ok, self._it = ans
            g = np.asarray(g, dtype=float).reshape(-1, 1)
        x = np.asarray(ans, dtype=float).ravel()
        return {"x": x.reshape(-1, 1), "g": g, "f": 0.0,
                "lam_x": np.zeros_like(x), "lam_g": np.zeros(ng)}

    def stats(self):
        return {"iter_count": getattr(self, "_it", 0), "success": getattr(self, "_ok", True),
                "return_status": "stub"}


def experiment_setup(which, RobotModel, get_default_path, R):
    """Constants of nodes/experiment{1,2}_runner.py (values only)."""
    q0 = np.zeros(7)
    if which == 1:
        q0[1] = np.pi / 3.5
        q0[3] = -np.pi / 3.5
        q0[5] = -12.85714286 * np.pi / 180
    else:
        q0[3] = -np.pi / 1.8
        q0[5] = np.pi / 2 - np.pi / 1.8
    rm = RobotModel()
    p0fk, _, _ = rm.forward_kinematics(q0, np.zeros(7))
    p0 = p0fk[:3]
    r0 = R.from_rotvec(p0fk[3:])
    (p_via, r_via, p_limits, r_limits, bp1_list, br1_list, s, e_p_min, e_r_min,
     e_p_max, e_r_max) = get_default_path(p0, r0, 5)
    if which == 1:
        p_via = [p0, p0 + np.array([-p0[0] * 2, 0.0, 0.0]),
                 p0 + np.array([-p0[0], p0[0], 0.0]),
                 p0 + np.array([-p0[0], -p0[0], 0.0]), p0]
        r1 = R.from_euler('XYZ', [0, 0, -np.pi]) * r0
        r2 = R.from_euler('XYZ', [0, 0, -np.pi / 2]) * r1
        r3 = R.from_euler('XYZ', [0, np.pi / 2, 0]) * R.from_euler('XYZ', [np.pi / 1.001, 0, 0]) * r2
        r4 = r0
        e_p_max = [0.5 for _ in e_p_max]
        br1_list[0] = np.array([0, 1.0, 0])
        br1_list[1] = np.array([0, 1.0, 0])
    else:
        r1 = R.from_euler('XYZ', [np.pi / 2, 0, 0]) * r0
        r2 = R.from_euler('XYZ', [0, 0, -np.pi / 3]) * r1
        r3 = (R.from_euler('XYZ', [0, 0, np.pi / 2.01]) * R.from_euler('XYZ', [np.pi / 2, 0, 0])
              * R.from_euler('XYZ', [0, 0, -np.pi / 2]) * r1)
        r4 = (R.from_euler('XYZ', [0, 0, np.pi / 2]) * R.from_euler('XYZ', [np.pi / 2, 0, 0])
              * R.from_euler('XYZ', [0, 0, -np.pi / 2]) * r1)
        p_via = [p0, p0 + np.array([-0.2, -0.0, 0.1]), p0 + np.array([-0.6, -0.6, 0.1]),
                 p0 + np.array([-0.8, -0.5, -0.2]), p0 + np.array([-0.8, -0.5, -0.5])]
        p_lower = [np.array(v) for v in ([-1.0, -1.0], [-0.01, -1.0], [-1.0, -1.0], [-0.1, -0.1], [-0.1, -0.1])]
        p_upper = [np.array(v) for v in ([1.0, 1.0], [0.01, 1.0], [1.0, 1.0], [0.1, 0.1], [0.1, 0.1])]
        p_limits = [p_lower, p_upper]
        r_lower = [np.array(v) for v in ([-1.0, -1.0], [-0.11, -0.11], [-1.0, -1.0], [-0.1, -0.1], [-0.1, -0.1])]
        r_upper = [np.array(v) for v in ([1.0, 1.0], [0.11, 0.11], [1.0, 1.0], [0.1, 0.1], [0.1, 0.1])]
        r_limits = [r_lower, r_upper]
        bp1_list = [np.array(v) for v in ([0., 0., 1.], [0., 0., 1.], [0., 0., 1.], [0., 1., 0.], [0., 1., 0.])]
        br1_list = [np.array(v) for v in ([0., 0., 1.], [0., 1., 0.], [0., 0., 1.], [0., 1., 0.], [0., 1., 0.])]
    r_via = [r.as_matrix() for r in (r0, r1, r2, r3, r4)]
    return dict(q0=q0, p0fk=p0fk, p_via=p_via, r_via=r_via, p_limits=p_limits, r_limits=r_limits,
                bp1=bp1_list, br1=br1_list, s=s, e_p_min=e_p_min, e_r_min=e_r_min,
                e_p_max=e_p_max, e_r_max=e_r_max)


def _cp(x):
    return copy.deepcopy(x)


def UNDEF_MASK(S=4):
    """True where the reference defines p; False at row S of a4..a0 (np.empty, BoundMPC.py:235-240)."""
    mask = np.ones(141 + 91 * S, dtype=bool)
    base0 = 42 + 9 * S + 3 + 7 + 1 + (S + 1) + 18 + 12 * S + 3 * S * 5
    for b in range(5):
        for ch in range(9):
            mask[base0 + b * 9 * (S + 1) + ch * (S + 1) + S] = False
    return mask


def make_mpc(BoundMPCmod, setup, weights, n=10, dt=0.1, nr_segs=4):
    """Mirror of MPCNode.reset (bound_mpc_node.py:48-83): the node hands the limits over as
    [upper, lower] after create_traj_msg swapped them once, so ReferencePath sees
    p_limit[0] = runner's lower list."""
    stub = _StubSolver()

    def fake_setup(N, nr_joints, nr_segs_, dt_, *a, **k):
        lbg = ([0.0] * 36 + [-np.inf] * 7) * N      # casadi_ocp_formulation.py:272-349 (restated)
        return stub, [0.0] * (44 * N), [0.0] * (44 * N), lbg, [0.0] * (43 * N), []
    BoundMPCmod.setup_optimization_problem = fake_setup
    params = _Params(n=n, dt=dt, weights=list(weights), nr_segs=nr_segs)
    s = setup
    mpc = BoundMPCmod.BoundMPC(_cp(s["p_via"]), _cp(s["r_via"]),
                               [_cp(s["p_limits"][0]), _cp(s["p_limits"][1])],
                               [_cp(s["r_limits"][0]), _cp(s["r_limits"][1])],
                               _cp(s["bp1"]), _cp(s["br1"]), _cp(s["s"]), _cp(s["e_p_min"]),
                               _cp(s["e_r_min"]), _cp(s["e_p_max"]), _cp(s["e_r_max"]),
                               p0=np.copy(s["p0fk"]), params=params)
    return mpc, stub


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--closed-loop", type=int, default=0, metavar="TICKS",
                    help="drive the reference's own host code (step(), compute_return_data, integrate_joint) "
                         "for TICKS ticks with the CPU oracle solver (oracle/bmpc_oracle.c) standing where "
                         "Ipopt would be, and record G6(ticks)/G7")
    args = ap.parse_args()
    _install_standins()
    from scipy.spatial.transform import Rotation as R
    from bound_mpc.RobotModel import RobotModel
    from bound_mpc.ReferencePath import ReferencePath
    from bound_mpc.utils import (get_default_path, get_default_weights, compute_initial_rot_errors,
                                 integrate_rotation_reference, jac_SO3_inv_left, jac_SO3_inv_right,
                                 integrate_joint)
    from bound_mpc.BoundMPC.jerk_trajectory_casadi import calcAngle, calcVelocity, calcAcceleration
    from bound_mpc.BoundMPC.mpc_utils_casadi import (compute_bound_params,
                                                     compute_fourth_order_error_bound_general,
                                                     compute_position_error, decompose_orthogonal_error,
                                                     integrate_rot_error_diff)
    import bound_mpc.BoundMPC.BoundMPC as B

    rng = np.random.default_rng(20250314)
    rm = RobotModel()
    qlim = np.array(rm.q_lim_upper)
    dqlim = np.array(rm.dq_lim_upper)

    # ---- G1 kinematics -------------------------------------------------------------
    n = 256
    q = rng.uniform(-1, 1, (n, 7)) * qlim
    dq = rng.uniform(-1, 1, (n, 7)) * dqlim
    ddq = rng.uniform(-3, 3, (n, 7))
    g1 = dict(q=q, dq=dq, ddq=ddq,
              fk_pos=np.array([rm.fk_pos(a) for a in q]),
              velocity_ee=np.array([rm.velocity_ee(a, b) for a, b in zip(q, dq)]),
              omega_ee=np.array([rm.omega_ee(a, b) for a, b in zip(q, dq)]),
              jacobian_fk=np.array([rm.jacobian_fk(a) for a in q]),
              djacobian_fk=np.array([rm.djacobian_fk(a, b) for a, b in zip(q, dq)]),
              ddjacobian_fk=np.array([rm.ddjacobian_fk(a, b, c) for a, b, c in zip(q, dq, ddq)]),
              hom=np.array([rm.hom_transform_endeffector(a) for a in q]),
              fk=np.array([rm.fk(a) for a in q]),
              limits=np.array([rm.q_lim_lower, rm.q_lim_upper, rm.dq_lim_lower, rm.dq_lim_upper]),
              u_lim=np.array([rm.u_min, rm.u_max], dtype=float))
    np.savez_compressed(os.path.join(OUT, "g1_kinematics.npz"), **g1)

    # ---- G2 jerk integrator ----------------------------------------------------------
    h = 0.1
    m = 64
    jm2 = rng.uniform(-35, 35, (m, 7, 2))
    q0s = rng.uniform(-1, 1, (m, 7))
    dq0s = rng.uniform(-1, 1, (m, 7))
    ddq0s = rng.uniform(-2, 2, (m, 7))
    g2 = dict(h=h, jm2=jm2, q0=q0s, dq0=dq0s, ddq0=ddq0s,
              ang2=np.array([calcAngle(jm2[i], h, q0s[i], dq0s[i], ddq0s[i], h) for i in range(m)]),
              vel2=np.array([calcVelocity(jm2[i], h, dq0s[i], ddq0s[i], h) for i in range(m)]),
              acc2=np.array([calcAcceleration(jm2[i], h, ddq0s[i], h) for i in range(m)]))
    jm11 = rng.uniform(-35, 35, (8, 7, 11))
    ang = np.zeros((8, 10, 7)); vel = np.zeros((8, 10, 7)); acc = np.zeros((8, 10, 7))
    for i in range(8):
        for k in range(10):
            t = h * (k + 1)
            ang[i, k] = calcAngle(jm11[i], t, q0s[i], dq0s[i], ddq0s[i], h)
            vel[i, k] = calcVelocity(jm11[i], t, dq0s[i], ddq0s[i], h)
            acc[i, k] = calcAcceleration(jm11[i], t, ddq0s[i], h)
    g2.update(jm11=jm11, ang11=ang, vel11=vel, acc11=acc)
    # three-column matrix evaluated at t = h, as the node does (bound_mpc_node.py:320-323)
    jm3 = rng.uniform(-35, 35, (16, 7, 3))
    g2.update(jm3=jm3,
              ang3=np.array([calcAngle(jm3[i], h, q0s[i], dq0s[i], ddq0s[i], h) for i in range(16)]),
              vel3=np.array([calcVelocity(jm3[i], h, dq0s[i], ddq0s[i], h) for i in range(16)]),
              acc3=np.array([calcAcceleration(jm3[i], h, ddq0s[i], h) for i in range(16)]))
    np.savez_compressed(os.path.join(OUT, "g2_integrator.npz"), **g2)

    # ---- G3 tubes ------------------------------------------------------------------------
    m = 32
    phi1 = rng.uniform(0.2, 3.0, m)
    e0 = rng.uniform(-0.3, 0.3, (m, 9)); e1 = rng.uniform(-0.3, 0.3, (m, 9))
    sv = rng.uniform(-0.5, 0.5, (m, 9)); emax = rng.uniform(-1, 1, (m, 9))
    coef = np.array([np.array(compute_bound_params(0, phi1[i], e0[i], e1[i], sv[i], emax[i])) for i in range(m)])
    ph = rng.uniform(0, 1, (m, 5)) * phi1[:, None]
    val = np.array([[compute_fourth_order_error_bound_general(ph[i, j], *coef[i]) for j in range(5)]
                    for i in range(m)])
    np.savez_compressed(os.path.join(OUT, "g3_tubes.npz"), phi1=phi1, e0=e0, e1=e1, s=sv, emax=emax,
                        coef_a4_a3_a2_a1_a0=coef, phi_eval=ph, bound_eval=val)

    # ---- G5 Lie helpers / leaves -----------------------------------------------------------
    m = 64
    ax = rng.normal(size=(m, 3)); ax *= (rng.uniform(0.01, 3.0, (m, 1)) / np.linalg.norm(ax, axis=1, keepdims=True))
    ax[0] = 0.0
    ax[1] = [1e-9, 0, 0]
    pr = rng.normal(size=(m, 3)); pr *= (rng.uniform(0.0, 3.1, (m, 1)) / np.linalg.norm(pr, axis=1, keepdims=True))
    prr = rng.normal(size=(m, 3)); prr *= (rng.uniform(0.0, 3.1, (m, 1)) / np.linalg.norm(prr, axis=1, keepdims=True))
    dpr = rng.normal(size=(m, 3)); dpr[2] = 0.0
    ire = []
    b1s = []; b2s = []
    for i in range(m):
        nrm = np.linalg.norm(dpr[i])
        om = dpr[i] / nrm if nrm > 1e-4 else np.array([0, 1.0, 0])
        b = rng.normal(size=3); b -= (om @ b) * om; b /= np.linalg.norm(b)
        b1s.append(b); b2s.append(np.cross(om, b))
        ire.append(np.array(compute_initial_rot_errors(pr[i], prr[i], dpr[i], b1s[-1], b2s[-1])))
    om = rng.normal(size=(m, 3)); om[3] = 0
    ph0 = rng.uniform(0, 2, m); ph1 = ph0 + rng.uniform(0, 1, m)
    e = rng.normal(size=(m, 3)); dd = rng.normal(size=(m, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    pe = [compute_position_error(e[i] + 1.0, e[i] * 0.3, np.ones(3), dd[i], 0 * dd[i], 0.7) for i in range(m)]
    np.savez_compressed(
        os.path.join(OUT, "g5_lie.npz"), axis=ax,
        jac_right=np.array([jac_SO3_inv_right(a) for a in ax]),
        jac_left=np.array([jac_SO3_inv_left(a) for a in ax]),
        pr=pr, pr_ref=prr, dp_ref=dpr, br1=np.array(b1s), br2=np.array(b2s),
        init_rot_errors=np.array(ire),  # [m][4: dtau_init, par, orth1, orth2][3]
        omega=om, phi0=ph0, phi1=ph1,
        rot_ref_int=np.array([integrate_rotation_reference(pr[i], om[i], ph0[i], ph1[i]) for i in range(m)]),
        pe_p=e + 1.0, pe_v=e * 0.3, pe_d=dd,
        pe_out=np.array([[np.asarray(x, dtype=float) for x in t] for t in pe]))

    # ---- G4 ReferencePath, G6 packing ----------------------------------------------------------
    w64 = get_default_weights()
    w32 = w64.astype(np.float32).astype(np.float64)
    dt32 = float(np.float32(0.1))
    for which in (1, 2):
        setup = experiment_setup(which, RobotModel, get_default_path, R)
        # G4: reference path window at several phi, walking forward (update() is stateful)
        s = setup
        rp = ReferencePath(_cp(s["p_via"]), _cp(s["r_via"]), [_cp(s["p_limits"][0]), _cp(s["p_limits"][1])],
                           [_cp(s["r_limits"][0]), _cp(s["r_limits"][1])], _cp(s["bp1"]), _cp(s["br1"]),
                           _cp(s["s"]), _cp(s["e_p_min"]), _cp(s["e_r_min"]), _cp(s["e_p_max"]),
                           _cp(s["e_r_max"]), 4)
        phis = np.linspace(0.0, rp.phi_max, 25)
        rec = {k: [] for k in ("pd", "dpd_normed", "dpd", "phi_switch", "asymm_lower", "asymm_upper",
                               "bp1", "bp2", "br1", "br2", "sector", "e_p_min", "e_r_min", "e_p_max",
                               "e_r_max", "s")}
        for ph_ in phis:
            pd, dn, dpd, _, psw = rp.get_parameters(ph_)
            al, au, b1, b2, r1_, r2_ = rp.get_limits()
            epm, erm, epx, erx, ss = rp.get_bound_params()
            for k, v in zip(rec.keys(), (pd, dn, dpd, psw, al, au, b1, b2, r1_, r2_, rp.sector,
                                         epm, erm, epx, erx, ss)):
                rec[k].append(np.array(v, dtype=float).copy())
        inputs = dict(q0=s["q0"], p0fk=s["p0fk"], p_via=np.array(s["p_via"]), r_via=np.array(s["r_via"]),
                      p_lower=np.array(s["p_limits"][0]), p_upper=np.array(s["p_limits"][1]),
                      r_lower=np.array(s["r_limits"][0]), r_upper=np.array(s["r_limits"][1]),
                      bp1_in=np.array(s["bp1"]), br1_in=np.array(s["br1"]), s_in=np.array(s["s"]),
                      e_p_min_in=np.array(s["e_p_min"]), e_r_min_in=np.array(s["e_r_min"]),
                      e_p_max_in=np.array(s["e_p_max"]), e_r_max_in=np.array(s["e_r_max"]))
        np.savez_compressed(os.path.join(OUT, f"g4_refpath_exp{which}.npz"), phis=phis, phi_max=rp.phi_max,
                            **{k: np.array(v) for k, v in rec.items()}, **inputs)

        # G6: (x0, p) at tick 0 for exact-double and float32-rounded (node) parameters
        out = dict(inputs)
        for tag, wts, dt in (("f64", w64, 0.1), ("f32", w32, dt32)):
            mpc, stub = make_mpc(B, setup, wts, dt=dt)
            x_phi_d = np.array([mpc.phi_max[0], 0, 0])
            try:
                mpc.step(s["q0"], np.zeros(7), np.zeros(7), np.copy(s["p0fk"]), np.zeros(6), x_phi_d, np.zeros(7))
            except _Captured:
                pass
            x0, p = stub.calls[-1]
            # rows S of a4..a0 are np.empty garbage in the reference (BoundMPC.py:235-240): zero them in
            # the fixture and record the mask of defined entries
            p = p.copy()
            p[~UNDEF_MASK()] = 0.0
            out[f"x0_{tag}"] = x0
            out[f"p_{tag}"] = p
            out[f"weights_{tag}"] = wts
            out[f"dt_{tag}"] = dt
            out[f"phi_max_{tag}"] = mpc.phi_max[0]
        out["p_defined_mask"] = UNDEF_MASK()
        np.savez_compressed(os.path.join(OUT, f"g6_pack_exp{which}_tick0.npz"), **out)

    # ---- G6 (ticks) / G7: drive the reference's host code with the build's solutions ---------------
    if args.closed_loop:
        sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
        from oracle import c_oracle
        opts = c_oracle.default_opts()
        for which in (1, 2):
            setup = experiment_setup(which, RobotModel, get_default_path, R)
            mpc, stub = make_mpc(B, setup, w64, dt=0.1)
            stats = []

            def answer(x0, p):
                p = p.copy(); p[~UNDEF_MASK()] = 0.0
                out = c_oracle.solve(p, x0, 10, 4, 0.1, opts, 1)
                stats.append((int(out["iters"][0]), int(out["status"][0]), float(out["kkt"][0])))
                return out["x"][0], out["g"][0], int(out["status"][0]) == 0, int(out["iters"][0])
            stub.answer = answer
            q = setup["q0"].copy(); dq_ = np.zeros(7); ddq_ = np.zeros(7); jerk = np.zeros(7)
            v = np.zeros(6); p_lie = setup["p0fk"].copy()
            x_phi_d = np.array([mpc.phi_max[0], 0, 0])
            rec = {k: [] for k in ("x0", "p", "x", "q", "dq", "ddq", "jerk", "p_lie", "v",
                                   "traj_p", "traj_v", "traj_a", "traj_q", "traj_dq", "traj_ddq", "traj_dddq",
                                   "traj_phi", "traj_dphi", "traj_ddphi", "traj_dddphi",
                                   "phi_current", "dphi_current", "ddphi_current", "dddphi_current",
                                   "pr_ref", "iw_ref", "sector", "iters", "status", "kkt", "error_count")}
            for i in range(args.closed_loop):
                if mpc.phi_max[0] - mpc.phi_current[0] <= 0.01:   # experiment1_runner.py:109
                    break
                p_lie, jac, _ = rm.forward_kinematics(q, dq_)
                st = dict(q=q.copy(), dq=dq_.copy(), ddq=ddq_.copy(), jerk=jerk.copy(), p_lie=p_lie.copy(), v=v.copy())
                traj, _, _, _, _ = mpc.step(q, dq_, ddq_, p_lie, v, x_phi_d, jerk)
                x0, p = stub.calls[-1]
                for k_, v_ in st.items():
                    rec[k_].append(v_)
                p = p.copy(); p[~UNDEF_MASK()] = 0.0
                rec["x0"].append(x0); rec["p"].append(p); rec["x"].append(np.array(mpc.prev_solution, dtype=float).copy())
                rec["iters"].append(stats[-1][0]); rec["status"].append(stats[-1][1]); rec["kkt"].append(stats[-1][2])
                rec["error_count"].append(mpc.error_count)
                for k_ in ("p", "v", "a", "q", "dq", "ddq", "dddq", "phi", "dphi", "ddphi", "dddphi"):
                    rec["traj_" + k_].append(np.array(traj[k_], dtype=float).copy())
                rec["phi_current"].append(mpc.phi_current[0]); rec["dphi_current"].append(mpc.dphi_current[0])
                rec["ddphi_current"].append(mpc.ddphi_current[0]); rec["dddphi_current"].append(mpc.dddphi_current[0])
                rec["pr_ref"].append(np.array(mpc.pr_ref, dtype=float).copy())
                rec["iw_ref"].append(np.array(mpc.iw_ref, dtype=float).copy())
                rec["sector"].append(mpc.ref_path.sector)
                jm = np.concatenate((jerk[:, None], traj["dddq"][:, :2]), axis=1)
                ns = integrate_joint(rm, jm, q, dq_, ddq_, mpc.dt)
                q, dq_, ddq_, p_lie, v = ns[0], ns[1], ns[2], ns[3], ns[4]
                jerk = traj["dddq"][:, 0].copy()
            np.savez_compressed(os.path.join(OUT, f"g7_closedloop_exp{which}.npz"),
                                **{k_: np.array(v_) for k_, v_ in rec.items()})
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
