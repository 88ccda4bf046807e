"""Host object of the MPC: parameter packing, warm start, solver call, feasibility fallback and
post-integration -- the mirror of the reference's `BoundMPC`
(/root/reference/bound_mpc/bound_mpc/BoundMPC/BoundMPC.py): constructor :20-33, `update` :163-217,
`compute_error_bounds` :219-265, `compute_orientation_projection_vectors` :267-304, `step` :306-506,
`compute_return_data` :508-611,757-770, with the helpers it pulls from utils
(`compute_initial_rot_errors` util_functions.py:11-31, `integrate_rotation_reference` :88-99,
`jac_SO3_inv_right/left` lie_functions.py:41-64, `compute_bound_params` mpc_utils_casadi.py:130-137).

Same names, argument meaning and error behaviour, so `bound_mpc_node.py:304-305` can call
`mpc.step(q, dq, ddq, p_lie, v, x_phi_d, jerk)` unchanged.  The one difference is what sits behind
`self.solver`: the HIP batched solver (boundmpc_amd.solver.NlpSolverShim) instead of
CasADi/Ipopt.  `solve()` / `pack()` are additions (the reference has no `solve`).
The logging half of `compute_return_data` (BoundMPC.py:614-752: ref_data / err_data, consumed by the node's RViz / Logger
branch bound_mpc_node.py:306-360) is produced when `params.real_time` is false, as in the reference; the values are pinned by
fixture G10 (the reference's own numeric reference_function / error_function run on recorded ticks)."""
import copy
import time

import numpy as np
from scipy.spatial.transform import Rotation as R

from .reference_path import ReferencePath
from .robot_model import RobotModel

NZ = 44


# ------------------------------------------------------------------------------------------
# small leaves
# ------------------------------------------------------------------------------------------
def _skew(a):
    return np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])


def jac_SO3_inv_right(axis):
    """J_r^{-1} of SO(3) with the reference's angle = |axis| + 1e-6 (lie_functions.py:41-51)."""
    th = np.linalg.norm(axis) + 1e-6
    K = _skew(axis)
    return np.eye(3) + 0.5 * K + (1 / th ** 2 - (1 + np.cos(th)) / (2 * th * np.sin(th))) * (K @ K)


def jac_SO3_inv_left(axis):
    th = np.linalg.norm(axis) + 1e-6
    K = _skew(axis)
    return np.eye(3) - 0.5 * K + (1 / th ** 2 - (1 + np.cos(th)) / (2 * th * np.sin(th))) * (K @ K)


def compute_initial_rot_errors(pr, pr_ref, dp_ref, br1, br2):
    """Initial orientation error and its zyx-Euler split in the frame [br2, d, br1] (util_functions.py:11-31)."""
    dtau = R.from_matrix(R.from_rotvec(pr).as_matrix() @ R.from_rotvec(pr_ref).as_matrix().T).as_rotvec()
    n = np.linalg.norm(dp_ref)
    dn = dp_ref / n if n > 1e-4 else np.array([0.0, 1.0, 0.0])
    F = np.column_stack([br2, dn, br1])
    eul = R.from_matrix(F.T @ R.from_rotvec(dtau).as_matrix() @ F).as_euler('zyx')
    return dtau, eul[1] * dn, eul[0] * br1, eul[2] * br2     # dtau, par, orth1, orth2


def integrate_rotation_reference(pr_ref, omega, phi0, phi1):
    """Rotate the reference by the constant angular velocity omega over phi1-phi0 (util_functions.py:88-99)."""
    r0 = R.from_rotvec(pr_ref).as_matrix()
    n = np.linalg.norm(omega)
    if n > 1e-4:
        K = _skew(omega / n)
        ang = float(np.squeeze((phi1 - phi0) * n))
        r0 = (np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)) @ r0
    return R.from_matrix(r0).as_rotvec()


def compute_bound_params(phi1, e0, e1, s, e_max):
    """Quartic a4..a0 on [0, phi1] with f(0)=e0, f(phi1)=e1, f(phi1/2)=e_max, f'(0)=s, f'(phi1)=-s
    (closed form of mpc_utils_casadi.py:130-137 at phi0 = 0)."""
    L = phi1
    r1 = e1 - e0 - s * L
    r2 = e_max - e0 - s * L / 2
    r3 = -2 * s * L
    A = 16 * r2 - 5 * r1 + r3
    B = -32 * r2 + 14 * r1 - 3 * r3
    C = 16 * r2 - 8 * r1 + 2 * r3
    return C / L ** 4, B / L ** 3, A / L ** 2, s + 0 * e0, e0 + 0 * s


def integrate_chain(x, dx, ddx, u_prev, u, h):
    """One step of the hat-function jerk integrator (jerk_trajectory_casadi.py:78-175 closed form)."""
    return (x + h * dx + h * h / 2 * ddx + h ** 3 / 8 * u_prev + h ** 3 / 24 * u,
            dx + h * ddx + h * h / 3 * u_prev + h * h / 6 * u,
            ddx + h / 2 * (u_prev + u))


def integrate_joint(model, jerk_matrix, q, dq, ddq, dt):
    """util_functions.py:152-161 (the node's kinematic simulation step)."""
    qn, dqn, ddqn = integrate_chain(q, dq, ddq, jerk_matrix[:, 0], jerk_matrix[:, 1], dt)
    pn, J, dJ = model.forward_kinematics(qn, dqn)
    ddJ = model.ddjacobian_fk(q, dq, ddq)
    return qn, dqn, ddqn, pn, J @ dqn, dJ @ dqn + J @ ddqn, ddJ @ dqn + 2 * dJ @ ddqn + J @ ddqn


class BoundMPC:
    def __init__(self, pos_points, rot_points, pos_lim, rot_lim, bp1, br1, s, e_p_min, e_r_min, e_p_max, e_r_max,
                 p0=np.zeros(6), params=None, solver=None):
        self.N = params.n
        self.robot_model = RobotModel()
        self.updated = False
        self.updated_once = False
        self.build = params.build
        self.log = not params.real_time
        self.p0 = p0
        self.error_count = 0
        self.dt = params.dt
        self.T = self.dt * self.N
        self.nr_segs = params.nr_segs
        self.ref_path = ReferencePath(pos_points, rot_points, pos_lim, rot_lim, bp1, br1, s, e_p_min, e_r_min,
                                      e_p_max, e_r_max, self.nr_segs)
        S = self.nr_segs
        self.dtau_init = np.empty((3, S)); self.dtau_init_par = np.empty((3, S))
        self.dtau_init_orth1 = np.empty((3, S)); self.dtau_init_orth2 = np.empty((3, S))
        self.phi_max = np.array([self.ref_path.phi_max - 0.0001])
        self.weights = np.array(params.weights, dtype=float)
        self.dphi_max = np.array([self.weights[4]])
        self.pr_ref = np.array(p0[3:], dtype=float)
        self.iw_ref = np.zeros(3)
        (self.q_lim_upper, self.q_lim_lower, self.dq_lim_upper, self.dq_lim_lower, self.tau_lim_upper,
         self.tau_lim_lower, self.u_max, self.u_min) = self.robot_model.get_robot_limits()
        self.ut_max, self.ut_min = self.u_max, self.u_min
        self.phi_current = np.array([0.0]); self.phi_prev = np.array([0.0])
        self.dphi_current = np.array([0.0]); self.ddphi_current = np.array([0.0]); self.dddphi_current = np.array([0.0])
        self.nr_joints, self.nr_u, self.nr_x = 7, 8, NZ
        self.prev_solution = None
        self.prev_infeasible_solution = None
        self.lam_g0 = 0; self.lam_x0 = 0
        if solver is None:
            # the product path: HIP batched solver behind the nlpsol call convention; raises without a GPU
            from .solver import BatchedOCPSolver, NlpSolverShim
            self.batched = BatchedOCPSolver(self.N, S, self.dt, tol=getattr(params, "tol", 1e-8),
                                            max_iter=getattr(params, "max_iter", 500))
            solver = NlpSolverShim(self.batched)
            self.lbu, self.ubu, self.lbg, self.ubg = (a.tolist() for a in self.batched.bounds())
        else:
            self.batched = None
            self.lbu, self.ubu, self.lbg, self.ubg = structural_bounds(self.N)
        self.solver = solver
        if self.build:
            self.solver.generate_dependencies('gen_traj_opt_nlp_deps.cpp', {'cpp': True})

    # --------------------------------------------------------------------------------------
    def update(self, pos_points, rot_points, pos_lim, rot_lim, bp1, br1, s, e_p_min, e_r_min, e_p_max, e_r_max,
               p, v, a, jerk, p0=np.zeros(6), params=None):
        """Re-planning (BoundMPC.py:163-217)."""
        self.updated = True
        self.updated_once = True
        self.p0 = p0
        self.ref_path = ReferencePath(pos_points, rot_points, pos_lim, rot_lim, bp1, br1, s, e_p_min, e_r_min,
                                      e_p_max, e_r_max, self.nr_segs)
        self.phi_max = np.array([self.ref_path.phi_max - 0.0001])
        self.weights = np.array(params.weights, dtype=float)
        dp0 = self.ref_path.dp[0] / np.linalg.norm(self.ref_path.dp[0])
        self.phi_current = np.array([(p0[:3] - pos_points[0]).T @ dp0])
        self.phi_prev = self.phi_current
        dp_new = self.ref_path.dpd[:3, 0]
        self.dphi_current = np.array([v[:3].T @ dp_new])
        self.ddphi_current = np.array([a[:3].T @ dp_new])
        self.dddphi_current = np.array([jerk[:3].T @ dp_new])
        self.pr_ref = integrate_rotation_reference(R.from_matrix(rot_points[0]).as_rotvec(), self.ref_path.dr[0], 0.0,
                                                   self.phi_current)
        self.iw_ref = self.ref_path.pd[3:, 0] + self.phi_current * self.ref_path.dpd[3:, 0]

    def compute_error_bounds(self, asymm_upper, asymm_lower, phi_switch, s, e_p_min, e_r_min, e_p_max, e_r_max):
        """Quartic tube coefficients a0..a4, each (S+1) x 9 (BoundMPC.py:219-265).  Only element [0] of the bound
        parameters is used for every segment, as in the reference (:224-232).  Row S, which the reference leaves as
        np.empty garbage (:235-240), is defined as a copy of row S-1."""
        S = self.nr_segs
        ab = np.concatenate((asymm_upper[:2], -asymm_lower[:2], asymm_upper[2:], -asymm_lower[2:]))      # 8 x S
        pm, rmn, px, rx, sl = e_p_min[0], e_r_min[0], e_p_max[0], e_r_max[0], s[0]
        e0 = np.array([pm, pm, -pm, -pm, rmn, rmn, -rmn, -rmn, rmn])
        sign = np.array([1, 1, -1, -1, 1, 1, -1, -1, 1.0])
        emax0 = sign * np.array([px, px, px, px, rx, rx, rx, rx, rx])
        sv0 = sign * sl
        a = [np.zeros((S + 1, 9)) for _ in range(5)]
        for i in range(S):
            scale = np.concatenate((ab[:, i], ab[-1:, i]))
            c = compute_bound_params(phi_switch[i + 1] - phi_switch[i], e0, e0, sv0 * scale, emax0 * scale)
            for j in range(5):
                a[j][i] = c[j]
        for j in range(5):
            a[j][S] = a[j][S - 1]
        a4, a3, a2, a1, a0 = a
        return a0, a1, a2, a3, a4

    def compute_orientation_projection_vectors(self, br1, br2, dp_normed_ref):
        """Dual basis (v1,v2,v3) of the Jacobian-mapped directions (BoundMPC.py:267-304)."""
        S = dp_normed_ref.shape[1]
        jac_r = jac_SO3_inv_right(self.dtau_init[:, 0])
        jac_l = jac_SO3_inv_left(self.dtau_init[:, 0])
        v_1 = np.empty((3, S)); v_2 = np.empty((3, S)); v_3 = np.empty((3, S))
        Rinit = R.from_rotvec(self.dtau_init[:, 0]).as_matrix()
        for i in range(S):
            rest1 = Rinit @ R.from_rotvec(self.dtau_init_orth1[:, i]).as_matrix().T
            rest2 = rest1 @ R.from_rotvec(self.dtau_init_par[:, i]).as_matrix().T
            t1 = jac_r @ br1[:, i]
            t2 = jac_SO3_inv_right(R.from_matrix(rest1).as_rotvec()) @ dp_normed_ref[:, i]
            t3 = jac_SO3_inv_right(R.from_matrix(rest2).as_rotvec()) @ br2[:, i]
            # rows of the inverse of [t1 t2 t3] are the dual vectors
            Minv = np.linalg.inv(np.column_stack([t1, t2, t3]))
            v_1[:, i], v_2[:, i], v_3[:, i] = Minv[0], Minv[1], Minv[2]
        return v_1, v_2, v_3, jac_l, jac_r

    # --------------------------------------------------------------------------------------
    def pack(self, q0, dq0, ddq0, p0, v0, x_phi_d, jerk_current):
        """Steps 1-6 of the reference's step(): initial guess w0 and parameter vector (BoundMPC.py:310-443).
        Returns (w0 list, params ndarray, aux dict for the post-processing)."""
        N, S = self.N, self.nr_segs
        p_ref, dp_normed_ref, dp_ref, ddp_ref, phi_switch = self.ref_path.get_parameters(self.phi_current)
        asymm_lower, asymm_upper, bp1, bp2, br1, br2 = self.ref_path.get_limits()
        e_p_min, e_r_min, e_p_max, e_r_max, s = self.ref_path.get_bound_params()
        if self.prev_solution is None:          # cold start :316-321
            w0 = np.zeros((N, NZ))
            w0[:, 8:15] = q0
            w0[:, 29:35] = p0
        else:
            w0 = np.array(self.prev_solution, dtype=float).reshape(N, -1).copy()
            # The measured rotation vector may have flipped to its 2 pi-equivalent since the plan was made (scipy keeps |rotvec| <= pi):
            # a jump of more than 1.5 rad between it and the plan's first integrated-omega entry re-bases the planned increments on the
            # measurement (behaviour of BoundMPC.py:326-333; the last stage repeats its predecessor)
            iw = w0[:, 32:35]
            if np.linalg.norm(np.asarray(p0)[3:] - iw[0]) > 1.5:
                increments = iw[1:] - iw[0]
                iw[:-1] = np.asarray(p0)[3:] + increments
                iw[-1] = iw[-2]
            if self.updated:                                       # re-projection after a path update :335-369
                self._prev_cartesian_derivatives()
                dp_new, p_ref_new = dp_ref[:3, 0], p_ref[:3, 0]
                for i in range(N):
                    phik = phi_switch[0] + (self.prev_traj[:3, i] - p_ref_new) @ dp_new
                    if phik > phi_switch[1] - 0.01:
                        w0[i, 41], w0[i, 42], w0[i, 43] = phi_switch[1] - 0.01, 0.0, 0.0
                    elif phik < 0:
                        print("[WARNING] PHI TOO LOW")
                        w0[i, 8:15] = q0
                        w0[i, 41:44] = 0.0
                        w0[i, 29:35] = p0
                        w0[i, 35:39] = 0.0
                    else:
                        w0[i, 41] = phik
                        w0[i, 42] = self.prev_vel[:3, i] @ dp_new
                        w0[i, 43] = self.prev_acc[:3, i] @ dp_new
                        w0[i, 7] = self.prev_jerk[:3, i] @ dp_new
            else:                                                  # shift by one stage, last stage duplicated :372-375
                w0[:-1] = w0[1:].copy()
        w0 = w0.reshape(-1)
        for i in range(S):                                         # :379-385
            (self.dtau_init[:, i], self.dtau_init_par[:, i], self.dtau_init_orth1[:, i],
             self.dtau_init_orth2[:, i]) = compute_initial_rot_errors(p0[3:], self.pr_ref, dp_ref[3:, i], br1[:, i], br2[:, i])
        v_1, v_2, v_3, jac_l, jac_r = self.compute_orientation_projection_vectors(br1, br2, dp_normed_ref)
        a0, a1, a2, a3, a4 = self.compute_error_bounds(asymm_upper, asymm_lower, phi_switch, s, e_p_min, e_r_min, e_p_max, e_r_max)
        x_phi_d_current = np.array(x_phi_d, dtype=float).copy()
        weights_current = self.weights.copy()
        if x_phi_d[0] < 1:                                         # :400-403
            weights_current[6] *= min(1 / (self.phi_max[0] ** 2), 2.0)
        phi_max = np.array([min(self.phi_current[0] + 5.0, self.phi_max[0])])     # :407-408
        x_phi_d_current[0] = min(self.phi_current[0] + 5.0, x_phi_d_current[0])
        qd = np.array(q0, dtype=float) if phi_max[0] - self.phi_current[0] < 0.05 else np.zeros(7)   # :411-413
        params = np.concatenate((
            q0, dq0, ddq0, self.phi_current, self.dphi_current, self.ddphi_current, p0, v0,
            self.iw_ref, self.dtau_init[:, 0], self.dtau_init_par.T.ravel(), self.dtau_init_orth1.T.ravel(),
            self.dtau_init_orth2.T.ravel(), x_phi_d_current, jerk_current, self.dddphi_current, phi_switch,
            jac_r.T.ravel(), jac_l.T.ravel(), p_ref.ravel(), dp_ref.ravel(), dp_normed_ref.ravel(),
            bp1.ravel(), bp2.ravel(), br1.ravel(), br2.ravel(), a4.T.ravel(), a3.T.ravel(), a2.T.ravel(), a1.T.ravel(),
            a0.T.ravel(), weights_current, phi_max, self.dphi_max, v_1.ravel(), v_2.ravel(), v_3.ravel(), qd)).astype(float)
        aux = dict(p_ref=p_ref, dp_ref=dp_ref, phi_switch=phi_switch, dp_normed_ref=dp_normed_ref, bp1=bp1, bp2=bp2, br1=br1, br2=br2,
                   v1=v_1, v2=v_2, v3=v_3, jac_l=jac_l, jac_r=jac_r, a=(a4, a3, a2, a1, a0))
        return w0.tolist(), params, aux

    def solve(self, q0, dq0, ddq0, p0, v0, x_phi_d, jerk_current):
        """Pack + one solver call, no post-processing and no state advance of the path parameter
        (the warm-start bookkeeping of pack() still applies).  Returns (sol, stats)."""
        w0, params, _ = self.pack(q0, dq0, ddq0, p0, v0, x_phi_d, jerk_current)
        sol = self.solver(x0=w0, lbx=self.lbu, ubx=self.ubu, lbg=self.lbg, ubg=self.ubg, p=params)
        return sol, self.solver.stats()

    def step(self, q0, dq0, ddq0, p0, v0, x_phi_d, jerk_current, x_des=None):
        """One optimisation step: pack, solve, decide which plan to execute, post-process (behaviour of BoundMPC.py:306-506).

        Plan selection.  A solve counts when the solver reports success or when the summed violation of the constraint bounds
        (beyond 1e-6 per row) stays below 1e-4.  Otherwise the stored plan is executed once more and `error_count` says how many of
        its leading stages have been consumed; with no stored plan the infeasible iterate itself is used and the counter restarts.
        After N failures in a row there is nothing left to replay and five Nones go back to the caller."""
        w0, params, aux = self.pack(q0, dq0, ddq0, p0, v0, x_phi_d, jerk_current)
        t0 = time.perf_counter()
        sol = self.solver(x0=w0, lbx=self.lbu, ubx=self.ubu, lbg=self.lbg, ubg=self.ubg, p=params)
        candidate = np.asarray(sol['x'], dtype=float).ravel()
        time_elapsed = time.perf_counter() - t0
        stats = self.solver.stats()
        g = np.asarray(sol['g'], dtype=float).ravel()
        below = np.asarray(self.lbg, dtype=float) - 1e-6 - g
        above = g - np.asarray(self.ubg, dtype=float) - 1e-6
        violation = float(np.sum(g[above > 0]) - np.sum(g[below > 0]))
        accepted = bool(stats['success']) or violation < 1e-4
        replay = not accepted
        if accepted:
            self.error_count = 0
            plan = candidate
            self.prev_solution = candidate.copy()
            self.prev_infeasible_solution = candidate
            self.lam_g0, self.lam_x0 = sol['lam_g'], sol['lam_x']
        else:
            self.error_count += 1
            print(f"[boundmpc_amd] solve rejected ({stats['return_status']}; constraint violation sum {violation:.3e}); "
                  f"consecutive failures: {self.error_count}")
            if self.prev_solution is None:
                print("[boundmpc_amd] no stored plan to fall back on: executing the infeasible iterate")
                self.error_count = 0
                plan = candidate
                self.prev_infeasible_solution = None
                self.lam_g0, self.lam_x0 = sol['lam_g'], sol['lam_x']
            else:
                plan = np.array(self.prev_solution, dtype=float)
                self.prev_infeasible_solution = candidate
        if self.error_count >= self.N:
            return None, None, None, None, None
        traj_data, ref_data, err_data = self.compute_return_data(q0, dq0, ddq0, jerk_current, p0, plan, replay, aux)
        return traj_data, ref_data, err_data, time_elapsed, stats['iter_count']

    def _prev_cartesian_derivatives(self):
        """Cartesian acceleration / jerk of the previous plan (BoundMPC.py:560-566), evaluated on demand."""
        if self.prev_acc is not None:
            return
        w, rm, N = self._prev_w, self.robot_model, self.N
        self.prev_acc = np.empty((6, N)); self.prev_jerk = np.empty((6, N))
        for i in range(N):
            _, J, dJ = rm.forward_kinematics(w[8:15, i], w[15:22, i])
            ddJ = rm.ddjacobian_fk(w[8:15, i], w[15:22, i], w[22:29, i])
            self.prev_acc[:, i] = J @ w[22:29, i] + dJ @ w[15:22, i]
            self.prev_jerk[:, i] = J @ w[:7, i] + dJ @ w[22:29, i] + ddJ @ w[15:22, i]

    def compute_return_data(self, q0, dq0, ddq0, jerk_current, p0, w_opt, using_previous, aux):
        """Re-integration of the joint / path states from the optimal jerks, Cartesian trajectory, advance of the
        path-parameter state and of the rotation reference (BoundMPC.py:513-611,757-770)."""
        N, ec, dt = self.N, self.error_count, self.dt
        p_ref, dp_ref, phi_switch = aux["p_ref"], aux["dp_ref"], aux["phi_switch"]
        w = np.array(w_opt, dtype=float).reshape(N, -1).T.copy()          # 44 x N
        n = N - ec
        jerk, jerk_phi = w[:7, ec:], w[7, ec:]
        q, dq, ddq = np.array(q0, dtype=float), np.array(dq0, dtype=float), np.array(ddq0, dtype=float)
        ph, dph, ddph = self.phi_current[0], self.dphi_current[0], self.ddphi_current[0]
        u_prev, up_prev = np.array(jerk_current, dtype=float), self.dddphi_current[0]
        self.phi_prev = np.copy(self.phi_current)
        for i in range(n):                                                 # :536-555 (closed-form chaining)
            q, dq, ddq = integrate_chain(q, dq, ddq, u_prev, jerk[:, i], dt)
            ph, dph, ddph = integrate_chain(ph, dph, ddph, up_prev, jerk_phi[i], dt)
            u_prev, up_prev = jerk[:, i], jerk_phi[i]
            w[8:15, ec + i], w[15:22, ec + i], w[22:29, ec + i] = q, dq, ddq
            w[41, ec + i], w[42, ec + i], w[43, ec + i] = ph, dph, ddph
        rm = self.robot_model
        vel = np.empty((6, n)); acc = np.empty((6, n))
        for i in range(n):                                                 # :568-587
            p_c, J, dJ = rm.forward_kinematics(w[8:15, ec + i], w[15:22, ec + i])
            w[29:35, ec + i] = p_c
            vel[:, i] = J @ w[15:22, ec + i]
            acc[:, i] = J @ w[22:29, ec + i] + dJ @ w[15:22, ec + i]
        self.prev_traj = w[29:35, :]
        self.prev_vel = w[35:41, :]
        self._prev_w = w          # prev_acc / prev_jerk (:560-566) are only consumed after update(): computed lazily
        self.prev_acc = self.prev_jerk = None
        phi_opt = w[41, ec:]
        iw_ref_before = np.copy(self.iw_ref)
        if phi_opt[0] > phi_switch[1]:                                     # :594-604
            self.pr_ref = R.from_matrix(self.ref_path.r[self.ref_path.sector + 1]).as_rotvec()
            self.pr_ref = integrate_rotation_reference(self.pr_ref, dp_ref[3:, 1], phi_switch[1], phi_opt[0])
            self.iw_ref = p_ref[3:, 1] + (phi_opt[0] - phi_switch[1]) * dp_ref[3:, 1]
        else:
            self.pr_ref = integrate_rotation_reference(self.pr_ref, dp_ref[3:, 0], self.phi_current, phi_opt[0])
            self.iw_ref = p_ref[3:, 0] + (phi_opt[0] - phi_switch[0]) * dp_ref[3:, 0]
        self.phi_current = np.array([phi_opt[0]])                          # :607-611
        self.dphi_current = np.array([w[42, ec]])
        self.ddphi_current = np.array([w[43, ec]])
        self.dddphi_current = np.array([jerk_phi[0]])
        traj_data = {'p': w[29:35, ec:], 'v': vel, 'a': acc, 'q': w[8:15, ec:], 'dq': w[15:22, ec:], 'ddq': w[22:29, ec:],
                     'dddq': jerk, 'phi': phi_opt, 'dphi': w[42, ec:], 'ddphi': w[43, ec:], 'dddphi': jerk_phi}
        if not self.log:
            return traj_data, None, None
        ref_data, err_data = self._logging_data(q0, dq0, p0, iw_ref_before, w[29:35, ec:], vel, phi_opt, w[42, ec:], aux)
        return traj_data, ref_data, err_data

    def _segment_rows(self, phi, phi_switch):
        """Row picked by the reference's nested if_else selections at path parameter phi (bound_mpc_functions.py:13-40):
        -> (row of S-row arrays, row of the two-row window used for bp1/bp2, row of the (S+1)-row tube coefficients)."""
        S = self.nr_segs
        seg = S - 1
        for i in reversed(range(S - 1)):
            if phi < phi_switch[i + 1]:
                seg = i
        return seg, min(seg, S - 2), seg          # tube rows: row S (phi beyond the window) is defined as row S-1

    def _logging_data(self, q0, dq0, p0, iw_ref_before, traj, vel, phi_opt, dphi_opt, aux):
        """ref_data / err_data of BoundMPC.py:614-752: per stage of the (re-integrated) plan the path reference with its tube
        bounds and the position / orientation errors in the path frame, then the rotation references and orientation errors
        recomputed from the integrated rotation reference (:712-752).  Lists of flat arrays, one entry per stage, same keys."""
        n, dt, S = traj.shape[1], self.dt, self.nr_segs
        p_ref, dp_ref, sw = aux["p_ref"], aux["dp_ref"], aux["phi_switch"]
        jl, jr = aux["jac_l"], aux["jac_r"]
        a4, a3, a2, a1, a0 = aux["a"]
        # integrated angular velocity of the plan (trapezoid from the measured state, :574-586), sign flip after failures (:587-589)
        iw = np.empty((6, n))
        om_prev = (self.robot_model.jacobian_fk(np.asarray(q0, dtype=float)) @ np.asarray(dq0, dtype=float))[3:]
        for i in range(n):
            base = iw[3:, i - 1] if i > 0 else np.asarray(p0, dtype=float)[3:]
            iw[3:, i] = base + 0.5 * dt * (om_prev + vel[3:, i])
            om_prev = vel[3:, i].copy()
        iw[:3] = traj[:3]
        if self.error_count > 0 and np.linalg.norm(p0[3:] - iw[3:, 0]) > 3.1:
            iw[3:] *= -1
        ref_data = {k: [] for k in ("p", "dp", "ddp", "dp_normed", "r_par_bound", "bound_lower", "bound_upper", "e_p_off", "e_r_off",
                                    "bp1", "bp2", "br1", "br2", "v1", "v2", "v3")}
        err_data = {k: [] for k in ("e_p", "de_p", "e_p_par", "e_p_orth", "de_p_par", "de_p_orth", "e_r", "de_r", "e_r_par", "e_r_orth1",
                                    "e_r_orth2")}
        dtau0 = self.dtau_init[:, 0]
        for i in range(n):
            phi, dphi = phi_opt[i], dphi_opt[i]
            seg, segb, sega = self._segment_rows(phi, sw)
            x = phi - sw[seg]
            dp_d = dp_ref[:, seg].copy()
            p_d = dp_d * x + p_ref[:, seg]
            b = (((a4[sega] * x + a3[sega]) * x + a2[sega]) * x + a1[sega]) * x + a0[sega]         # 9 tube channels
            upper, lower = np.array([b[0], b[1], b[4], b[5]]), np.array([b[2], b[3], b[6], b[7]])
            dn = aux["dp_normed_ref"][:, seg].copy()
            br1, br2 = aux["br1"][:, seg].copy(), aux["br2"][:, seg].copy()
            v1, v2, v3 = aux["v1"][:, seg].copy(), aux["v2"][:, seg].copy(), aux["v3"][:, seg].copy()
            for k, val in (("p", p_d), ("dp", dp_d), ("ddp", 0 * dp_d), ("dp_normed", dn), ("r_par_bound", np.array([b[8]])),
                           ("bound_lower", lower), ("bound_upper", upper), ("e_p_off", 0.5 * (upper[:2] + lower[:2])),
                           ("e_r_off", 0.5 * (upper[2:] + lower[2:])), ("bp1", aux["bp1"][:, segb].copy()), ("bp2", aux["bp2"][:, segb].copy()),
                           ("br1", br1), ("br2", br2), ("v1", v1), ("v2", v2), ("v3", v3)):
                ref_data[k].append(val)
            d = dp_d[:3]
            e = iw[:3, i] - p_d[:3]
            e_par = (d @ e) * d
            de = vel[:3, i] - d * dphi
            de_par = (d @ de) * d
            e_r = dtau0 + jl @ (iw[3:, i] - p0[3:]) - jr @ (p_d[3:] - iw_ref_before)
            de_r = jl @ vel[3:, i] - jr @ (dp_d[3:] * dphi)
            dl = e_r - dtau0
            for k, val in (("e_p", e), ("de_p", de), ("e_p_par", e_par), ("e_p_orth", e - e_par), ("de_p_par", de_par),
                           ("de_p_orth", de - de_par), ("e_r", e_r.copy()), ("de_r", de_r),
                           ("e_r_par", self.dtau_init_par[:, seg] + (dl @ v2) * dn),
                           ("e_r_orth1", self.dtau_init_orth1[:, seg] + (dl @ v1) * br1),
                           ("e_r_orth2", self.dtau_init_orth2[:, seg] + (dl @ v3) * br2)):
                err_data[k].append(val)
        # rotation reference integrated along the plan and the orientation error against it (:712-752)
        rot_err = lambda rv, ref: R.from_matrix(R.from_rotvec(rv).as_matrix() @ R.from_rotvec(ref).as_matrix().T).as_rotvec()
        ref_data["p"][0][3:] = self.pr_ref
        pr = np.copy(self.pr_ref)
        sec = self.ref_path.sector
        for i in range(n - 1):
            ref_data["p"][i][3:] = pr
            err_data["e_r"][i] = rot_err(traj[3:, i], pr)
            phi, nxt = phi_opt[i], phi_opt[i + 1]
            if nxt > sw[1] and phi < sw[1]:
                pr = integrate_rotation_reference(R.from_matrix(self.ref_path.r[sec + 1]).as_rotvec(), dp_ref[3:, 1], sw[1], nxt)
            elif nxt > sw[2] and phi < sw[2]:
                pr = integrate_rotation_reference(R.from_matrix(self.ref_path.r[sec + 2]).as_rotvec(), dp_ref[3:, 2], sw[2], nxt)
            elif nxt > sw[2]:
                pr = integrate_rotation_reference(pr, dp_ref[3:, 2], phi, nxt)
            elif nxt > sw[1]:
                pr = integrate_rotation_reference(pr, dp_ref[3:, 1], phi, nxt)
            else:
                pr = integrate_rotation_reference(pr, dp_ref[3:, 0], phi, nxt)
        ref_data["p"][-1][3:] = pr
        err_data["e_r"][-1] = rot_err(traj[3:, -1], pr)
        return ref_data, err_data


def structural_bounds(N):
    """lbx, ubx, lbg, ubg lists of the formulation (casadi_ocp_formulation.py:92-153,272-349)."""
    rm = RobotModel()
    inf = np.inf
    lbz = [float(rm.u_min)] * 8 + list(rm.q_lim_lower) + list(rm.dq_lim_lower) + [-inf] * 19 + [0.0, -inf, -inf]
    ubz = [float(rm.u_max)] * 8 + list(rm.q_lim_upper) + list(rm.dq_lim_upper) + [inf] * 19 + [inf, inf, inf]
    return lbz * N, ubz * N, ([0.0] * 36 + [-inf] * 7) * N, [0.0] * (43 * N)
