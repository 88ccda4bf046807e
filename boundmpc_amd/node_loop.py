"""The control loop of the reference's ROS 2 node WITHOUT ROS.

`NodeLoop` mirrors what `MPCNode` (bound_mpc/nodes/bound_mpc_node.py) does around `BoundMPC.step()`: `reset()` (:48-83: build the MPC
object from the received trajectory, x_phi_d = [phi_max, 0, 0], state at rest) and `step()` (:292-372: forward kinematics of the current
joint state, `mpc.step(...)`, the `fails` / switch-time bookkeeping, the kinematic plant simulation `integrate_joint` with the jerk
matrix [jerk_current, u_0, u_1], the new jerk).  rclpy, the services, the `MPCData` message and RViz are not here: what the node
publishes per tick (`publish_mpc_data`, :169-289) is handed to an optional callback as a plain dict with the message's field names, so a
ROS node -- or a test -- can forward it.  `run()` is the `main()` loop (:375-401) without the 10 Hz rate: until `phi_max - phi <= 0.01`
(what experiment1_runner.py:... waits for) or a tick budget.

The solver is whatever `BoundMPC` is given: the GPU solver behind `NlpSolverShim` (the drop-in), or any nlpsol-shaped callable.
"""
import time

import numpy as np

from .bound_mpc import BoundMPC, integrate_joint
from .robot_model import RobotModel


class NodeLoop:
    def __init__(self, p_via, r_via, p_limits, r_limits, bp1, br1, s, e_p_min, e_r_min, e_p_max, e_r_max, p0, q0, params, solver=None, publish=None):
        """Arguments = what the experiment runners send in the Trajectory service request (experiment1_runner.py:36,91: the tuple of
        path_utils.get_default_path; p_limits = [lower list, upper list] per via point, likewise r_limits) and their MPCParams (`params`: any
        object with n, dt, weights, nr_segs, real_time, build); `solver`: passed to BoundMPC (None: the GPU solver through NlpSolverShim);
        `publish`: callable(dict) or None.  (On the wire the reference names the two limit lists the other way round --
        util_functions.py:58-63 puts p_limits[0] into `p_upper` -- and the node passes [p_upper, p_lower] on, :55-56: ReferencePath receives
        [lower, upper] either way, ReferencePath.py:52-55.)"""
        self.traj = dict(p_via=p_via, r_via=r_via, p_limits=p_limits, r_limits=r_limits, bp1=bp1, br1=br1, s=s,
                         e_p_min=e_p_min, e_r_min=e_r_min, e_p_max=e_p_max, e_r_max=e_r_max)
        self.p0, self.q0, self.params, self.solver, self.publish = np.asarray(p0, dtype=float), np.asarray(q0, dtype=float), params, solver, publish
        self.robot_model = RobotModel()
        self.phi_bias = 0.0
        self.reset()

    def reset(self):
        """bound_mpc_node.py:48-83"""
        t = self.traj
        self.mpc = BoundMPC(t["p_via"], t["r_via"], t["p_limits"], t["r_limits"], t["bp1"], t["br1"], t["s"],
                            t["e_p_min"], t["e_r_min"], t["e_p_max"], t["e_r_max"], p0=self.p0.copy(), params=self.params, solver=self.solver)
        self.x_phi_d = np.array([self.mpc.phi_max[0], 0.0, 0.0])
        self.q, self.dq, self.ddq, self.jerk = self.q0.copy(), np.zeros(7), np.zeros(7), np.zeros(7)
        self.p_lie, self.v = self.p0.copy(), np.zeros(6)
        self.t_current, self.t0 = 0.0, 0.0
        self.t_mpc, self.t_overhead = 0.0, 0.0
        self.fails, self.t_switch, self.phi_switch = [], [], []
        self.log = True

    def step(self):
        """One tick, bound_mpc_node.py:292-372.  Returns (traj_data, ref_data, err_data) of mpc.step, or None when the MPC has run out
        of plan (N consecutive solver failures: BoundMPC.py:498-506 returns five Nones; the reference node then crashes at :318)."""
        start = time.time()
        self.p_lie, _, _ = self.robot_model.forward_kinematics(self.q, self.dq)
        traj_data, ref_data, err_data, self.t_mpc, iters = self.mpc.step(self.q, self.dq, self.ddq, self.p_lie, self.v, self.x_phi_d, self.jerk)
        if traj_data is None:
            return None
        if ref_data is None:
            self.log = False
        self.fails.append(1.0 if self.mpc.error_count > 0 else 0.0)
        if self.mpc.ref_path.switched:
            self.t_switch.append(self.t_current - self.mpc.dt)
            self.phi_switch.append(self.mpc.ref_path.phi_switch[0])
        self.t_current += self.mpc.dt
        jerk_traj = traj_data["dddq"]
        jerk_matrix = np.concatenate((self.jerk[:, None], jerk_traj[:, :2]), axis=1)
        new_state = integrate_joint(self.robot_model, jerk_matrix, self.q, self.dq, self.ddq, self.mpc.dt)
        self.q, self.dq, self.ddq, self.p_lie, self.v, self.a, self.j_cart = new_state
        self.jerk = jerk_traj[:, 0].copy()
        t_loop = time.time() - start
        self.t_overhead = t_loop - self.t_mpc
        if self.publish is not None:
            self.publish(self.mpc_data(traj_data, iters, t_loop))
        return traj_data, ref_data, err_data

    def mpc_data(self, traj_data, iters, t_loop):
        """The fields of the MPCData message the node fills per tick (publish_mpc_data, :169-289), without the logging lists of ref_data."""
        return dict(stamp=self.t_current, sector=int(self.mpc.ref_path.sector), phi_switch_vector=(np.asarray(self.mpc.ref_path.phi_switch) + self.phi_bias).tolist(),
                    t_comp=self.t_mpc, t_loop=t_loop, t_overhead=self.t_overhead, iterations=iters,
                    t_switch=self.t_switch + [self.t_current], phi_switch=(np.array(self.phi_switch + [self.mpc.phi_current[0]]) + self.phi_bias).tolist(),
                    fails=list(self.fails), phi=(traj_data["phi"] + self.phi_bias), dphi=traj_data["dphi"], ddphi=traj_data["ddphi"], dddphi=traj_data["dddphi"],
                    phi_max=self.mpc.phi_max[0] + self.phi_bias, p=traj_data["p"], v=traj_data["v"], a=traj_data["a"],
                    q=traj_data["q"], dq=traj_data["dq"], ddq=traj_data["ddq"], dddq=traj_data["dddq"])

    def run(self, max_ticks=1000, goal_tol=0.01):
        """main() (:375-401) without the 10 Hz rate; stops at the goal the experiment runners wait for (phi_max - phi <= goal_tol),
        when the plan is lost, or after max_ticks.  Returns the number of ticks."""
        for k in range(max_ticks):
            if self.step() is None:
                return k
            if self.mpc.phi_max[0] - self.mpc.phi_current[0] <= goal_tol:
                return k + 1
        return max_ticks
