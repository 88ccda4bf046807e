"""Synthetic problem batches of SURVEY.md section 8(d): random initial joint states around the
experiment1 start (nodes/experiment1_runner.py:22-27 of the reference), each with its own copy of the
experiment1 via-point pattern rebuilt from p0_i = FK(q0_i) (:57-74), packed by the host mirror of
BoundMPC.step() (boundmpc_amd.bound_mpc.BoundMPC.pack) into (p [B][n_p], x0 [B][44N]).
Pure host code (numpy/scipy); no GPU is touched, so it can run in forked worker processes."""
import multiprocessing as mp
import os

import numpy as np
from scipy.spatial.transform import Rotation as R

from .bound_mpc import BoundMPC
from .robot_model import RobotModel

Q0_EXP1 = np.array([0.0, np.pi / 3.5, 0.0, -np.pi / 3.5, 0.0, -12.85714286 * np.pi / 180, 0.0])


def default_weights():
    """path_utils.get_default_weights (path_utils.py:42-68) -- values."""
    return np.array([1000.0, 1.0, 0.1, 0.1, 0.5, 0.05, 8.0, 5.0, 4.0, 0.5, 0.01, 0.01, 0.001, 0.0001, 10.0])


class Params:
    """Fields of MPCParams.srv that BoundMPC reads (BoundMPC.py:35-62)."""

    def __init__(self, n=10, dt=0.1, weights=None, nr_segs=4, real_time=True, build=False):
        self.n, self.dt, self.nr_segs, self.real_time, self.build = n, dt, nr_segs, real_time, build
        self.weights = list(default_weights() if weights is None else weights)


class _NoSolver:
    def generate_dependencies(self, *a, **k):
        pass

    def __call__(self, **k):
        raise RuntimeError("workload packing objects have no solver attached")


def experiment1_path(p0fk, e_p_min=0.01, e_r_min=15 * np.pi / 180, e_p_max=0.5, e_r_max=45 * np.pi / 180):
    """Via points / rotations / limits of experiment1 for a start pose p0fk (6)."""
    p0 = p0fk[:3]
    r0 = R.from_rotvec(p0fk[3:])
    x = p0[0]
    p_via = [p0, p0 + np.array([-2 * x, 0.0, 0.0]), p0 + np.array([-x, x, 0.0]), p0 + np.array([-x, -x, 0.0]), p0.copy()]
    r1 = R.from_euler('XYZ', [0, 0, -np.pi]) * r0
    r2 = R.from_euler('XYZ', [0, 0, -np.pi / 2]) * r1
    r3 = R.from_euler('XYZ', [0, np.pi / 2, 0]) * R.from_euler('XYZ', [np.pi / 1.001, 0, 0]) * r2
    r_via = [r.as_matrix() for r in (r0, r1, r2, r3, r0)]
    n = 5
    lim = lambda v: [np.array([v, v]) for _ in range(n)]
    bp1 = [np.array([0.0, 0.0, 1.0]) for _ in range(n)]
    br1 = [np.array([0.0, 1.0, 0.0]), np.array([0.0, 1.0, 0.0])] + [np.array([0.0, 0.0, 1.0]) for _ in range(n - 2)]
    return dict(pos_points=p_via, rot_points=r_via, pos_lim=[lim(-1.0), lim(1.0)], rot_lim=[lim(-1.0), lim(1.0)], bp1=bp1, br1=br1,
                s=[0.0] * n, e_p_min=[e_p_min] * n, e_r_min=[e_r_min] * n, e_p_max=[e_p_max] * n, e_r_max=[e_r_max] * n)


Q0_EXP2 = np.array([0.0, 0.0, 0.0, -np.pi / 1.8, 0.0, np.pi / 2 - np.pi / 1.8, 0.0])


def experiment2_path(p0fk):
    """Via points / rotations / limits of experiment2 (nodes/experiment2_runner.py:20-127 of the reference; defaults of
    path_utils.get_default_path :4-39): five via points with per-segment ASYMMETRIC tube limits (segment 1: position +-[0.01, 1.0],
    rotation +-0.11; segments 3, 4: +-0.1) and mixed basis vectors -- the secondary parity target of SURVEY 8(d), config 1b."""
    p0 = p0fk[:3]
    r0 = R.from_rotvec(p0fk[3:])
    r1 = R.from_euler('XYZ', [np.pi / 2, 0, 0]) * r0
    r2 = R.from_euler('XYZ', [0, 0, -np.pi / 3]) * r1
    turn = R.from_euler('XYZ', [np.pi / 2, 0, 0]) * R.from_euler('XYZ', [0, 0, -np.pi / 2]) * r1
    r3 = R.from_euler('XYZ', [0, 0, np.pi / 2.01]) * turn
    r4 = R.from_euler('XYZ', [0, 0, np.pi / 2]) * turn
    p_via = [p0, p0 + np.array([-0.2, -0.0, 0.1]), p0 + np.array([-0.6, -0.6, 0.1]), p0 + np.array([-0.8, -0.5, -0.2]), p0 + np.array([-0.8, -0.5, -0.5])]
    r_via = [r.as_matrix() for r in (r0, r1, r2, r3, r4)]
    arr = lambda rows: [np.array(v) for v in rows]
    p_up = [[1.0, 1.0], [0.01, 1.0], [1.0, 1.0], [0.1, 0.1], [0.1, 0.1]]
    r_up = [[1.0, 1.0], [0.11, 0.11], [1.0, 1.0], [0.1, 0.1], [0.1, 0.1]]
    n = 5
    return dict(pos_points=p_via, rot_points=r_via, pos_lim=[arr([[-a, -b] for a, b in p_up]), arr(p_up)], rot_lim=[arr([[-a, -b] for a, b in r_up]), arr(r_up)],
                bp1=arr([[0., 0., 1.], [0., 0., 1.], [0., 0., 1.], [0., 1., 0.], [0., 1., 0.]]),
                br1=arr([[0., 0., 1.], [0., 1., 0.], [0., 0., 1.], [0., 1., 0.], [0., 1., 0.]]),
                s=[0.0] * n, e_p_min=[0.01] * n, e_r_min=[15 * np.pi / 180] * n, e_p_max=[0.20] * n, e_r_max=[45 * np.pi / 180] * n)


def make_mpc(q0, N=10, S=4, dt=0.1, tight=False, solver=None, weights=None, experiment=1):
    rm = RobotModel()
    p0fk = rm.fk(q0)
    kw = dict(e_p_min=0.002, e_p_max=0.05, e_r_min=3 * np.pi / 180, e_r_max=10 * np.pi / 180) if tight else {}
    path = experiment1_path(p0fk, **kw) if experiment == 1 else experiment2_path(p0fk)      # (tight tubes: the synthetic batches of experiment 1 only)
    mpc = BoundMPC(p0=p0fk.copy(), params=Params(n=N, dt=dt, nr_segs=S, weights=weights),
                   solver=solver if solver is not None else _NoSolver(), **path)
    return mpc, p0fk


def pack_cold(q0, N=10, S=4, dt=0.1, tight=False):
    """(p, x0) of the first tick (cold start, zero velocities) for start configuration q0."""
    mpc, p0fk = make_mpc(q0, N, S, dt, tight)
    x_phi_d = np.array([mpc.phi_max[0], 0.0, 0.0])
    w0, params, _ = mpc.pack(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), x_phi_d, np.zeros(7))
    return params, np.array(w0)


def random_q0(B, seed):
    rng = np.random.default_rng(seed)
    qlim = 0.9 * np.array(RobotModel().q_lim_upper)
    return np.clip(Q0_EXP1 + rng.uniform(-0.3, 0.3, (B, 7)), -qlim, qlim)


def _chunk(args):
    q0s, N, S, dt, tight = args
    P = np.empty((len(q0s), 141 + 91 * S)); X = np.empty((len(q0s), 44 * N))
    for i, q0 in enumerate(q0s):
        P[i], X[i] = pack_cold(q0, N, S, dt, tight)
    return P, X


def make_batch(B, seed=0, N=10, S=4, dt=0.1, tight=False, workers=None, rows=None):
    """SURVEY 8(d) config 2/3/4 generator -> (p [B][n_p], x0 [B][44N], q0 [B][7]).
    rows=(lo, hi): only that slice of the B-problem batch (a rank's shard of a sharded batch, boundmpc_amd.distributed.shard_range)."""
    q0 = random_q0(B, seed)
    if rows is not None:
        q0 = q0[rows[0]:rows[1]]
        B = len(q0)
    workers = workers if workers is not None else max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 2)))
    if workers <= 1 or B < 64:
        P, X = _chunk((q0, N, S, dt, tight))
    else:
        parts = np.array_split(q0, min(B, workers * 4))
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_chunk, [(c, N, S, dt, tight) for c in parts])
        P = np.concatenate([r[0] for r in res]); X = np.concatenate([r[1] for r in res])
    return P, X, q0
