"""Batched receding-horizon streams: the whole tick of `BoundMPC.step()` on the device.

One stream = one trajectory (what one reference `BoundMPC` object is, BoundMPC.py:20-33 -- SURVEY.md 8b: "a
BoundMPC instance is stateful per trajectory").  The static part of a stream is its *path table* (the arrays
`ReferencePath.__init__` derives from the via points, ReferencePath.py:10-163), built here on the host once;
the dynamic part is the *stream state* (phi-state, rotation reference, window sector, error count, previous
solution).  Per tick the device runs  pack -> solve -> post  (csrc/bmpc_stream.inl + the solver kernel),
optionally as one captured hipGraph and optionally with the node's kinematic plant simulation, so a closed loop
needs no host round trip.  Re-planning (`BoundMPC.update`, BoundMPC.py:163-217): `StreamBatch.update` / `apply_update` write the
new path table and the state scalars update() sets, from then on the device takes the re-projection branch of step() (:335-369).
"""
import ctypes

import numpy as np
from scipy.spatial.transform import Rotation as R

from . import _lib

# csrc/bmpc_stream.inl
PT = dict(P=0, IW=3, DPN=6, DR=9, RRV=12, PLO=15, PUP=17, RLO=19, RUP=21, BP1=23, BP2=26, BR1=29, BR2=32, CUM=35, EPMIN=36, ERMIN=37,
          EPMAX=38, ERMAX=39, S=40, LEN=48)
SS = dict(SECTOR=0, HASPREV=1, ERRCNT=2, PHI=3, DPHI=4, DDPHI=5, DDDPHI=6, PRREF=7, IWREF=10, PHIMAX=13, W=14, NENT=29, USINGPREV=30, VALID=31,
          PREV=32)
RB = dict(Q=0, DQ=7, DDQ=14, P=21, V=27, XPHID=33, JERK=36, LEN=43)


def ss_len(N):
    """[header 32 | previous solution 44 N | Cartesian pos, vel, acc, jerk of the previous plan 4 x 3 x N | updated flag, pad]"""
    return SS["PREV"] + 56 * N + 2


def ss_updated(N):
    return SS["PREV"] + 56 * N


def path_table(rp, entries=None):
    """Flatten a host `ReferencePath` (boundmpc_amd.reference_path) into the device table [entries][48]."""
    M = len(rp.p)
    assert rp.phi_bias == 0
    n_rows = entries if entries is not None else M
    assert n_rows >= M
    T = np.zeros((n_rows, PT["LEN"]))
    at = lambda lst, j: lst[min(j, len(lst) - 1)]
    for j in range(M):
        e = T[j]
        e[0:3], e[3:6] = rp.p[j], at(rp.iw, j)
        dpj = at(rp.dp, j)
        e[6:9] = dpj / np.linalg.norm(dpj)
        e[9:12] = at(rp.dr, j)
        e[12:15] = R.from_matrix(at(rp.r, j)).as_rotvec()
        e[15:17], e[17:19] = at(rp.p_lower, j), at(rp.p_upper, j)
        e[19:21], e[21:23] = at(rp.r_lower, j), at(rp.r_upper, j)
        e[23:26], e[26:29], e[29:32], e[32:35] = at(rp.bp1, j), at(rp.bp2, j), at(rp.br1, j), at(rp.br2, j)
        e[35] = rp._cum[j]
        e[36], e[37], e[38], e[39], e[40] = at(rp.e_p_min, j), at(rp.e_r_min, j), at(rp.e_p_max, j), at(rp.e_r_max, j), at(rp.s, j)
    T[M:] = T[M - 1]
    return T, M


def initial_state(mpc, N):
    """Stream state of a freshly constructed host `BoundMPC` (before its first step)."""
    if getattr(mpc, "updated", False):
        raise ValueError("the host BoundMPC has been re-planned (update()): its warm start re-projects from Cartesian arrays the stream state does not "
                         "carry over; build the StreamBatch from a fresh object and call StreamBatch.update()")
    s = np.zeros(ss_len(N))
    s[SS["SECTOR"]] = mpc.ref_path.sector
    s[SS["PHI"]], s[SS["DPHI"]], s[SS["DDPHI"]], s[SS["DDDPHI"]] = mpc.phi_current[0], mpc.dphi_current[0], mpc.ddphi_current[0], mpc.dddphi_current[0]
    s[SS["PRREF"]:SS["PRREF"] + 3] = mpc.pr_ref
    s[SS["IWREF"]:SS["IWREF"] + 3] = mpc.iw_ref
    s[SS["PHIMAX"]] = mpc.phi_max[0]
    s[SS["W"]:SS["W"] + 15] = mpc.weights
    if mpc.prev_solution is not None:
        s[SS["HASPREV"]] = 1.0
        s[SS["PREV"]:SS["PREV"] + 44 * N] = np.asarray(mpc.prev_solution, dtype=float).ravel()
    s[SS["ERRCNT"]] = mpc.error_count
    return s


def apply_update(ss, N, S, pos_points, rot_points, pos_lim, rot_lim, bp1, br1, s, e_p_min, e_r_min, e_p_max, e_r_max, p, v, a, jerk, p0, weights,
                 entries=None):
    """Re-planning of one stream, the state part of BoundMPC.update() (BoundMPC.py:163-217) on the stream-state row `ss` (numpy,
    modified in place): new path table (returned, [entries][48]) and window start, path-parameter state projected onto the new first
    segment from the measured Cartesian state (p0, v, a, jerk), rotation reference, phi_max, weights, and the `updated` flag that
    switches the device's warm start to the re-projection branch of step() for good (the reference never clears it).  The previous
    solution, its Cartesian derivatives and the error count stay, as in the reference -- with one exception: a stream that had LOST its plan
    (error count >= N: the fused tick skips such a stream, status 3) is given a new attempt, its error count goes back to N - 1, so the tick after
    a re-planning solves it again on every launch shape (one more failure loses it again, a success resets the count as in BoundMPC.py:468)."""
    from .reference_path import ReferencePath
    from .bound_mpc import integrate_rotation_reference
    rp = ReferencePath(pos_points, rot_points, pos_lim, rot_lim, bp1, br1, s, e_p_min, e_r_min, e_p_max, e_r_max, S)
    if entries is not None and len(rp.p) > entries:
        raise ValueError(f"the new path has {len(rp.p)} table entries, the stream batch was built for {entries}: construct the StreamBatch "
                         "with the longest path it will ever follow")
    T, M = path_table(rp, entries)
    dp0 = rp.dp[0] / np.linalg.norm(rp.dp[0])
    phi = float((np.asarray(p0, dtype=float)[:3] - np.asarray(pos_points[0], dtype=float)) @ dp0)
    dpn = rp.dpd[:3, 0]
    ss[SS["SECTOR"]] = rp.sector
    ss[SS["PHI"]], ss[SS["DPHI"]], ss[SS["DDPHI"]], ss[SS["DDDPHI"]] = phi, float(v[:3] @ dpn), float(a[:3] @ dpn), float(jerk[:3] @ dpn)
    ss[SS["PRREF"]:SS["PRREF"] + 3] = integrate_rotation_reference(R.from_matrix(rot_points[0]).as_rotvec(), rp.dr[0], 0.0, phi)
    ss[SS["IWREF"]:SS["IWREF"] + 3] = rp.pd[3:, 0] + phi * rp.dpd[3:, 0]
    ss[SS["PHIMAX"]] = rp.phi_max - 0.0001
    ss[SS["W"]:SS["W"] + 15] = np.asarray(weights, dtype=float)
    ss[SS["NENT"]] = M
    ss[ss_updated(N)] = 1.0
    if ss[SS["ERRCNT"]] >= N:
        ss[SS["ERRCNT"]] = N - 1
    return T, M


def _poff(S):
    """Offsets of the parameter vector p (casadi_ocp_formulation.py:361-376; the same map as make_poff in csrc/bmpc_wave.inl)."""
    o, c = {}, 0
    for k, n in (("q0", 7), ("dq0", 7), ("ddq0", 7), ("phi0", 3), ("p0", 6), ("v0", 6), ("iwref0", 3), ("dtau", 3), ("ipar", 3 * S), ("io1", 3 * S), ("io2", 3 * S),
                 ("xphid", 3), ("jerk", 7), ("jerkphi", 1), ("sw", S + 1), ("jacr", 9), ("jacl", 9), ("pref", 6 * S), ("dpref", 6 * S), ("dpn", 3 * S),
                 ("bp1", 3 * S), ("bp2", 3 * S), ("br1", 3 * S), ("br2", 3 * S), ("a4", 9 * (S + 1)), ("a3", 9 * (S + 1)), ("a2", 9 * (S + 1)),
                 ("a1", 9 * (S + 1)), ("a0", 9 * (S + 1)), ("w", 15), ("phimax", 1), ("dphimax", 1), ("v1", 3 * S), ("v2", 3 * S), ("v3", 3 * S), ("qd", 7)):
        o[k] = c; c += n
    o["size"] = c
    return o


def tube_excess_of_state(p, S=4, rows=False):
    """BoundMPC's contract is the error bound: how far is the MEASURED state of a tick outside its tubes?  `p` [B][n_p] are the parameter vectors
    packed for a tick (device or host copies of what bmpc_stream_pack wrote): they hold the measured pose p0, the path parameter phi0 and the
    window of the path with its tube quartics, i.e. everything the five tube rows of casadi_ocp_formulation.py:316-349 need -- evaluated here at
    NODE 0, the state the plant is in (the NLP constrains nodes 1..N: node 0 is where the previous plans have taken the plant).  At node 0 the
    orientation-error components are the packed init_par / init_orth1 / init_orth2 of the current segment (compute_initial_rot_errors,
    util_functions.py:11-31: the exact zyx split of the measured orientation error), the position error is p0 - p_d(phi0).
    Returns (excess_pos [B][2], excess_rot [B][3]) in the LINEAR form |l_m| - |w_m| (metres, radians; <= 0: inside): position rows along bp1, bp2
    (rows 39, 40 of a stage), orientation rows tangential, br1, br2 (rows 38, 41, 42).  rows=True: (l [B][5], w [B][5]) in the row order 38..42
    instead (the reference's rows are l^2 - w^2)."""
    p = np.atleast_2d(np.asarray(p, dtype=float))
    o, B = _poff(S), len(p)
    want_rows, rows = rows, np.arange(B)
    sw = p[:, o["sw"]:o["sw"] + S + 1]
    phi = p[:, o["phi0"]]
    seg = np.full(B, S - 1)
    for i in range(S - 2, -1, -1):      # bound_mpc_functions.py:13-20
        seg = np.where(phi < sw[:, i + 1], i, seg)
    segb = np.minimum(seg, max(S - 2, 0))      # bp1 / bp2 never select the last window segment (:34-40)
    x = phi - sw[rows, seg]
    col = lambda key, n, sg: np.stack([p[rows, o[key] + c * S + sg] for c in range(n)], axis=1)      # [coord][seg] blocks
    b = np.zeros((B, 9))
    for ch in range(9):
        a4, a3, a2, a1, a0 = (p[rows, o[k] + ch * (S + 1) + seg] for k in ("a4", "a3", "a2", "a1", "a0"))
        b[:, ch] = (((a4 * x + a3) * x + a2) * x + a1) * x + a0
    pref, dpref = col("pref", 6, seg), col("dpref", 6, seg)
    e_p = p[:, o["p0"]:o["p0"] + 3] - (pref[:, :3] + dpref[:, :3] * x[:, None])
    bp = (col("bp1", 3, segb), col("bp2", 3, segb))
    l, w = np.zeros((B, 5)), np.zeros((B, 5))
    for m in range(2):
        off, hw = 0.5 * (b[:, m] + b[:, 2 + m]), 0.5 * (b[:, m] - b[:, 2 + m])
        l[:, 1 + m], w[:, 1 + m] = np.einsum("bi,bi->b", e_p, bp[m]) - off, hw
    dn, br1, br2 = col("dpn", 3, seg), col("br1", 3, seg), col("br2", 3, seg)
    ini = lambda key: np.stack([p[rows, o[key] + 3 * seg + c] for c in range(3)], axis=1)      # [seg][xyz] blocks
    ipar, io1, io2 = ini("ipar"), ini("io1"), ini("io2")
    l[:, 0], w[:, 0] = np.einsum("bi,bi->b", dn, ipar), b[:, 8]
    for m, (br, io) in enumerate(((br1, io1), (br2, io2))):
        off, hw = 0.5 * (b[:, 4 + m] + b[:, 6 + m]), 0.5 * (b[:, 4 + m] - b[:, 6 + m])
        l[:, 3 + m], w[:, 3 + m] = np.einsum("bi,bi->b", br, io) - off, hw
    if want_rows:
        return l, w
    ex = np.abs(l) - np.abs(w)
    return ex[:, 1:3], ex[:, [0, 3, 4]]


def robot_record(q, dq, ddq, p_lie, v, x_phi_d, jerk):
    return np.concatenate([q, dq, ddq, p_lie, v, x_phi_d, jerk]).astype(float)


def unpack_traj(t, N):
    """Trajectory record -> the reference's traj_data dict (BoundMPC.py:757-770) + flags."""
    t = np.asarray(t)
    n = int(t[-4])
    o, out = 0, {}
    for k in ("q", "dq", "ddq", "dddq"):
        out[k] = t[o:o + 7 * N].reshape(7, N)[:, :n]; o += 7 * N
    for k in ("p", "v", "a"):
        out[k] = t[o:o + 6 * N].reshape(6, N)[:, :n]; o += 6 * N
    for k in ("phi", "dphi", "ddphi", "dddphi"):
        out[k] = t[o:o + N][:n]; o += N
    return out, dict(n_valid=n, using_previous=bool(t[-3]), success=bool(t[-2]), g_viol=float(t[-1]))


class StreamBatch:
    """B streams on the GPU.  `mpcs`: list of freshly constructed host `boundmpc_amd.bound_mpc.BoundMPC` objects (used only to
    read their path and initial state; they are not advanced)."""

    def __init__(self, solver, mpcs, device="cuda"):
        import torch
        self.solver, self.B, self.N, self.S = solver, len(mpcs), solver.N, solver.S
        # the warm starts of a stream are the reference's own (stream_pack: its cold start or the shifted plan, BoundMPC.py:316-375): taken as given, also
        # by the stateless ticks (cold duals); on closed loops the rollout of a start that is off its dynamics costs plans (oracle/bmpc_oracle.c solve_one)
        # (a setting of the caller's handle: the previous value comes back in close(); an A/B library from before the option has no such entry point)
        self._rollout_was = solver.get_start_rollout() if hasattr(solver._lib, "bmpc_get_start_rollout") else None
        if self._rollout_was is not None:
            solver.set_start_rollout(False)
        lens = [ctypes.c_int() for _ in range(4)]
        _lib.check(solver._lib.bmpc_stream_lengths(solver._h, *[ctypes.byref(v) for v in lens]), "bmpc_stream_lengths")
        self.pt_len, self.ss_len, self.rb_len, self.tr_len = (v.value for v in lens)
        assert self.pt_len == PT["LEN"] and self.rb_len == RB["LEN"] and self.ss_len == ss_len(self.N)
        self.entries = max(len(m.ref_path.p) for m in mpcs)
        tabs, states = [], []
        for m in mpcs:
            assert m.N == self.N and m.nr_segs == self.S and abs(m.dt - solver.dt) < 1e-15
            T, M = path_table(m.ref_path, self.entries)
            s = initial_state(m, self.N); s[SS["NENT"]] = M
            tabs.append(T); states.append(s)
        t64 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=device)
        self.path, self.state = t64(np.stack(tabs)), t64(np.stack(states))
        self.robot = torch.zeros((self.B, self.rb_len), dtype=torch.float64, device=device)
        self.p = torch.empty((self.B, solver.n_p), dtype=torch.float64, device=device)
        self.x0 = torch.empty((self.B, solver.n_w), dtype=torch.float64, device=device)
        self.dual = solver.new_state(self.B, device)
        self.x = torch.empty((self.B, solver.n_w), dtype=torch.float64, device=device)
        self.g = torch.empty((self.B, solver.n_g), dtype=torch.float64, device=device)
        self.iters = torch.zeros((self.B,), dtype=torch.int32, device=device)
        self.status = torch.zeros((self.B,), dtype=torch.int32, device=device)
        self.kkt = torch.zeros((self.B,), dtype=torch.float64, device=device)
        self.traj = torch.zeros((self.B, self.tr_len), dtype=torch.float64, device=device)
        self._graphs = {}
        solver._children.add(self)

    def update(self, b, *args, **kw):
        """Re-plan stream b: `apply_update` on a host copy of its state row, then the new table and state go back to the device
        (a rare event; the per-tick work stays on the device).  Arguments as apply_update after (ss, N, S)."""
        import torch
        torch.cuda.synchronize(self.state.device)
        row = self.state[b].cpu().numpy().copy()
        T, M = apply_update(row, self.N, self.S, *args, entries=self.entries, **kw)
        self.path[b].copy_(torch.as_tensor(T, dtype=torch.float64))
        self.state[b].copy_(torch.as_tensor(row, dtype=torch.float64))
        return float(row[SS["PHIMAX"]])

    def set_robot(self, rec):
        import torch
        self.robot.copy_(torch.as_tensor(np.ascontiguousarray(rec), dtype=torch.float64))

    def _stream(self, stream):
        import torch
        return ctypes.c_void_p((stream if stream is not None else torch.cuda.current_stream(self.path.device)).cuda_stream)

    def pack(self, warm_dual=False, stream=None, continue_rejected=False):
        """continue_rejected (real-time mode): the warm start continues from the solver's last iterate when the acceptance rule rejected
        it (bmpc_stream_pack_rt with xlast = x), as the fused tick does; False = restart from the last accepted plan (the reference)."""
        dp = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(self.solver._lib.bmpc_stream_pack_rt(self.solver._h, self.B, dp(self.path), self.entries, dp(self.state), dp(self.robot), dp(self.p),
                                                        dp(self.x0), dp(self.dual) if warm_dual else None, dp(self.x) if continue_rejected else None,
                                                        self._stream(stream)), "bmpc_stream_pack_rt")

    def post(self, simulate=True, stream=None, accept_capped=False):
        dp = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(self.solver._lib.bmpc_stream_post(self.solver._h, self.B, dp(self.path), self.entries, dp(self.state), dp(self.robot), dp(self.x),
                                                     dp(self.g), dp(self.status), dp(self.traj), int(bool(simulate)) | (2 if accept_capped else 0),
                                                     self._stream(stream)), "bmpc_stream_post")

    def tick(self, max_iter=0, warm_dual=False, simulate=True, stream=None, accept_capped=False, fused=True):
        """pack -> solve -> post of one tick: one fused launch (bmpc_stream_tick; N <= 11, B within the resident waves) or, with
        fused=False, the three launches of the separate entry points (see tick_graph for the captured form).
        accept_capped: real-time mode -- an iteration-capped iterate is judged by the reference's violation rule with the handle's
        real-time threshold (BatchedOCPSolver.set_rt_feasibility_tol) and replaced by the previous plan if it fails."""
        if max_iter and not warm_dual:
            raise ValueError("an iteration cap needs the dual state (warm_dual=True): the multipliers must be shifted with the plan")
        if fused:
            dp = lambda t: ctypes.c_void_p(t.data_ptr())
            _lib.check(self.solver._lib.bmpc_stream_tick(
                self.solver._h, self.B, dp(self.path), self.entries, dp(self.state), dp(self.robot), dp(self.p), dp(self.x0),
                dp(self.dual) if warm_dual else None, int(max_iter), dp(self.x), dp(self.g), dp(self.iters), dp(self.status), dp(self.kkt),
                dp(self.traj), int(bool(simulate)) | (2 if accept_capped else 0), self._stream(stream)), "bmpc_stream_tick")
            return
        self.pack(warm_dual, stream, continue_rejected=accept_capped)      # the same continuation rule as the fused launch
        out = dict(x=self.x, g=self.g, iters=self.iters, status=self.status, kkt=self.kkt)
        self.solver.solve_batch(self.p, self.x0, out=out, want=("g", "iters", "status", "kkt"), stream=stream,
                                state=self.dual if warm_dual else None, max_iter=max_iter)
        self.post(simulate, stream, accept_capped)

    def tick_with_fallback(self, fallback, max_iter=24, simulate=True, stream=None):
        """A tick of the converged loops with a barrier-level fallback (round 5): pack -> solve to tolerance with at most `max_iter` iterations (dual state
        carried) -> the streams whose solve did not converge are solved AGAIN from the same warm start by `fallback`, a handle on a fixed barrier level
        (BatchedOCPSolver(fixed_barrier=1.0, max_iter=14, ...)): a dozen Newton steps on a smooth barrier problem, far from the tube walls -> post, where
        the reference's acceptance rule (BoundMPC.py:460-465: solver success or summed violation below the handle's threshold, set it to the reference's 1e-4)
        decides for every stream.  What the converged loops lose their plans on are ticks whose minimiser cannot be reached (or does not exist) from the
        shifted plan; a plan of the level-1 problem is still a feasible trajectory, and the next tick usually converges again (CPU replay, 256 streams:
        14 lose their plan instead of 33; the restoration phase: 17-19, at three times the tick).  The fallback streams restart their dual state cold.
        Not captured in a graph: the count of failed streams is read on the host."""
        import torch
        N57 = self.N * 57
        self.pack(True, stream)
        dual0 = self.dual.clone()
        out = dict(x=self.x, g=self.g, iters=self.iters, status=self.status, kkt=self.kkt)
        self.solver.solve_batch(self.p, self.x0, out=out, want=("g", "iters", "status", "kkt"), stream=stream, state=self.dual, max_iter=max_iter)
        idx = torch.nonzero(self.status != 0).flatten()
        if idx.numel():
            st = dual0[idx].contiguous(); st[:, N57].clamp_(min=1e-9)      # warm: the multipliers of the shifted plan bound the initial slacks
            o = fallback.solve_batch(self.p[idx].contiguous(), self.x0[idx].contiguous(), want=("g", "iters", "status", "kkt"), stream=stream, state=st)
            self.x[idx] = o["x"]; self.g[idx] = o["g"]; self.iters[idx] += o["iters"]; self.kkt[idx] = o["kkt"]
            self.status[idx] = torch.where(o["status"] == 0, torch.ones_like(o["status"]), o["status"])      # a level plan is never "converged": the rule decides
            self.dual[idx] = 0.0
        self.post(simulate, stream, accept_capped=True)
        return int(idx.numel())

    def tick_graph(self, max_iter=0, warm_dual=False, simulate=True, stream=None, accept_capped=False):
        """The same tick replayed from a hipGraph captured on first use (bmpc_stream_graph_create)."""
        key = (int(max_iter), bool(warm_dual), bool(simulate), bool(accept_capped))
        if key not in self._graphs:
            if max_iter and not warm_dual:
                raise ValueError("an iteration cap needs the dual state (warm_dual=True)")
            dp = lambda t: ctypes.c_void_p(t.data_ptr())
            g = ctypes.c_void_p()
            _lib.check(self.solver._lib.bmpc_stream_graph_create(
                self.solver._h, self.B, dp(self.path), self.entries, dp(self.state), dp(self.robot), dp(self.p), dp(self.x0),
                dp(self.dual) if warm_dual else None, int(max_iter), dp(self.x), dp(self.g), dp(self.iters), dp(self.status), dp(self.kkt),
                dp(self.traj), int(bool(simulate)) | (2 if accept_capped else 0), ctypes.byref(g)), "bmpc_stream_graph_create")
            self._graphs[key] = g
        # (a replay requested on the legacy null stream is run by the library on a stream of the handle, bracketed by events:
        # bmpc_graph_launch)
        _lib.check(self.solver._lib.bmpc_graph_launch(self._graphs[key], self._stream(stream)), "bmpc_graph_launch")

    def close(self):
        for g in self._graphs.values():
            self.solver._lib.bmpc_graph_destroy(g)
        self._graphs = {}
        if getattr(self, "_rollout_was", None) is not None and getattr(self.solver, "_h", None):      # the handle's start-rollout setting as it was found
            self.solver.set_start_rollout(self._rollout_was)
            self._rollout_was = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
