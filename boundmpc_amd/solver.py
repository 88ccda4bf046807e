"""Batched OCP solver handle and the CasADi-`nlpsol`-compatible single-problem shim.

`BatchedOCPSolver` is the torch-facing wrapper of the C ABI (PyTorch-ROCm tensors are used
for device memory and streams only).  `NlpSolverShim` mirrors the call convention of the
object the reference creates at casadi_ocp_formulation.py:389 and calls at
BoundMPC.py:446-456, so that `BoundMPC.step()` reads like the reference's."""
import ctypes

import numpy as np

from . import _lib

NZ, NG = 44, 43


class BatchedOCPSolver:
    def __init__(self, N, S, dt, tol=1e-8, max_iter=500, mu_init=None, slack_push=None, exact_hessian=True, mu_warm=1e-2, stall_window=None, bound_margin=0.0,
                 restoration=None, resto_short=None, resto_cap=None, start_rollout=None, mu_min_fac=None, fixed_barrier=None, level_c=0.02):
        self._lib = _lib.load()
        o = _lib.Options()
        self._lib.bmpc_default_options_for(int(N), ctypes.byref(o))      # mu_init 0.1 / slack_push 1e-2 for N <= 11, 3.0 / 0.1 for longer horizons
        if mu_init is None:
            mu_init = o.mu_init
        if slack_push is None:
            slack_push = o.slack_push
        o.tol, o.max_iter, o.mu_init, o.slack_push, o.exact_hessian = tol, int(max_iter), mu_init, slack_push, int(exact_hessian)
        if fixed_barrier is not None:
            # Real-time iteration on ONE barrier level (closed-loop ticks under a time budget, bench_stream.py rtfix-*): mu_init = mu_warm = final level =
            # fixed_barrier.  A tick then spends its few iterations as Newton steps on the barrier problem whose solution the previous tick left nearby,
            # instead of restarting the barrier at mu_warm and re-converging through its levels; `tol` never fires (the complementarity stays at the
            # level): the tick's budget or iteration cap ends it and the caller's acceptance rule decides.  256 closed loops at 1 kHz: 95.7 % of the
            # streams keep a plan at tick p99 0.96 ms with 0.1 (restarted barrier: 76.6 % at p99 1.00 ms; loops solved to 1e-8: 93.0 % at 12.5 ms).
            # fixed_barrier = "auto" (round 6) or a pair (lo, hi): the level sets itself per stream -- the handle HOLDS the level a solve starts on
            # (bmpc_set_barrier_hold) and the device-side pack writes clamp(c (phi_max - phi), lo, hi) into the stream's dual state
            # (bmpc_stream_set_level_rule): hi far from the end of the path, lower near it, where the barrier of phi <= phi_max would stall the stream.
            # "auto" = (0.01, 0.1) with level_c = 0.02: 94.9 % of the 256 benchmark streams keep their plan over 130 ticks (level 0.1: 95.7 %, 0.01: 87.5 %)
            # AND the reference's two experiments reach their goals after 161 / 64 ticks (0.1: stalls 0.21 / 0.37 short; 0.01: 160 / 62; solved to 1e-8: 155 / 59).
            auto = fixed_barrier == "auto" or isinstance(fixed_barrier, (tuple, list))
            lo, hi = (0.01, 0.1) if fixed_barrier == "auto" else ((float(fixed_barrier[0]), float(fixed_barrier[1])) if auto else (float(fixed_barrier),) * 2)
            mu_init, mu_warm = hi, lo
            mu_min_fac = lo / float(tol)
            self._level_rule = (level_c, lo, hi) if auto else None
        o.mu_init = mu_init if fixed_barrier is not None else o.mu_init
        o.mu_warm = mu_warm
        if mu_min_fac is not None:
            o.mu_min_fac = float(mu_min_fac)      # final barrier level = tol * mu_min_fac (default 0.1); mu_init = mu_warm = tol * mu_min_fac: a FIXED barrier level (real-time ticks)
        o.bound_margin = float(bound_margin)      # joint limits tightened inside the solver (real-time modes; 0 = the reference's limits)
        if stall_window is not None:
            o.stall_window = int(stall_window)      # default: 40 for N <= 11, 20 for longer horizons (bmpc_default_options_for)
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.bmpc_create(int(N), int(S), float(dt), ctypes.byref(o), ctypes.byref(self._h)), "bmpc_create")
        if getattr(self, "_level_rule", None):
            _lib.check(self._lib.bmpc_set_barrier_hold(self._h, 1), "bmpc_set_barrier_hold")
            _lib.check(self._lib.bmpc_stream_set_level_rule(self._h, *[float(v) for v in self._level_rule]), "bmpc_stream_set_level_rule")
        # restoration phase (include/boundmpc_hip.h bmpc_set_restoration): None keeps the handle's default (on for N <= 11; 6 short steps; 40 iterations)
        if not (restoration is None and resto_short is None and resto_cap is None):
            _lib.check(self._lib.bmpc_set_restoration(self._h, -1 if restoration is None else int(restoration), -1 if resto_short is None else int(resto_short),
                                                      -1 if resto_cap is None else int(resto_cap)), "bmpc_set_restoration")
        if start_rollout is not None:      # rollout of a cold start that is not a trajectory (bmpc_set_start_rollout; default on)
            _lib.check(self._lib.bmpc_set_start_rollout(self._h, int(bool(start_rollout))), "bmpc_set_start_rollout")
        self.N, self.S, self.dt = int(N), int(S), float(dt)
        self.n_w, self.n_g, self.n_p = N * NZ, N * NG, 141 + 91 * S
        self.state_len = int(self._lib.bmpc_state_len(self._h))
        import weakref
        self._children = weakref.WeakSet()      # captured graphs (StepGraph, StreamBatch): closed before the handle, see close()
        # the handle's workspace, work queue and occupancy belong to the device that was current at bmpc_create
        try:
            import torch
            self.device_index = torch.cuda.current_device() if torch.cuda.is_available() else None
        except Exception:
            self.device_index = None

    def close(self):
        """Destroys the captured graphs made from this handle first, then the handle (a graph destroyed later would still be safe --
        the C handle is reference-counted by its graphs -- but could no longer launch)."""
        if getattr(self, "_h", None):
            for c in list(getattr(self, "_children", ())):
                try:
                    c.close()
                except Exception:
                    pass
            self._lib.bmpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- structural constants (casadi_ocp_formulation.py:384-391) ----
    def bounds(self):
        lbx, ubx, lbg, ubg = np.zeros(self.n_w), np.zeros(self.n_w), np.zeros(self.n_g), np.zeros(self.n_g)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(self._lib.bmpc_get_bounds(self._h, vp(lbx), vp(ubx), vp(lbg), vp(ubg)), "bmpc_get_bounds")
        return lbx, ubx, lbg, ubg

    def launch_info(self):
        g, l, s = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
        self._lib.bmpc_launch_info(self._h, ctypes.byref(g), ctypes.byref(l), ctypes.byref(s))
        return dict(grid=g.value, lds_bytes=l.value, scratch_bytes=s.value)

    def set_restoration(self, enabled=None, short_steps=None, cap=None):
        """Restoration phase of the solver (include/boundmpc_hip.h bmpc_set_restoration).  enabled: False / 0 never, True / 1 full (jam, stall, numerical
        breakdown; default for N <= 11), 2 after a numerical breakdown only (default for N > 11).  None keeps a value.  Re-capture graphs after changing it."""
        _lib.check(self._lib.bmpc_set_restoration(self._h, -1 if enabled is None else int(enabled), -1 if short_steps is None else int(short_steps),
                                                  -1 if cap is None else int(cap)), "bmpc_set_restoration")

    def set_start_rollout(self, enabled=True):
        """A stateless solve (solve_batch / solve_host without a dual state) whose x0 violates its own integrator chains by more than 0.5 starts from the
        rollout of x0's jerks (include/boundmpc_hip.h bmpc_set_start_rollout; default on; solves with a dual state and the reference's own starts are never touched).  Re-capture graphs after changing it."""
        _lib.check(self._lib.bmpc_set_start_rollout(self._h, int(bool(enabled))), "bmpc_set_start_rollout")

    def get_start_rollout(self):
        return bool(self._lib.bmpc_get_start_rollout(self._h))

    def get_restoration(self):
        e, s_, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self._lib.bmpc_get_restoration(self._h, ctypes.byref(e), ctypes.byref(s_), ctypes.byref(c)), "bmpc_get_restoration")
        return dict(enabled=e.value == 1, mode=e.value, short_steps=s_.value, cap=c.value)

    def set_second_attempt(self, cap):
        """Iterations of the SECOND ATTEMPT of a stateless solve that ends with status 2: once more from x0 on the barrier start of the short horizons
        (mu 0.1, slacks pushed to 1e-2); iterations add up, a second attempt that hits its cap keeps status 2.  Default 100 for N > 11 (on BASELINE
        configs[3] 22 of the 26 status-2 problems are feasible and converge this way), 0 = off for shorter horizons (include/boundmpc_hip.h)."""
        _lib.check(self._lib.bmpc_set_second_attempt(self._h, int(cap)), "bmpc_set_second_attempt")

    def get_second_attempt(self):
        return int(self._lib.bmpc_get_second_attempt(self._h))

    def set_queue_order(self, mode):
        """Work-queue order of a stateless batch larger than the resident waves: 1 = longest-expected-first by the objective at x0 (default for N > 11),
        0 = natural order (include/boundmpc_hip.h bmpc_set_queue_order).  Results do not depend on it."""
        _lib.check(self._lib.bmpc_set_queue_order(self._h, int(mode)), "bmpc_set_queue_order")

    def get_queue_order(self):
        return int(self._lib.bmpc_get_queue_order(self._h))

    def set_team_waves(self, waves=0):
        """Waves per problem: 0 (default) automatic -- a batch that fits into the resident teams of the device (256 on an MI355X) is solved by
        workgroups of 4 cooperating waves, one that fits into the resident pairs (512) by workgroups of 2, a larger one by one wave per problem;
        1 never teams; 2 pairs whatever the batch (N <= 11, S <= 4); 4 teams whenever the kernel exists (N <= 10, S <= 4).  Re-capture graphs after changing it."""
        _lib.check(self._lib.bmpc_set_team_waves(self._h, int(waves)), "bmpc_set_team_waves")

    def team_info(self, B):
        w, r, l = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self._lib.bmpc_team_info(self._h, int(B), ctypes.byref(w), ctypes.byref(r), ctypes.byref(l)), "bmpc_team_info")
        return dict(waves=w.value, resident_teams=r.value, lds_bytes=l.value)

    def set_rt_position_row_cap(self, cap_m2):
        """Real-time stream ticks: a position tube row (l^2 - w^2 of any stage, m^2) above the cap vetoes the iterate (include/boundmpc_hip.h
        bmpc_stream_set_rt_position_row_cap; 0 = off, the default).  Set it before the tick graph is captured."""
        _lib.check(self._lib.bmpc_stream_set_rt_position_row_cap(self._h, float(cap_m2)), "bmpc_stream_set_rt_position_row_cap")

    def set_rt_feasibility_tol(self, tol):
        """Threshold of the reference's acceptance rule (summed violation of g, BoundMPC.py:462-465) that stream ticks in real-time mode
        apply to an iteration-capped iterate (default 1e-4, the reference's).  Set it before the tick graph is captured."""
        _lib.check(self._lib.bmpc_stream_set_rt_feasibility_tol(self._h, float(tol)), "bmpc_stream_set_rt_feasibility_tol")

    def set_time_budget_us(self, microseconds):
        """Time budget of a fused closed-loop tick (0 = none): no further solver iteration is started once it is used up
        (bmpc_stream_set_time_budget).  Set it before the tick graph is captured."""
        _lib.check(self._lib.bmpc_stream_set_time_budget(self._h, float(microseconds)), "bmpc_stream_set_time_budget")

    def set_timing(self, keep=1):
        """HIP events around the solver kernel on its launch stream; the pairs of the last `keep` launches are kept (0/False = off)."""
        _lib.check(self._lib.bmpc_set_timing(self._h, int(keep)), "bmpc_set_timing")

    def set_latency_buffer(self, buf):
        """buf: float64 GPU tensor [>= B] that receives each solve's in-kernel duration in microseconds, or None."""
        self._lat = buf          # keep it alive while registered
        _lib.check(self._lib.bmpc_set_latency_buffer(self._h, ctypes.c_void_p(buf.data_ptr()) if buf is not None else None), "bmpc_set_latency_buffer")

    def kernel_ms(self, back=0):
        """duration of the launch `back` launches ago (waits for its stop event only)."""
        ms = ctypes.c_float()
        _lib.check(self._lib.bmpc_kernel_ms(self._h, int(back), ctypes.byref(ms)), "bmpc_kernel_ms")
        return ms.value

    def last_kernel_ms(self):
        return self.kernel_ms(0)

    # ---- dual state of a receding-horizon stream (bmpc_solve_batch_warm) ----
    def new_state(self, B, device="cuda"):
        """Zeroed state = cold start on first use."""
        import torch
        return torch.zeros((B, self.state_len), dtype=torch.float64, device=device)

    def shift_state(self, state):
        """Advance the horizon by one stage, as the host does with x0 (BoundMPC.py:372-375): rows of node k+1 -> node k,
        last node duplicated.  In place."""
        nu = state[:, :self.N * 57].view(-1, self.N, 57)
        nu[:, :-1] = nu[:, 1:].clone()
        return state

    def _check_io(self, p, x0, state):
        import torch
        if not (p.is_cuda and x0.is_cuda and p.dtype == torch.float64 and x0.dtype == torch.float64):
            raise ValueError("p and x0 must be float64 tensors on the GPU")
        B = p.shape[0]
        if p.shape != (B, self.n_p) or x0.shape != (B, self.n_w):
            raise ValueError(f"shape mismatch: p {tuple(p.shape)} x0 {tuple(x0.shape)}")
        if not (p.is_contiguous() and x0.is_contiguous()):
            raise ValueError("p and x0 must be contiguous (the kernel reads them asynchronously on the launch stream; a temporary copy "
                             "made here could be recycled by the allocator before the kernel has run)")
        if self.device_index is not None and (p.device.index != self.device_index or x0.device.index != self.device_index):
            raise ValueError(f"p / x0 live on cuda:{p.device.index}, the solver handle was created on cuda:{self.device_index}")
        if state is not None and not (state.is_cuda and state.dtype == torch.float64 and state.is_contiguous()
                                      and state.shape == (B, self.state_len)):
            raise ValueError(f"state must be a contiguous float64 GPU tensor of shape ({B}, {self.state_len})")
        return B

    # ---- batched device solve ----
    def solve_batch(self, p, x0, out=None, want=("g", "lam_g", "lam_x", "f", "iters", "status", "kkt"), stream=None, state=None, max_iter=0):
        """p [B][n_p], x0 [B][n_w]: CUDA(ROCm) float64 contiguous tensors.  Returns dict of tensors.
        Asynchronous on `stream` (default: torch's current stream).  With `state` (see new_state) the solve is warm-started
        from it and updates it in place; `max_iter` > 0 caps the Newton steps of this call (real-time iteration)."""
        import torch
        self._check_io(p, x0, state)
        B = p.shape[0]
        o = out if out is not None else {}
        dev = p.device

        def buf(name, shape, dtype):
            if name not in o or o[name] is None:
                o[name] = torch.empty(shape, dtype=dtype, device=dev)
            return o[name]
        x = buf("x", (B, self.n_w), torch.float64)
        ptr = {k: None for k in ("g", "lam_g", "lam_x", "f", "iters", "status", "kkt")}
        if "g" in want: ptr["g"] = buf("g", (B, self.n_g), torch.float64)
        if "lam_g" in want: ptr["lam_g"] = buf("lam_g", (B, self.n_g), torch.float64)
        if "lam_x" in want: ptr["lam_x"] = buf("lam_x", (B, self.n_w), torch.float64)
        if "f" in want: ptr["f"] = buf("f", (B,), torch.float64)
        if "iters" in want: ptr["iters"] = buf("iters", (B,), torch.int32)
        if "status" in want: ptr["status"] = buf("status", (B,), torch.int32)
        if "kkt" in want: ptr["kkt"] = buf("kkt", (B,), torch.float64)
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        dp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        if state is None and not max_iter:
            _lib.check(self._lib.bmpc_solve_batch(self._h, B, dp(p), dp(x0), dp(x), dp(ptr["g"]), dp(ptr["lam_g"]), dp(ptr["lam_x"]),
                                                  dp(ptr["f"]), dp(ptr["iters"]), dp(ptr["status"]), dp(ptr["kkt"]),
                                                  ctypes.c_void_p(st.cuda_stream)), "bmpc_solve_batch")
        else:
            if state is None:
                raise ValueError("max_iter per call needs a state buffer (new_state)")
            _lib.check(self._lib.bmpc_solve_batch_warm(self._h, B, dp(p), dp(x0), dp(state), int(max_iter), dp(x), dp(ptr["g"]), dp(ptr["lam_g"]),
                                                       dp(ptr["lam_x"]), dp(ptr["f"]), dp(ptr["iters"]), dp(ptr["status"]), dp(ptr["kkt"]),
                                                       ctypes.c_void_p(st.cuda_stream)), "bmpc_solve_batch_warm")
        # The launch is asynchronous: the kernel reads p / x0 / state and writes the outputs on the launch stream after this call has
        # returned.  The handle keeps them alive until its next launch (a caller that passes temporaries or drops the returned dict would
        # otherwise hand their memory back to the allocator while the kernel is still using it).
        self._inflight = (p, x0, state, o)
        return o

    def capture_step(self, p, x0, state=None, max_iter=0, want=("iters", "status", "kkt")):
        """hipGraph-captured step over FIXED buffers: returns a StepGraph whose launch() replays {queue reset, solver kernel};
        the caller refreshes p / x0 / state in place between launches and reads graph.out."""
        return StepGraph(self, p, x0, state, max_iter, want)

    # ---- host-buffer solve (numpy in/out) ----
    def solve_host(self, p, x0):
        p = np.ascontiguousarray(np.atleast_2d(p), dtype=np.float64)
        x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
        B = p.shape[0]
        if p.shape != (B, self.n_p) or x0.shape != (B, self.n_w):
            raise ValueError(f"shape mismatch: p {p.shape} x0 {x0.shape}")
        out = dict(x=np.zeros((B, self.n_w)), g=np.zeros((B, self.n_g)), lam_g=np.zeros((B, self.n_g)), lam_x=np.zeros((B, self.n_w)),
                   f=np.zeros(B), iters=np.zeros(B, dtype=np.int32), status=np.zeros(B, dtype=np.int32), kkt=np.zeros(B))
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(self._lib.bmpc_solve_batch_host(self._h, B, vp(p), vp(x0), vp(out["x"]), vp(out["g"]), vp(out["lam_g"]),
                                                   vp(out["lam_x"]), vp(out["f"]), vp(out["iters"]), vp(out["status"]), vp(out["kkt"])),
                   "bmpc_solve_batch_host")
        return out


class StepGraph:
    """One solver step captured into a hipGraph (bmpc_graph_create / _launch / _destroy)."""

    def __init__(self, solver, p, x0, state, max_iter, want):
        import torch
        B = solver._check_io(p, x0, state)
        if not (p.is_contiguous() and x0.is_contiguous()):
            raise ValueError("capture needs contiguous p and x0 (their addresses are baked into the graph)")
        self._solver, self.p, self.x0, self.state = solver, p, x0, state
        dev = p.device
        shapes = dict(g=((B, solver.n_g), torch.float64), lam_g=((B, solver.n_g), torch.float64), lam_x=((B, solver.n_w), torch.float64),
                      f=((B,), torch.float64), iters=((B,), torch.int32), status=((B,), torch.int32), kkt=((B,), torch.float64))
        self.out = {"x": torch.empty((B, solver.n_w), dtype=torch.float64, device=dev)}
        for k in want:
            self.out[k] = torch.empty(shapes[k][0], dtype=shapes[k][1], device=dev)
        dp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        g = lambda k: dp(self.out.get(k))
        self._g = ctypes.c_void_p()
        _lib.check(solver._lib.bmpc_graph_create(solver._h, B, dp(p), dp(x0), dp(state), int(max_iter), dp(self.out["x"]), g("g"), g("lam_g"),
                                                 g("lam_x"), g("f"), g("iters"), g("status"), g("kkt"), ctypes.byref(self._g)), "bmpc_graph_create")
        solver._children.add(self)

    def launch(self, stream=None):
        import torch
        st = stream if stream is not None else torch.cuda.current_stream(self.p.device)
        _lib.check(self._solver._lib.bmpc_graph_launch(self._g, ctypes.c_void_p(st.cuda_stream)), "bmpc_graph_launch")
        return self.out

    def close(self):
        if getattr(self, "_g", None):
            self._solver._lib.bmpc_graph_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_STATUS = {0: "Solve_Succeeded", 1: "Maximum_Iterations_Exceeded", 2: "Infeasible_Problem_Detected", 3: "Error_In_Step_Computation"}


class NlpSolverShim:
    """Stands where `ca.nlpsol('solver','ipopt',prob,opts)` stands in the reference
    (casadi_ocp_formulation.py:389): `sol = solver(x0=, lbx=, ubx=, lbg=, ubg=, p=)` returns a
    dict with 'x','f','g','lam_x','lam_g' (column vectors like CasADi DMs); `stats()` returns
    'iter_count', 'success', 'return_status' (BoundMPC.py:446-474)."""

    def __init__(self, batched: BatchedOCPSolver):
        self._s = batched
        # behind the reference's BoundMPC every x0 is the reference's own (its cold start or a shifted plan, BoundMPC.py:316-375): taken as given, like Ipopt does
        # (a setting of the caller's handle, kept while the shim lives: close() puts the previous value back)
        self._rollout_was = batched.get_start_rollout() if hasattr(batched._lib, "bmpc_get_start_rollout") else None
        if self._rollout_was is not None:
            batched.set_start_rollout(False)
        self._stats = {"iter_count": 0, "success": False, "return_status": "not run"}
        self._lbx, self._ubx, self._lbg, self._ubg = batched.bounds()

    def close(self):
        """Gives the batched solver back as it was found (the start-rollout setting); the handle itself stays the caller's."""
        if getattr(self, "_rollout_was", None) is not None and getattr(self._s, "_h", None):
            self._s.set_start_rollout(self._rollout_was)
        self._rollout_was = None

    def generate_dependencies(self, *a, **k):   # BoundMPC.py:155-157 -- nothing to generate
        return None

    def __call__(self, x0=None, lbx=None, ubx=None, lbg=None, ubg=None, p=None, lam_x0=None, lam_g0=None):
        # the bound vectors are structural constants of the formulation; refuse silently different ones
        for given, mine, nm in ((lbx, self._lbx, "lbx"), (ubx, self._ubx, "ubx"), (lbg, self._lbg, "lbg"), (ubg, self._ubg, "ubg")):
            if given is not None and not np.array_equal(np.asarray(given, dtype=float).ravel(), mine):
                raise ValueError(f"{nm} differs from the formulation's structural bounds (casadi_ocp_formulation.py:92-153,272-349)")
        out = self._s.solve_host(np.asarray(p, dtype=float).ravel()[None, :], np.asarray(x0, dtype=float).ravel()[None, :])
        st = int(out["status"][0])
        self._stats = {"iter_count": int(out["iters"][0]), "success": st == 0, "return_status": _STATUS.get(st, f"status_{st}"),
                       "kkt_error": float(out["kkt"][0])}
        col = lambda a: np.asarray(a[0]).reshape(-1, 1)
        return {"x": col(out["x"]), "f": float(out["f"][0]), "g": col(out["g"]), "lam_x": col(out["lam_x"]), "lam_g": col(out["lam_g"])}

    def stats(self):
        return dict(self._stats)
