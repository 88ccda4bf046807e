"""TEST-ONLY: ctypes binding of the CPU lane emulator of the wave program (tests/emu/bmpc_emu.cpp).
Not part of the product; see the header of bmpc_emu.cpp."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libbmpc_emu.so")
_SRC = [os.path.join(_HERE, "bmpc_emu.cpp"), os.path.join(_HERE, "..", "..", "boundmpc_amd", "csrc", "bmpc_wave.inl"),
        os.path.join(_HERE, "..", "..", "boundmpc_amd", "csrc", "bmpc_stream.inl")]


class Opts(ctypes.Structure):
    _fields_ = [("tol", ctypes.c_double), ("max_iter", ctypes.c_int), ("mu_init", ctypes.c_double),
                ("mu_min_fac", ctypes.c_double), ("slack_push", ctypes.c_double),
                ("exact_hessian", ctypes.c_int), ("verbose", ctypes.c_int), ("mu_warm", ctypes.c_double), ("stall_window", ctypes.c_int),
                ("bound_margin", ctypes.c_double), ("restoration", ctypes.c_int), ("resto_short", ctypes.c_int), ("resto_cap", ctypes.c_int), ("start_rollout", ctypes.c_int), ("hold_mu", ctypes.c_int), ("retry_cap", ctypes.c_int)]


def build(force=False):
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(_LIB) < os.path.getmtime(s) for s in _SRC):
        subprocess.check_call(["g++", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=c++17", "-Wno-unknown-pragmas", "-Wno-enum-compare",
                               "-o", _LIB, _SRC[0]])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
    return _lib


def default_opts(**kw):
    o = Opts(1e-8, 500, 0.1, 0.1, 1e-2, 1, 0, 1e-2, 40, 0.0, 1, 6, 40, 1, 0, 0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def solve(p, x0, N, S, h, opts=None, lane_order=0, nthreads=0, state=None):
    p = np.ascontiguousarray(np.atleast_2d(p), dtype=np.float64)
    x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
    B = p.shape[0]
    out = dict(x=np.zeros((B, N * 44)), g=np.zeros((B, N * 43)), lam_g=np.zeros((B, N * 43)), lam_x=np.zeros((B, N * 44)),
               f=np.zeros(B), iters=np.zeros(B, dtype=np.int32), status=np.zeros(B, dtype=np.int32), kkt=np.zeros(B))
    o = opts if opts is not None else (default_opts(mu_init=3.0, slack_push=0.1, stall_window=20, restoration=2, retry_cap=100) if N > 11 else default_opts())
    rc = lib().bmpc_emu_solve(ctypes.c_int(N), ctypes.c_int(S), ctypes.c_double(h), ctypes.byref(o), ctypes.c_int(B), _p(p), _p(x0), _p(state) if state is not None else None,
                              _p(out["x"]), _p(out["g"]), _p(out["lam_g"]), _p(out["lam_x"]), _p(out["f"]), _p(out["iters"]),
                              _p(out["status"]), _p(out["kkt"]), ctypes.c_int(lane_order), ctypes.c_int(nthreads))
    assert rc == 0
    return out


# ---- team variant of the wave program (tests/emu/bmpc_emu_team.cpp: NW cooperating waves per problem) ----
_TLIBS = {}


def team_lib(nw=4):
    """nw = 4 / 2: teams (workspace rows in LDS); nw = "pair": the two-wave team with the workspace in the global slab (BMPC_WSG, csrc/bmpc_pair.hip)"""
    if nw not in _TLIBS:
        pair = nw == "pair"
        path = os.path.join(_HERE, "libbmpc_emu_pair.so" if pair else f"libbmpc_emu_team{nw}.so")
        src = [os.path.join(_HERE, "bmpc_emu_team.cpp"), _SRC[1]]
        if not os.path.exists(path) or any(os.path.getmtime(path) < os.path.getmtime(s) for s in src):
            subprocess.check_call(["g++", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=c++17", "-Wno-unknown-pragmas", "-Wno-enum-compare"]
                                  + (["-DBMPC_NW=2", "-DBMPC_WSG"] if pair else [f"-DBMPC_NW={nw}"]) + ["-o", path, src[0]])
        _TLIBS[nw] = ctypes.CDLL(path)
        assert _TLIBS[nw].bmpc_emu_team_waves() == (2 if pair else nw)
    return _TLIBS[nw]


def solve_team(p, x0, N, S, h, nw=4, opts=None, lane_order=0, wave_order=0, nthreads=0, state=None):
    """The team program (nw waves per problem) on the CPU: wide phases run wave after wave in `wave_order` (0 forward, 1 reverse,
    2 scrambled), lanes in `lane_order`."""
    p = np.ascontiguousarray(np.atleast_2d(p), dtype=np.float64)
    x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
    B = p.shape[0]
    out = dict(x=np.zeros((B, N * 44)), g=np.zeros((B, N * 43)), lam_g=np.zeros((B, N * 43)), lam_x=np.zeros((B, N * 44)),
               f=np.zeros(B), iters=np.zeros(B, dtype=np.int32), status=np.zeros(B, dtype=np.int32), kkt=np.zeros(B))
    o = opts if opts is not None else (default_opts(mu_init=3.0, slack_push=0.1, stall_window=20, restoration=2, retry_cap=100) if N > 11 else default_opts())
    rc = team_lib(nw).bmpc_emu_team_solve(ctypes.c_int(N), ctypes.c_int(S), ctypes.c_double(h), ctypes.byref(o), ctypes.c_int(B), _p(p), _p(x0),
                                          _p(state) if state is not None else None, _p(out["x"]), _p(out["g"]), _p(out["lam_g"]), _p(out["lam_x"]),
                                          _p(out["f"]), _p(out["iters"]), _p(out["status"]), _p(out["kkt"]), ctypes.c_int(lane_order),
                                          ctypes.c_int(wave_order), ctypes.c_int(nthreads))
    assert rc == 0
    return out


# ---- CPU build of the stream functions (boundmpc_amd/csrc/bmpc_stream.inl) ----
def stream_lengths(N):
    out = (ctypes.c_int * 4)()
    lib().bmpc_emu_stream_lengths(ctypes.c_int(N), out)
    return dict(path_entry=out[0], state=out[1], robot=out[2], traj=out[3])


def stream_pack(N, S, path, ss, rb, dual=None, xlast=None, level_rule=(0.0, 0.0, 0.0)):
    """path [M][48], ss (updated in place), rb -> (p, x0); xlast: the solver's previous iterate (real-time continuation, bmpc_stream_pack_rt)"""
    p = np.zeros(141 + 91 * S); x0 = np.zeros(44 * N)
    assert path.flags.c_contiguous and ss.flags.c_contiguous and rb.flags.c_contiguous
    lib().bmpc_emu_stream_pack(ctypes.c_int(N), ctypes.c_int(S), _p(path), _p(ss), _p(rb), _p(p), _p(x0), _p(dual) if dual is not None else None,
                               _p(np.ascontiguousarray(xlast, dtype=np.float64)) if xlast is not None else None, ctypes.c_double(level_rule[0]), ctypes.c_double(level_rule[1]), ctypes.c_double(level_rule[2]))
    return p, x0


def stream_post(N, S, h, path, ss, rb, x, g, status, simulate=True, rt_tol=1e-4, flags=0, rt_row_cap=0.0):
    """flags: extra bits of the post flags (bit 1 = real-time acceptance rule)"""
    traj = np.zeros(stream_lengths(N)["traj"])
    x = np.ascontiguousarray(x, dtype=np.float64); g = np.ascontiguousarray(g, dtype=np.float64)
    lib().bmpc_emu_stream_post(ctypes.c_int(N), ctypes.c_int(S), ctypes.c_double(h), _p(path), _p(ss), _p(rb), _p(x), _p(g), ctypes.c_int(int(status)),
                               _p(traj), ctypes.c_int(int(simulate) | int(flags)), ctypes.c_double(rt_tol), ctypes.c_double(rt_row_cap))
    return traj


def jacobian_lin_ddot(q, dq, ddq):
    """Second time derivative of the linear rows of the Jacobian (csrc/bmpc_stream.inl) -> [3][7]."""
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (q, dq, ddq)]
    out = np.zeros((3, 7))
    lib().bmpc_emu_jacobian_lin_ddot(_p(a[0]), _p(a[1]), _p(a[2]), _p(out))
    return out


def fk_motion(q, dq, ddq, u):
    """Pose, v = J dq, a = J ddq + dJ dq and the linear rows of J u + dJ ddq + ddJ dq in one pass over the chain (csrc/bmpc_stream.inl fk_motion)."""
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (q, dq, ddq, u)]
    out = np.zeros(21)
    lib().bmpc_emu_fk_motion(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(out))
    return out[:6], out[6:12], out[12:18], out[18:21]


# ---- flop-counting build of the same kernel text (tests/emu/bmpc_emu_flops.cpp) ----
_FLIB = os.path.join(_HERE, "libbmpc_emu_flops.so")
_fl = None


def count_flops(p, x0, N, S, h, opts=None):
    """fp64 operations the kernel text executes (summed over the lanes of every phase) while solving the batch: dict with iterations,
    converged, flops, special, flops_per_iteration, per_phase (slot id of tests/gpu_profile_phases.py -> flops)."""
    global _fl
    src = os.path.join(_HERE, "bmpc_emu_flops.cpp")
    if not os.path.exists(_FLIB) or any(os.path.getmtime(_FLIB) < os.path.getmtime(s) for s in (src, _SRC[1])):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wno-unknown-pragmas", "-Wno-enum-compare", "-Wno-format", "-o", _FLIB, src])
        _fl = None
    if _fl is None:
        _fl = ctypes.CDLL(_FLIB)
    p = np.ascontiguousarray(np.atleast_2d(p), dtype=np.float64)
    x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
    o = opts if opts is not None else (default_opts(mu_init=3.0, slack_push=0.1, stall_window=20, restoration=2, retry_cap=100) if N > 11 else default_opts())
    out = np.zeros(36, dtype=np.uint64)
    rc = _fl.bmpc_emu_count_flops(ctypes.c_int(N), ctypes.c_int(S), ctypes.c_double(h), ctypes.byref(o), ctypes.c_int(p.shape[0]), _p(p), _p(x0), _p(out))
    assert rc == 0
    its = int(out[0])
    return dict(iterations=its, converged=int(out[1]), flops=int(out[2]), special=int(out[3]), flops_per_iteration=float(out[2]) / max(its, 1),
                per_phase={i: int(out[4 + i]) for i in range(32) if out[4 + i]})


# ---- mask-aware flop count of the same kernel text (tests/emu/bmpc_emu_useful.cpp): executed vs useful operations by data flow ----
_ULIB = os.path.join(_HERE, "libbmpc_emu_useful.so")
_ul = None


def count_useful(p, x0, N, S, h, opts=None):
    """ONE problem: fp64 operations the kernel text executes and, of those, the ones whose result reaches a store (not the dummy word, not a
    clamped duplicate within the phase) or a decision.  dict: iterations, executed, useful, stores, duplicate_stores, dummy_stores,
    per_phase {slot: (executed, useful)}."""
    global _ul
    src = os.path.join(_HERE, "bmpc_emu_useful.cpp")
    if not os.path.exists(_ULIB) or any(os.path.getmtime(_ULIB) < os.path.getmtime(s) for s in (src, _SRC[1])):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wno-unknown-pragmas", "-Wno-enum-compare", "-Wno-format", "-o", _ULIB, src])
        _ul = None
    if _ul is None:
        _ul = ctypes.CDLL(_ULIB)
    p = np.ascontiguousarray(np.asarray(p, dtype=np.float64).ravel()); x0 = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).ravel())
    o = opts if opts is not None else (default_opts(mu_init=3.0, slack_push=0.1, stall_window=20, restoration=2, retry_cap=100) if N > 11 else default_opts())
    out = np.zeros(72, dtype=np.uint64)
    rc = _ul.bmpc_emu_count_useful(ctypes.c_int(N), ctypes.c_int(S), ctypes.c_double(h), ctypes.byref(o), _p(p), _p(x0), _p(out))
    assert rc == 0
    return dict(iterations=int(out[0]), converged=int(out[1]), executed=int(out[2]), useful=int(out[3]), stores=int(out[4]), duplicate_stores=int(out[5]),
                dummy_stores=int(out[6]), per_phase={s_: (int(out[8 + 2 * s_]), int(out[9 + 2 * s_])) for s_ in range(32) if out[8 + 2 * s_]})

