"""ORACLE (test infrastructure only): ctypes binding of oracle/bmpc_oracle.c.

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product path (boundmpc_amd/).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libbmpc_oracle.so")
NZ, NG, NE, NI = 44, 43, 36, 57


class Opts(ctypes.Structure):
    _fields_ = [("tol", ctypes.c_double), ("max_iter", ctypes.c_int), ("mu_init", ctypes.c_double),
                ("mu_min_fac", ctypes.c_double), ("slack_push", ctypes.c_double),
                ("exact_hessian", ctypes.c_int), ("verbose", ctypes.c_int), ("mu_warm", ctypes.c_double), ("stall_window", ctypes.c_int),
                ("restoration", ctypes.c_int), ("resto_short", ctypes.c_int), ("resto_cap", ctypes.c_int), ("start_rollout", ctypes.c_int), ("hold_mu", ctypes.c_int), ("retry_cap", ctypes.c_int)]


def build(force=False):
    src = os.path.join(_HERE, "bmpc_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libbmpc_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
        _lib.bmpc_oracle_default_opts.argtypes = [ctypes.POINTER(Opts)]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def default_opts(**kw):
    o = Opts()
    lib().bmpc_oracle_default_opts(ctypes.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def eval_fg(p, x, N, S, h):
    p = np.ascontiguousarray(p, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    f = ctypes.c_double()
    g = np.zeros(N * NG)
    lib().bmpc_oracle_eval(ctypes.c_int(N), ctypes.c_int(S), ctypes.c_double(h), _p(p), _p(x), ctypes.byref(f), _p(g))
    return f.value, g


def adjoint(p, x, nu, N, S, h):
    p = np.ascontiguousarray(p, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    lam = np.zeros(N * NE)
    rj = np.zeros(N * 8)
    gz = np.zeros(N * NZ)
    lib().bmpc_oracle_adjoint(ctypes.c_int(N), ctypes.c_int(S), ctypes.c_double(h), _p(p), _p(x), _p(nu), _p(lam), _p(rj), _p(gz))
    return lam, rj, gz


def kin(q, dq, mu_p, mu_v, mu_w):
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (q, dq, mu_p, mu_v, mu_w)]
    pos = np.zeros(3); v = np.zeros(6); J = np.zeros((6, 7)); D = np.zeros((6, 7)); W = np.zeros((14, 14))
    lib().bmpc_oracle_kin(*[_p(v_) for v_ in a], _p(pos), _p(v), _p(J), _p(D), _p(W))
    return pos, v, J, D, W


def newton_dir(p, x, t, nu, mu, N, S, h, exact=1, delta=0.0):
    arrs = [np.ascontiguousarray(v, dtype=np.float64) for v in (p, x, t, nu)]
    dZ = np.zeros(N * NZ)
    rc = lib().bmpc_oracle_newton_dir(ctypes.c_int(N), ctypes.c_int(S), ctypes.c_double(h), *[_p(v) for v in arrs],
                                      ctypes.c_double(mu), ctypes.c_int(exact), ctypes.c_double(delta), _p(dZ))
    return rc, dZ


def state_len(N):
    return N * 57 + 2


def solve(p, x0, N, S, h, opts=None, nthreads=0, state=None):
    """Batched CPU solve.  p [B][n_p], x0 [B][44N] -> dict of outputs.  state: None or float64 [B][state_len(N)],
    dual state of a receding-horizon stream, read (warm start where its mu entry > 0) and updated in place."""
    p = np.ascontiguousarray(np.atleast_2d(p), dtype=np.float64)
    x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
    B = p.shape[0]
    assert p.shape[1] == 141 + 91 * S and x0.shape == (B, N * NZ)
    out = dict(x=np.zeros((B, N * NZ)), g=np.zeros((B, N * NG)), lam_g=np.zeros((B, N * NG)),
               lam_x=np.zeros((B, N * NZ)), f=np.zeros(B), iters=np.zeros(B, dtype=np.int32),
               status=np.zeros(B, dtype=np.int32), kkt=np.zeros(B))
    o = opts if opts is not None else (default_opts(mu_init=3.0, slack_push=0.1, stall_window=20, restoration=2, retry_cap=100) if N > 11 else default_opts())     # horizon rule of bmpc_default_options_for
    if state is not None:
        assert state.dtype == np.float64 and state.flags.c_contiguous and state.shape == (B, state_len(N))
    lib().bmpc_oracle_solve_warm(ctypes.c_int(N), ctypes.c_int(S), ctypes.c_double(h), ctypes.byref(o), ctypes.c_int(B),
                            _p(p), _p(x0), _p(state) if state is not None else None, _p(out["x"]), _p(out["g"]), _p(out["lam_g"]), _p(out["lam_x"]),
                            _p(out["f"]), _p(out["iters"]), _p(out["status"]), _p(out["kkt"]), ctypes.c_int(nthreads))
    return out


# ---- flop-counting build (oracle/flopcount.cpp): the same source text with a counting number type, one thread ----
_FLOPS_LIB = os.path.join(_HERE, "libbmpc_oracle_flops.so")
FLOP_REGIONS = ("driver (row passes, line-search bookkeeping)", "eval (kinematics, references, residuals)", "adjoint + gradients",
                "build_qp (node Hessians, stage maps)", "riccati (factorise + forward)", "kkt errors", "outputs")
_flib = None


def build_flops(force=False):
    srcs = [os.path.join(_HERE, "flopcount.cpp"), os.path.join(_HERE, "bmpc_oracle.c")]
    if force or not os.path.exists(_FLOPS_LIB) or os.path.getmtime(_FLOPS_LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libbmpc_oracle_flops.so"])
    return _FLOPS_LIB


def count_flops(p, x0, N, S, h, opts=None):
    """Executed fp64 operations of the oracle's solve of the batch (p, x0), one thread.  Returns a dict: iterations (total
    interior-point iterations), converged, flops (total), special (divisions, roots, transcendentals among them), per_region
    [(name, flops, special)], flops_per_iteration.  Convention: oracle/flopcount.cpp header."""
    global _flib
    if _flib is None:
        if not os.path.exists(_FLOPS_LIB):
            build_flops()
        _flib = ctypes.CDLL(_FLOPS_LIB)
    p = np.ascontiguousarray(np.atleast_2d(p), dtype=np.float64)
    x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
    B = p.shape[0]
    assert p.shape[1] == 141 + 91 * S and x0.shape == (B, N * NZ)
    o = opts if opts is not None else (default_opts(mu_init=3.0, slack_push=0.1, stall_window=20, restoration=2, retry_cap=100) if N > 11 else default_opts())
    out = np.zeros(18, dtype=np.uint64)
    rc = _flib.bmpc_oracle_count_flops(ctypes.c_int(N), ctypes.c_int(S), ctypes.c_double(h), ctypes.byref(o), ctypes.c_int(B), _p(p), _p(x0), _p(out))
    assert rc == 0
    fl, sp = out[2:10].astype(np.int64), out[10:18].astype(np.int64)
    its = int(out[0])
    return dict(iterations=its, converged=int(out[1]), flops=int(fl.sum()), special=int(sp.sum()),
                per_region=[(FLOP_REGIONS[r], int(fl[r]), int(sp[r])) for r in range(len(FLOP_REGIONS))],
                flops_per_iteration=float(fl.sum()) / max(its, 1))
