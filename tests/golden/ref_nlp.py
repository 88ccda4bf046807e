"""The reference's OWN NLP, made callable: f(x, p), g(x, p) and the bound vectors obtained by running the
reference's unmodified `setup_optimization_problem` (BoundMPC/casadi_ocp_formulation.py:9-391, with the
stage functions of bound_mpc_functions.py, mpc_utils_casadi.py, jerk_trajectory_casadi.py and the Maple
kinematics of RobotModel.py) over the numeric SX stand-in of `numeric_sx.py`.

Fixture generation only -- build container only (/root/reference does not exist on the GPU box);
nothing under boundmpc_amd/ or tests/test_*.py imports this.

How (x, p) reach the symbols without assuming any layout: a first run of the builder gives every element
of every `SX.sym` a unique code number; the codes found in the `prob['x']` and `prob['p']` the reference
hands to `ca.nlpsol` tell which symbol element sits at which offset.  A later evaluation at (x, p) scatters
the numbers into the symbols through that table and re-runs the builder.
"""
import os
import sys
import tempfile
import textwrap

import numpy as np

REF = "/root/reference/bound_mpc"
HERE = os.path.dirname(os.path.abspath(__file__))
_installed = []


def install_standins():
    """Temp-dir modules for the three absent third-party packages (never written into the repo):
    `casadi` -> numeric_sx (see its header), `sensor_msgs` / `bound_mpc_msg` -> empty message classes
    (imported by utils/util_functions.py:3-5, never used on the path)."""
    if _installed:
        return _installed[0]
    d = tempfile.mkdtemp(prefix="bmpc_standin_")
    with open(os.path.join(d, "casadi.py"), "w") as f:
        f.write(textwrap.dedent(f'''
            import sys
            sys.path.insert(0, {HERE!r})
            from numeric_sx import *          # noqa
            from numeric_sx import SX, MX, DM, nlpsol
        '''))
    for pkg, sub, names in (("sensor_msgs", "msg", ["JointState"]),
                            ("bound_mpc_msg", "msg", ["Vector", "MPCData"])):
        os.makedirs(os.path.join(d, pkg, sub))
        open(os.path.join(d, pkg, "__init__.py"), "w").close()
        with open(os.path.join(d, pkg, sub, "__init__.py"), "w") as f:
            for n in names:
                f.write(f"class {n}:\n    pass\n")
    os.makedirs(os.path.join(d, "bound_mpc_msg", "srv"))
    with open(os.path.join(d, "bound_mpc_msg", "srv", "__init__.py"), "w") as f:
        f.write("class Trajectory:\n    pass\nclass MPCParams:\n    pass\n")
    with open(os.path.join(d, "bound_mpc_msg", "srv", "_trajectory.py"), "w") as f:
        f.write("class Trajectory_Request:\n    pass\n")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, d)
    _installed.append(d)
    return d


class RefNlp:
    """Callable view of the reference's NLP for horizon N, S path segments and step dt."""

    def __init__(self, N, S, dt):
        install_standins()
        import casadi                                   # the stand-in
        from bound_mpc.BoundMPC.casadi_ocp_formulation import setup_optimization_problem
        from bound_mpc.RobotModel import RobotModel
        self._SX = casadi.SX
        self._setup = setup_optimization_problem
        self.N, self.S, self.dt = N, S, dt
        lim = RobotModel().get_robot_limits()           # as BoundMPC.__init__ does (BoundMPC.py:86-96)
        self._args = dict(u_min=lim[7], u_max=lim[6], ut_min=lim[7], ut_max=lim[6], q_lim_lower=lim[1],
                          q_lim_upper=lim[0], dq_lim_lower=lim[3], dq_lim_upper=lim[2])
        self._learn_layout()

    def _run(self, provider):
        SX = self._SX
        SX.provider, SX.created = provider, []
        a = self._args
        with np.errstate(all="ignore"):
            out = self._setup(self.N, 7, self.S, self.dt, a["u_min"], a["u_max"], a["ut_min"], a["ut_max"],
                              list(a["q_lim_lower"]), list(a["q_lim_upper"]), list(a["dq_lim_lower"]),
                              list(a["dq_lim_upper"]), {"ipopt": {}})
        created = SX.created
        SX.provider, SX.created = None, []
        return out, created

    def _learn_layout(self):
        code = {}
        nxt = [1000.0]

        def provider(idx, name, n, m):
            v = np.zeros((n, m))
            for r in range(n):
                for c in range(m):
                    v[r, c] = nxt[0]
                    code[nxt[0]] = (idx, r, c)
                    nxt[0] += 1.0
            return v
        (nlp, lbx, ubx, lbg, ubg, g_names), created = self._run(provider)
        self.sym = [(name, n, m) for name, n, m, _ in created]
        self.x_map = [code[float(v)] for v in nlp.x]
        self.p_map = [code[float(v)] for v in nlp.p]
        seen = self.x_map + self.p_map
        assert len(set(seen)) == len(seen) == len(code), "every symbol element must appear exactly once in x or p"
        self.n_x, self.n_p, self.n_g = len(self.x_map), len(self.p_map), len(nlp.g)
        self.lbx, self.ubx = np.array(lbx, dtype=float), np.array(ubx, dtype=float)
        self.lbg, self.ubg = np.array(lbg, dtype=float), np.array(ubg, dtype=float)
        self.g_names = list(g_names)

    def names(self, which="p"):
        """Reference symbol name and element of every entry of p (or x): 'name[r,c]'."""
        return ["%s[%d,%d]" % (self.sym[i][0], r, c) for i, r, c in (self.p_map if which == "p" else self.x_map)]

    def __call__(self, x, p):
        """-> (f, g) of the reference at (x, p); complex inputs give a complex step through the reference's code."""
        x, p = np.asarray(x), np.asarray(p)
        assert x.shape == (self.n_x,) and p.shape == (self.n_p,)
        dt = np.result_type(x.dtype, p.dtype, float)
        vals = [np.zeros((n, m), dtype=dt) for _, n, m in self.sym]
        for v, (i, r, c) in zip(x, self.x_map):
            vals[i][r, c] = v
        for v, (i, r, c) in zip(p, self.p_map):
            vals[i][r, c] = v
        (nlp, *_), created = self._run(lambda idx, name, n, m: vals[idx])
        assert [(n_, a, b) for n_, a, b, _ in created] == self.sym
        assert np.array_equal(nlp.x, x) and np.array_equal(nlp.p, p)
        f, g = nlp.f, nlp.g
        if dt.kind != "c":
            f, g = np.real(f), np.real(g)
        return f[()], g

    def grad_jac(self, x, p, eps=1e-30):
        """Exact (complex-step) gradient of f and Jacobian of g w.r.t. x, through the reference's own code."""
        n = self.n_x
        gf = np.zeros(n)
        Jg = np.zeros((self.n_g, n))
        xc = np.asarray(x).astype(complex)
        for i in range(n):
            xc[i] += 1j * eps
            f, g = self(xc, np.asarray(p))
            gf[i] = f.imag / eps
            Jg[:, i] = g.imag / eps
            xc[i] = x[i]
        return gf, Jg
