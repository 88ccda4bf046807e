"""ORACLE (test infrastructure only -- never imported by the product path).

Plain-numpy restatement of the BoundMPC per-step NLP: objective f(x, p), constraints
g(x, p), the bound vectors and the 505-parameter layout, following the reference line by
line (paths relative to /root/reference/bound_mpc/bound_mpc):

  BoundMPC/casadi_ocp_formulation.py:9-391   structure, ordering, bounds, parameter vector
  BoundMPC/bound_mpc_functions.py:13-310     segment select, reference, errors, objective,
                                             integrator, plane decomposition
  BoundMPC/mpc_utils_casadi.py:6-165         rotation-error linearisation, position error,
                                             quartic tubes
  BoundMPC/jerk_trajectory_casadi.py:78-175  hat-function jerk integrator (closed form below)
  RobotModel/RobotModel.py:7-100,1055-1107,1270-1303  iiwa14 fk_pos / velocity_ee / omega_ee,
                                             restated as a geometric chain (pinned by fixture G1)

Parity status: the numeric leaves are pinned against golden vectors produced by the
reference's own numeric code (tests/golden/g1..g6, tests/test_oracle_golden.py).  The NLP
*solution* is "parity unpinned" against Ipopt: casadi/Ipopt/MUMPS are absent from the image
and the reference holds no solution fixtures; solutions are certified by
KKT residuals and by an independent scipy solve instead (oracle/solve_scipy.py).

All functions are dtype-generic (float64 or complex128) so that complex-step
differentiation gives machine-precision derivatives for checking analytic Jacobians.
"""
import numpy as np

NJ = 7          # joints
NU = 8          # 7 joint jerks + path jerk
NZ = 44         # stage variables  (casadi_ocp_formulation.py:90-153)
NG = 43         # stage constraints (casadi_ocp_formulation.py:272-349)

# stage-variable offsets, z_k = [u(7) uphi | q dq ddq | p(3) iw(3) | v(3) w(3) | phi dphi ddphi]
IU, IUPHI, IQ, IDQ, IDDQ, IP, IIW, IV, IW, IPHI, IDPHI, IDDPHI = 0, 7, 8, 15, 22, 29, 32, 35, 38, 41, 42, 43

# RobotModel.py:9-16
D1, D2, D3, D4, D5, D6, D7, D8 = 0.1575, 0.2025, 0.2375, 0.1825, 0.2175, 0.1825, 0.081, 0.071 + 0.145
# RobotModel.py:20-43
Q_LIM = np.array([165, 115, 165, 115, 165, 115, 170]) * np.pi / 180
DQ_LIM = np.array([85, 85, 100, 75, 130, 135, 135]) * np.pi / 180
U_LIM = 35.0


# --------------------------------------------------------------------------------------
# parameter vector  (casadi_ocp_formulation.py:361-376; BoundMPC.py:416-443)
# --------------------------------------------------------------------------------------
def p_layout(S):
    """name -> (offset, shape) with CasADi column-major flattening of each block."""
    items = [("q0", (7,)), ("dq0", (7,)), ("ddq0", (7,)), ("phi0", (1,)), ("dphi0", (1,)), ("ddphi0", (1,)),
             ("p0", (6,)), ("v0", (6,)), ("iw_ref0", (3,)), ("dtau_init", (3,)),
             ("dtau_init_par", (S, 3)), ("dtau_init_orth1", (S, 3)), ("dtau_init_orth2", (S, 3)),
             # ^ CasADi (3,S) column-major == numpy [seg][xyz]
             ("x_phi_d", (3,)), ("jerk_cur", (7,)), ("jerk_phi_cur", (1,)), ("phi_switch", (S + 1,)),
             ("jac_dtau_r_T", (3, 3)), ("jac_dtau_l_T", (3, 3)),  # (3,3) column-major == numpy transpose
             ("p_ref", (6, S)), ("dp_ref", (6, S)), ("dp_normed_ref", (3, S)),  # (S,6) col-major == [coord][seg]
             ("bp1", (3, S)), ("bp2", (3, S)), ("br1", (3, S)), ("br2", (3, S)),
             ("a4", (9, S + 1)), ("a3", (9, S + 1)), ("a2", (9, S + 1)), ("a1", (9, S + 1)), ("a0", (9, S + 1)),
             ("weights", (15,)), ("phi_max", (1,)), ("dphi_max", (1,)),
             ("v1", (3, S)), ("v2", (3, S)), ("v3", (3, S)), ("qd", (7,))]
    lay = {}
    off = 0
    for name, shp in items:
        lay[name] = (off, shp)
        off += int(np.prod(shp))
    lay["_size"] = off
    return lay


def n_p(S):
    return 141 + 91 * S


def unpack_p(p, S):
    lay = p_layout(S)
    assert p.shape[-1] == lay["_size"] == n_p(S)
    out = {}
    for name, val in lay.items():
        if name == "_size":
            continue
        off, shp = val
        out[name] = p[off:off + int(np.prod(shp))].reshape(shp)
    out["jac_dtau_r"] = out.pop("jac_dtau_r_T").T
    out["jac_dtau_l"] = out.pop("jac_dtau_l_T").T
    return out


# --------------------------------------------------------------------------------------
# iiwa14 kinematics as a geometric chain (pinned by G1 against RobotModel.fk_pos etc.)
# --------------------------------------------------------------------------------------
_AXES = np.array([[0, 0, 1], [0, 1, 0], [0, 0, 1], [0, -1, 0], [0, 0, 1], [0, 1, 0], [0, 0, 1]], dtype=float)
# translation along the local z axis applied BEFORE joint j (0-based)
_PRE_Z = np.array([0.0, D1 + D2, 0.0, D3 + D4, 0.0, D5 + D6, 0.0])
_TOOL_Z = D7 + D8


def _rot(axis, ang):
    """Rodrigues rotation about a unit coordinate axis (dtype generic)."""
    c, s = np.cos(ang), np.sin(ang)
    x, y, z = axis
    K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=float)
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def chain(q):
    """-> (axes a_j [7][3], origins o_j [7][3], tool point p [3], tool rotation R [3][3]) in the base frame."""
    dt = np.result_type(q.dtype, float)
    R = np.eye(3, dtype=dt)
    o = np.zeros(3, dtype=dt)
    A = np.zeros((7, 3), dtype=dt)
    O = np.zeros((7, 3), dtype=dt)
    for j in range(7):
        o = o + R[:, 2] * _PRE_Z[j]
        A[j] = R @ _AXES[j]
        O[j] = o
        R = R @ _rot(_AXES[j], q[j])
    p = o + R[:, 2] * _TOOL_Z
    return A, O, p, R


def fk_pos(q):
    """RobotModel.fk_pos (RobotModel.py:62-100)."""
    return chain(q)[2]


def jacobian(q):
    """Geometric Jacobian [v; w] at the tool point (RobotModel.jacobian_fk, RobotModel.py:254-373)."""
    A, O, p, _ = chain(q)
    J = np.zeros((6, 7), dtype=A.dtype)
    for j in range(7):
        J[:3, j] = np.cross(A[j], p - O[j])
        J[3:, j] = A[j]
    return J


def velocity_ee(q, dq):
    """RobotModel.velocity_ee (RobotModel.py:1055-1107) = J_v(q) dq."""
    return jacobian(q)[:3] @ dq


def omega_ee(q, dq):
    """RobotModel.omega_ee (RobotModel.py:1270-1303) = J_w(q) dq."""
    return jacobian(q)[3:] @ dq


# --------------------------------------------------------------------------------------
# jerk integrator, closed form of calcAngle/Velocity/Acceleration for the 2-column matrix
# [u_prev, u] evaluated at t = h (jerk_trajectory_casadi.py:78-175; bound_mpc_functions.py:254-260)
# --------------------------------------------------------------------------------------
def integrate_chain(x, dx, ddx, u_prev, u, h):
    xn = x + h * dx + h * h / 2 * ddx + h ** 3 / 8 * u_prev + h ** 3 / 24 * u
    dxn = dx + h * ddx + h * h / 3 * u_prev + h * h / 6 * u
    ddxn = ddx + h / 2 * (u_prev + u)
    return xn, dxn, ddxn


def integrate_jerk_matrix(jm, t_idx, x, dx, ddx, h):
    """State at t = h*(t_idx+1) for an (n_cols)-column hat-function jerk matrix (node values
    jm[:, 0..]), by chaining the closed-form step (used for compute_return_data parity)."""
    for k in range(t_idx + 1):
        x, dx, ddx = integrate_chain(x, dx, ddx, jm[..., k], jm[..., k + 1], h)
    return x, dx, ddx


# --------------------------------------------------------------------------------------
# stage functions
# --------------------------------------------------------------------------------------
def seg_index(phi, phi_switch):
    """get_current_segment (bound_mpc_functions.py:13-20): first i with phi < phi_switch[i+1];
    default last row.  Returns (index into S-row arrays, index into (S+1)-row arrays)."""
    S = len(phi_switch) - 1
    ph = np.real(phi)
    sw = np.real(phi_switch)
    iS = S - 1       # arrays with S rows: loop over i = S-2..0
    for i in reversed(range(S - 1)):
        if ph < sw[i + 1]:
            iS = i
    iS1 = S          # arrays with S+1 rows: loop over i = S-1..0
    for i in reversed(range(S)):
        if ph < sw[i + 1]:
            iS1 = i
    # phi_start from get_current_and_next_segment(phi, phi_switch, phi_switch): default rows [-2:], loop S-2..0
    istart = S - 1
    for i in reversed(range(S - 1)):
        if ph < sw[i + 1]:
            istart = i
    return iS, iS1, istart


def reference_function(phi, P):
    """bound_mpc_functions.py:43-149 (numeric restatement)."""
    sw = P["phi_switch"]
    S = len(sw) - 1
    iS, iS1, ist = seg_index(phi, sw)
    # Row S of a4..a0 is np.empty garbage in the reference (BoundMPC.py:235-240) and is only
    # selected once phi >= phi_switch[S]; it is defined here (and in the build) as a copy of
    # row S-1 -- a deliberate, documented deviation.
    iS1 = min(iS1, S - 1)
    dp_d = P["dp_ref"][:, iS]
    phi_start = sw[ist]
    p_d = dp_d * (phi - phi_start) + P["p_ref"][:, iS]
    x = phi - phi_start
    b = (P["a4"][:, iS1] * x ** 4 + P["a3"][:, iS1] * x ** 3 + P["a2"][:, iS1] * x ** 2
         + P["a1"][:, iS1] * x + P["a0"][:, iS1])          # 9 channels (mpc_utils_casadi.py:163)
    p_e_bound = np.array([b[0], b[1], b[2], b[3]])          # [up0 up1 lo0 lo1]
    r_e_bound = np.array([b[4], b[5], b[6], b[7]])
    return dict(p_d=p_d, dp_d=dp_d, dp_normed_d=P["dp_normed_ref"][:, iS],
                # bp1/bp2 go through get_current_and_next_segment on an S-row array
                # (bound_mpc_functions.py:34-40,109-110): its default is rows [-2:], so the
                # last window segment S-1 is never selected -- reference behaviour, kept.
                bp1=P["bp1"][:, min(iS, S - 2)], bp2=P["bp2"][:, min(iS, S - 2)],
                br1=P["br1"][:, iS], br2=P["br2"][:, iS],
                v1=P["v1"][:, iS], v2=P["v2"][:, iS], v3=P["v3"][:, iS],
                bound_lower=np.array([p_e_bound[2], p_e_bound[3], r_e_bound[2], r_e_bound[3]]),
                bound_upper=np.array([p_e_bound[0], p_e_bound[1], r_e_bound[0], r_e_bound[1]]),
                r_par_bound=b[8],
                e_p_off=0.5 * (p_e_bound[:2] + p_e_bound[2:]),
                e_r_off=0.5 * (r_e_bound[:2] + r_e_bound[2:]),
                seg=iS)


def error_function(pk, ref, P):
    """bound_mpc_functions.py:152-202 + mpc_utils_casadi.py:6-67 (only the outputs the NLP uses)."""
    e_p = pk[:3] - ref["p_d"][:3]
    d = ref["dp_d"][:3]
    e_p_par = (d @ e_p) * d
    e_r = (P["dtau_init"] + P["jac_dtau_l"] @ (pk[3:] - P["p0"][3:])
           - P["jac_dtau_r"] @ (ref["p_d"][3:] - P["iw_ref0"]))
    seg = ref["seg"]
    dlt = e_r - P["dtau_init"]
    e_r_orth1 = P["dtau_init_orth1"][seg] + (dlt @ ref["v1"]) * ref["br1"]
    e_r_par = P["dtau_init_par"][seg] + (dlt @ ref["v2"]) * ref["dp_normed_d"]
    e_r_orth2 = P["dtau_init_orth2"][seg] + (dlt @ ref["v3"]) * ref["br2"]
    return dict(e_p=e_p, e_p_par=e_p_par, e_r=e_r, e_r_par=e_r_par, e_r_orth1=e_r_orth1, e_r_orth2=e_r_orth2)


def nlp_eval(x, p, N, S, h, stage_costs=False):
    """-> (f, g[43N]) exactly as casadi_ocp_formulation.py:88-349 builds them (stage_costs: f as the N summands)."""
    P = unpack_p(p, S)
    dt = np.result_type(x.dtype, p.dtype)
    z = x.reshape(N, NZ)
    w = P["weights"]
    f = np.zeros((), dtype=dt)
    fk = np.zeros(N, dtype=dt)
    g = np.zeros((N, NG), dtype=dt)
    qk, dqk, ddqk = P["q0"], P["dq0"], P["ddq0"]
    phik, dphik, ddphik = P["phi0"][0], P["dphi0"][0], P["ddphi0"][0]
    pk = P["p0"]
    vprev = P["v0"]
    u_prev = np.concatenate([P["jerk_cur"], P["jerk_phi_cur"]])
    for k in range(N):
        zk = z[k]
        uk = zk[0:8]
        # integration_function (bound_mpc_functions.py:249-295)
        qn, dqn, ddqn = integrate_chain(qk, dqk, ddqk, u_prev[:7], uk[:7], h)
        phin, dphin, ddphin = integrate_chain(phik, dphik, ddphik, u_prev[7], uk[7], h)
        J = jacobian(qn)
        pn_pos = fk_pos(qn)
        vn = J @ dqn
        k1 = omega_ee(qk, dqk)
        pn_rot = pk[3:] + 0.5 * h * (k1 + vn[3:])
        u_prev = uk
        # new stage variables
        qk, dqk, ddqk = zk[IQ:IQ + 7], zk[IDQ:IDQ + 7], zk[IDDQ:IDDQ + 7]
        pk_new = zk[IP:IP + 6]
        vk = zk[IV:IV + 6]
        phik_n, dphik_n, ddphik_n = zk[IPHI], zk[IDPHI], zk[IDDPHI]
        ref = reference_function(phik_n, P)
        err = error_function(pk_new, ref, P)
        ak = (vk - vprev) / h
        dp_d = ref["dp_d"]
        v_ref = dphik_n * dp_d
        a_ref = ddphik_n * dp_d
        sigm = 1 / (1 + np.exp(-100 * (phik_n - (P["phi_max"][0] - 0.02))))
        e_p_obj = sigm * err["e_p"] + (1 - sigm) * err["e_p_par"]
        e_r_obj = sigm * err["e_r"] + (1 - sigm) * err["e_r_par"]
        # objective_function (bound_mpc_functions.py:205-246)
        xd = P["x_phi_d"]
        fk[k] = (w[1] * np.sum(e_r_obj ** 2) + w[0] * np.sum(e_p_obj ** 2)
                 + w[2] * np.sum((vk - v_ref) ** 2) + w[5] * np.sum((ak - a_ref) ** 2)
                 + w[10] * np.sum((qk - P["qd"]) ** 2) + w[11] * np.sum(dqk ** 2) + w[12] * np.sum(ddqk ** 2)
                 + w[13] * np.sum(uk[:7] ** 2)
                 + w[6] * (xd[0] - phik_n) ** 2 + w[7] * (xd[1] - dphik_n) ** 2 + w[8] * (xd[2] - ddphik_n) ** 2
                 + w[9] * uk[7] ** 2)
        f = f + fk[k]
        vprev = vk
        # constraints (casadi_ocp_formulation.py:272-349)
        gk = g[k]
        gk[0:7] = qn - qk
        gk[7:14] = dqn - dqk
        gk[14:21] = ddqn - ddqk
        gk[21:24] = pn_pos - pk_new[:3]
        gk[24:27] = pn_rot - pk_new[3:]
        gk[27:33] = vn - vk
        gk[33] = phin - phik_n
        gk[34] = dphin - dphik_n
        gk[35] = ddphin - ddphik_n
        gk[36] = phik_n - P["phi_max"][0]
        gk[37] = dphik_n - P["dphi_max"][0]
        gk[38] = (ref["dp_normed_d"] @ err["e_r_par"]) ** 2 - ref["r_par_bound"] ** 2
        e_plane = np.array([err["e_p"] @ ref["bp1"], err["e_p"] @ ref["bp2"]])       # decomp_function :298-310
        e_diff = e_plane - ref["e_p_off"]
        bnd = (ref["bound_upper"][:2] - ref["bound_lower"][:2]) / 2
        gk[39] = e_diff[0] ** 2 - bnd[0] ** 2
        gk[40] = e_diff[1] ** 2 - bnd[1] ** 2
        bndr = (ref["bound_upper"][2:] - ref["bound_lower"][2:]) / 2
        gk[41] = (ref["br1"] @ err["e_r_orth1"] - ref["e_r_off"][0]) ** 2 - bndr[0] ** 2
        gk[42] = (ref["br2"] @ err["e_r_orth2"] - ref["e_r_off"][1]) ** 2 - bndr[1] ** 2
        pk = pk_new
        phik, dphik, ddphik = phik_n, dphik_n, ddphik_n
    return (fk if stage_costs else f), g.reshape(-1)


def internal_ineq(x, p, N, S):
    """The build's internal inequality rows h[N][57] <= 0 (two-sided form of the reference's
    squared tube constraints plus the simple bounds), in the row order of oracle/bmpc_oracle.c.
    Equivalent feasible set: l^2 - w^2 <= 0  <=>  -|w| <= l <= |w|."""
    P = unpack_p(p, S)
    z = x.reshape(N, NZ)
    dt = np.result_type(x.dtype, p.dtype)
    H = np.zeros((N, 57), dtype=dt)
    for k in range(N):
        zk = z[k]
        ref = reference_function(zk[IPHI], P)
        err = error_function(zk[IP:IP + 6], ref, P)
        H[k, 0:8] = zk[0:8] - U_LIM
        H[k, 8:16] = -zk[0:8] - U_LIM
        H[k, 16:23] = zk[IQ:IQ + 7] - Q_LIM
        H[k, 23:30] = -zk[IQ:IQ + 7] - Q_LIM
        H[k, 30:37] = zk[IDQ:IDQ + 7] - DQ_LIM
        H[k, 37:44] = -zk[IDQ:IDQ + 7] - DQ_LIM
        H[k, 44] = -zk[IPHI]
        H[k, 45] = zk[IPHI] - P["phi_max"][0]
        H[k, 46] = zk[IDPHI] - P["dphi_max"][0]
        bp = (ref["bound_upper"] - ref["bound_lower"]) / 2
        c = [ref["dp_normed_d"] @ err["e_r_par"],
             err["e_p"] @ ref["bp1"] - ref["e_p_off"][0], err["e_p"] @ ref["bp2"] - ref["e_p_off"][1],
             ref["br1"] @ err["e_r_orth1"] - ref["e_r_off"][0], ref["br2"] @ err["e_r_orth2"] - ref["e_r_off"][1]]
        w = [ref["r_par_bound"], bp[0], bp[1], bp[2], bp[3]]
        for m in range(5):
            wm = w[m] if np.real(w[m]) >= 0 else -w[m]
            H[k, 47 + 2 * m] = c[m] - wm
            H[k, 48 + 2 * m] = -c[m] - wm
    return H.reshape(-1)


def bounds(N):
    """lbx, ubx, lbg, ubg (casadi_ocp_formulation.py:92-153, 272-349)."""
    inf = np.inf
    lbz = np.concatenate([-U_LIM * np.ones(8), -Q_LIM, -DQ_LIM, -inf * np.ones(7), -inf * np.ones(6),
                          -inf * np.ones(6), [0.0, -inf, -inf]])
    ubz = np.concatenate([U_LIM * np.ones(8), Q_LIM, DQ_LIM, inf * np.ones(7), inf * np.ones(6),
                          inf * np.ones(6), [inf, inf, inf]])
    lbg = np.concatenate([np.zeros(36), -inf * np.ones(7)])
    ubg = np.zeros(43)
    return np.tile(lbz, N), np.tile(ubz, N), np.tile(lbg, N), np.tile(ubg, N)


def cold_start(q0, p0, N):
    """BoundMPC.py:316-321."""
    w0 = np.zeros((N, NZ))
    w0[:, IQ:IQ + 7] = q0
    w0[:, IP:IP + 6] = p0
    return w0.reshape(-1)


# --------------------------------------------------------------------------------------
# complex-step derivatives (exact to rounding for the analytic f, g above, away from the
# piecewise-constant segment switches, whose conditions carry zero derivative as in CasADi)
# --------------------------------------------------------------------------------------
def jac_g_banded_complex_step(x, p, N, S, h, eps=1e-30):
    """Same result as jac_g_complex_step in 88 evaluations instead of 44 N: stage k's cost and constraints depend on z_{k-1} and
    z_k only, so entry j of every second stage can carry the complex step in the same evaluation."""
    n = x.size
    Jg = np.zeros((NG * N, n))
    gf = np.zeros(n)
    for par in (0, 1):
        ks = np.arange(par, N, 2)
        for j in range(NZ):
            xc = x.astype(complex)
            xc[ks * NZ + j] += 1j * eps
            fk, g = nlp_eval(xc, p, N, S, h, stage_costs=True)
            g = g.reshape(N, NG)
            for k in ks:
                Jg[k * NG:(k + 1) * NG, k * NZ + j] = g[k].imag / eps
                gf[k * NZ + j] = fk[k].imag / eps
                if k + 1 < N:
                    Jg[(k + 1) * NG:(k + 2) * NG, k * NZ + j] = g[k + 1].imag / eps
                    gf[k * NZ + j] += fk[k + 1].imag / eps
    return gf, Jg


def jac_g_complex_step(x, p, N, S, h, eps=1e-30):
    n = x.size
    Jg = np.zeros((NG * N, n))
    gf = np.zeros(n)
    xc = x.astype(complex)
    for i in range(n):
        xc[i] += 1j * eps
        f, g = nlp_eval(xc, p, N, S, h)
        Jg[:, i] = g.imag / eps
        gf[i] = f.imag / eps
        xc[i] = x[i]
    return gf, Jg
