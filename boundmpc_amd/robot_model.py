"""iiwa14 kinematics and limits on the host (numpy), as a geometric chain.

Mirrors the numeric interface of the reference's RobotModel
(/root/reference/bound_mpc/bound_mpc/RobotModel/RobotModel.py): `fk` :102-116, `fk_pos` :62-100,
`jacobian_fk` :254-373, `djacobian_fk` :375-563, `ddjacobian_fk` :565-1053,
`forward_kinematics` :50-57, `get_robot_limits` :45-48, limits :20-43.  The reference holds
Maple-expanded trig polynomials; here the same functions are computed from the chain geometry
(joint axes z,y,z,-y,z,y,z; link lengths :9-16).  Equality is pinned by tests against golden
vectors generated from the reference's own code (tests/golden/g1_kinematics.npz)."""
import numpy as np
from scipy.spatial.transform import Rotation as R

_AX = (2, 1, 2, -1, 2, 1, 2)
_PRE_Z = (0.0, 0.1575 + 0.2025, 0.0, 0.2375 + 0.1825, 0.0, 0.2175 + 0.1825, 0.0)
_TOOL_Z = 0.081 + (0.071 + 0.145)


def _chain(q):
    Rm = np.eye(3)
    o = np.zeros(3)
    A = np.zeros((7, 3))
    O = np.zeros((7, 3))
    for j in range(7):
        o = o + Rm[:, 2] * _PRE_Z[j]
        c, s = np.cos(q[j]), np.sin(q[j])
        O[j] = o
        if _AX[j] == 2:
            A[j] = Rm[:, 2]
            c0, c1 = Rm[:, 0].copy(), Rm[:, 1].copy()
            Rm[:, 0], Rm[:, 1] = c * c0 + s * c1, -s * c0 + c * c1
        else:
            sg = float(_AX[j])
            A[j] = sg * Rm[:, 1]
            s = s * sg
            c0, c2 = Rm[:, 0].copy(), Rm[:, 2].copy()
            Rm[:, 0], Rm[:, 2] = c * c0 - s * c2, s * c0 + c * c2
    p = o + Rm[:, 2] * _TOOL_Z
    return A, O, p, Rm


class RobotModel:
    def __init__(self):
        d2r = np.pi / 180
        self.q_lim_upper = [165 * d2r, 115 * d2r, 165 * d2r, 115 * d2r, 165 * d2r, 115 * d2r, 170 * d2r]
        self.q_lim_lower = [-v for v in self.q_lim_upper]
        self.dq_lim_upper = [85 * d2r, 85 * d2r, 100 * d2r, 75 * d2r, 130 * d2r, 135 * d2r, 135 * d2r]
        self.dq_lim_lower = [-v for v in self.dq_lim_upper]
        self.tau_lim_upper = [320, 320, 176, 176, 110, 40, 40]
        self.tau_lim_lower = [-v for v in self.tau_lim_upper]
        self.u_max, self.u_min = 35, -35

    def get_robot_limits(self):
        return (self.q_lim_upper, self.q_lim_lower, self.dq_lim_upper, self.dq_lim_lower, self.tau_lim_upper,
                self.tau_lim_lower, self.u_max, self.u_min)

    def hom_transform_endeffector(self, q):
        _, _, p, Rm = _chain(np.asarray(q, dtype=float))
        H = np.eye(4)
        H[:3, :3], H[:3, 3] = Rm, p
        return H

    def fk_pos(self, q):
        return _chain(np.asarray(q, dtype=float))[2]

    def fk(self, q):
        _, _, p, Rm = _chain(np.asarray(q, dtype=float))
        return np.concatenate([p, R.from_matrix(Rm).as_rotvec()])

    def jacobian_fk(self, q):
        A, O, p, _ = _chain(np.asarray(q, dtype=float))
        J = np.zeros((6, 7))
        J[:3] = np.cross(A, p - O).T
        J[3:] = A.T
        return J

    @staticmethod
    def _dJ_dq(A, Wc, i):
        """dJ/dq_i (6x7): column j of J_v is w_j = a_j x r_j."""
        D = np.zeros((6, 7))
        for j in range(7):
            D[:3, j] = np.cross(A[i], Wc[j]) if i <= j else np.cross(A[j], Wc[i])
            if i < j:
                D[3:, j] = np.cross(A[i], A[j])
        return D

    def djacobian_fk(self, q, dq):
        A, O, p, _ = _chain(np.asarray(q, dtype=float))
        Wc = np.cross(A, p - O)
        dJ = np.zeros((6, 7))
        for i in range(7):
            dJ += self._dJ_dq(A, Wc, i) * dq[i]
        return dJ

    def ddjacobian_fk(self, q, q_p, q_pp):
        """d^2 J / dt^2 = sum_i dJ/dq_i ddq_i + sum_{i,l} d2J/(dq_i dq_l) dq_i dq_l."""
        A, O, p, _ = _chain(np.asarray(q, dtype=float))
        Wc = np.cross(A, p - O)
        out = np.zeros((6, 7))
        for i in range(7):
            out += self._dJ_dq(A, Wc, i) * q_pp[i]
        cr = np.cross
        for i in range(7):
            for l in range(7):
                lo, hi = (i, l) if i <= l else (l, i)
                f = q_p[i] * q_p[l]
                if f == 0.0:
                    continue
                for j in range(7):
                    # second derivative of w_j w.r.t. (q_lo, q_hi), lo <= hi
                    if j >= hi:
                        v = cr(A[lo], cr(A[hi], Wc[j]))
                    elif j >= lo:
                        v = cr(A[lo], cr(A[j], Wc[hi]))
                    else:
                        v = cr(A[j], cr(A[lo], Wc[hi]))
                    out[:3, j] += f * v
                    if j > hi:
                        out[3:, j] += f * cr(A[lo], cr(A[hi], A[j]))
        return out

    def forward_kinematics(self, q, dq):
        return self.fk(q), self.jacobian_fk(q), self.djacobian_fk(q, dq)

    def velocity_ee(self, q, dq):
        return self.jacobian_fk(q)[:3] @ np.asarray(dq, dtype=float)

    def omega_ee(self, q, dq):
        return self.jacobian_fk(q)[3:] @ np.asarray(dq, dtype=float)
