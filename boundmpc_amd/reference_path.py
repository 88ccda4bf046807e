"""Sliding window of `nr_segs` linear pose segments over a via-point list.

Own implementation of the behaviour of the reference's ReferencePath
(/root/reference/bound_mpc/bound_mpc/ReferencePath/ReferencePath.py:10-257): arc-length
parametrisation (:89-105), integrated-omega reference (:65-75,107-109), Gram-Schmidt error-plane
bases (:111-150), window shift on `phi > phi_switch[1]` (:190-212), getters (:221-238).
Unlike the reference it never mutates the caller's lists.
Pinned by tests/golden/g4_refpath_exp{1,2}.npz."""
import numpy as np
from scipy.spatial.transform import Rotation as R


def _unit_or(v, fallback, tol):
    n = np.linalg.norm(v)
    return (v / n) if n > tol else np.array(fallback, dtype=float)


class ReferencePath:
    def __init__(self, p, r, p_limit, r_limit, bp1, br1, s, e_p_min, e_r_min, e_p_max, e_r_max, nr_segs=2, phi_bias=0):
        n = len(p)
        S = self.nr_segs = int(nr_segs)
        pad = S - 1
        self.phi_bias = phi_bias
        self.switched = True
        self.sector = 0
        ext = lambda lst: [np.array(v, dtype=float) if np.ndim(v) else v for v in lst] + [lst[-1]] * pad
        self.s, self.e_p_min, self.e_r_min = ext(list(s)), ext(list(e_p_min)), ext(list(e_r_min))
        self.e_p_max, self.e_r_max = ext(list(e_p_max)), ext(list(e_r_max))
        self.p_lower, self.p_upper = ext(list(p_limit[0])), ext(list(p_limit[1]))
        self.r_lower, self.r_upper = ext(list(r_limit[0])), ext(list(r_limit[1]))
        P = [np.array(v, dtype=float) for v in p]
        Rm = [np.array(v, dtype=float) for v in r]
        # rotation increments and the integrated-omega reference
        dr = [R.from_matrix(Rm[i] @ Rm[i - 1].T).as_rotvec() for i in range(1, n)]
        iw = [np.zeros(3)]
        for i in range(n - 1):
            iw.append(iw[i] + dr[i])
        dr += [np.array([1.0, 1.0, 1.0])] * pad
        iw += [iw[-1]] * pad
        Rm += [Rm[-1]] * pad
        # position increments (a vanishing segment inherits the previous direction)
        dp = []
        for i in range(1, n):
            d = P[i] - P[i - 1]
            if np.linalg.norm(d) < 1e-3:
                d = dp[-1] if i > 1 else np.array([0.0, 1.0, 0.0])
            dp.append(d)
        P += [P[-1]] * pad
        dp += [dp[-1]] * pad
        # arc length per segment; a pure rotation gets |dr|/pi
        seg_len = []
        for i in range(1, n):
            li = np.linalg.norm(P[i] - P[i - 1])
            if li < 1e-3:
                li = np.linalg.norm(dr[i - 1]) / np.pi
            seg_len.append(li)
        self.phi = [0] + seg_len + [1] * pad
        self.phi_max = float(np.sum(seg_len)) + phi_bias
        dr = [dr[i] / self.phi[i + 1] if i < n else dr[i] for i in range(len(dr))]
        # error-plane bases
        B1, B2, C1, C2 = [], [], [], []
        for i in range(len(bp1)):
            dn = dp[i] / np.linalg.norm(dp[i])
            b = np.array(bp1[i], dtype=float)
            b = b - (dn @ b) * dn
            if np.abs(b @ dp[i]) > 1e-6:
                print(f"[WARNING] Pos Basis vector {i} not orthogonal on path")
            if np.linalg.norm(b) < 1e-3:
                print(f"[WARNING] Pos Basis vector {i} is too close to direction")
            b = b / np.linalg.norm(b)
            B1.append(b); B2.append(np.cross(dn, b))
            om = _unit_or(dr[i], [0.0, 1.0, 0.0], 1e-4)
            c = np.array(br1[i], dtype=float)
            c = c - (om @ c) * om
            if np.abs(c @ dr[i]) > 1e-6:
                print(f"[WARNING] Rot Basis vector {i} not orthogonal on path")
            if np.linalg.norm(c) < 1e-3:
                print(f"[WARNING] Rot Basis vector {i} is too close to direction")
            c = c / np.linalg.norm(c)
            C1.append(c); C2.append(np.cross(om, c))
        self.bp1, self.bp2, self.br1, self.br2 = (lst + [lst[-1]] * pad for lst in (B1, B2, C1, C2))
        self.p, self.r, self.dp, self.dr, self.iw = P, Rm, dp, dr, iw
        self._cum = np.array(self.phi).cumsum()
        self.pd = np.zeros((6, S)); self.dpd = np.zeros((6, S)); self.dpd_normed = np.zeros((3, S)); self.ddpd = np.zeros((6, S))
        self.asymm_lower = np.zeros((4, S)); self.asymm_upper = np.zeros((4, S))
        self.phi_switch = np.ones(S + 1) * phi_bias
        for i in range(S):
            self.set_point(i)
        self.compute_normed_velocity()

    def compute_normed_velocity(self):
        for i in range(self.nr_segs):
            self.dpd_normed[:, i] = _unit_or(self.dpd[3:, i], [0.0, 1.0, 0.0], 1e-4)

    def set_point(self, idx):
        j = self.sector + idx
        self.pd[:3, idx], self.pd[3:, idx] = self.p[j], self.iw[j]
        self.dpd[:3, idx] = self.dp[j] / np.linalg.norm(self.dp[j])
        self.dpd[3:, idx] = self.dr[j]
        self.asymm_lower[:2, idx], self.asymm_lower[2:, idx] = self.p_lower[j], self.r_lower[j]
        self.asymm_upper[:2, idx], self.asymm_upper[2:, idx] = self.p_upper[j], self.r_upper[j]
        self.phi_switch[idx + 1] = self._cum[j + 1] + self.phi_bias

    def update(self, phi_current):
        if phi_current <= self.phi_switch[1]:
            self.switched = False
        while phi_current > self.phi_switch[1]:
            self.switched = True
            self.sector += 1
            for arr in (self.pd, self.dpd, self.asymm_lower, self.asymm_upper):
                arr[:, :-1] = arr[:, 1:].copy()
            self.phi_switch[:self.nr_segs - 1] = self.phi_switch[1:self.nr_segs].copy()
            self.phi_switch[self.nr_segs - 1] = self.phi_switch[self.nr_segs] + self.phi_bias
            self.set_point(self.nr_segs - 1)
            self.compute_normed_velocity()

    def get_parameters(self, phi_current):
        self.update(phi_current)
        return self.pd, self.dpd_normed, self.dpd, self.ddpd, self.phi_switch

    def get_limits(self):
        sl = slice(self.sector, self.sector + self.nr_segs)
        return (self.asymm_lower, self.asymm_upper, np.array(self.bp1[sl]).T, np.array(self.bp2[sl]).T,
                np.array(self.br1[sl]).T, np.array(self.br2[sl]).T)

    def get_bound_params(self):
        sl = slice(self.sector, self.sector + self.nr_segs)
        return (np.array(self.e_p_min[sl]), np.array(self.e_r_min[sl]), np.array(self.e_p_max[sl]),
                np.array(self.e_r_max[sl]), np.array(self.s[sl]))
