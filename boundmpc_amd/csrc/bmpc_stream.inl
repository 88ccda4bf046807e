// bmpc_stream.inl -- the per-stream host arithmetic of BoundMPC.step() as device code.
//
// Two functions, one stream (trajectory) per 64-lane wave (phases over cooperating lanes; the CPU test build runs them with one lane):
//   stream_pack  (f1 + f3): sliding path window (ReferencePath.update/get_* , ReferencePath.py:178-238), warm-start vector
//                (cold start BoundMPC.py:316-321, integrated-omega unwrap :326-333, shift :372-375), initial orientation errors
//                (util_functions.py:11-31), SO(3) Jacobians and dual basis (BoundMPC.py:267-304, lie_functions.py:41-64), tube
//                quartics (BoundMPC.py:219-265, mpc_utils_casadi.py:130-137), parameter vector p in the layout of
//                casadi_ocp_formulation.py:361-376 / BoundMPC.py:416-443.
//   stream_post  (f2 + f3): feasibility rule and fallback to the previous plan (BoundMPC.py:460-506), re-integration of the
//                joint and path states from the optimal jerks (:513-555), Cartesian trajectory (:568-587), advance of the
//                path-parameter state and of the rotation reference (:594-611, util_functions.py:88-99), optional kinematic
//                plant step (util_functions.py:152-161) so that a closed loop runs without leaving the device.
// Re-planning (BoundMPC.update, BoundMPC.py:163-217): the host writes the new path table and the handful of state scalars update()
// sets (boundmpc_amd.stream.apply_update), and raises SS_UPDATED; from then on -- the reference never clears `self.updated` --
// stream_pack takes the re-projection branch of step() (:335-369: the path-parameter states of the warm start are re-projected
// from the Cartesian position / velocity / acceleration / jerk of the previous plan, no shift), for which stream_post keeps those
// four 3 x N arrays of the last plan in the state (:557-566; J, dJ and the second time derivative of J of the geometric chain).
// Not covered: the RViz logging dictionaries (host mirror).
//
// Rotation conversions follow scipy.spatial.transform.Rotation's algorithms (from_rotvec / as_matrix / from_matrix /
// as_rotvec / as_euler('zyx')) so that the parameters agree with the host mirror to round-off.
// The same text is compiled by g++ for the CPU tests (tests/emu).
#pragma once

#ifndef BMPCS_OPAQUE
#define BMPCS_OPAQUE(x)      // GPU builds: asm volatile("" : "+v"(x)) (bmpc_gpu_common.h); host builds of the same text: nothing
#endif
namespace bmpcs {
// sine / cosine of a BOUNDED angle (joint angles, rotation angles <= 2 pi, half angles): the wave program's two-constant Cody-Waite reduction
// (bmpc_wave.inl bmpc_sincos: < 1 ulp, checked against libm over [-50, 50]) instead of the library's full-range functions, whose
// Payne-Hanek path (v_trig_preop_f64, a private work array) was inlined at every call site of the pack / post kernels
BMPC_HD inline double sin_b(double x) { double s_, c_; BMPC_NAMESPACE::bmpc_sincos(x, &s_, &c_); return s_; }
BMPC_HD inline double cos_b(double x) { double s_, c_; BMPC_NAMESPACE::bmpc_sincos(x, &s_, &c_); return c_; }
constexpr int STREAM_NMAX = 40;      // the solver's longest horizon (stream_post: one role per stage, then the fixed roles behind them)

// path table entry (one per via-point slot), doubles
enum { PT_P = 0, PT_IW = 3, PT_DPN = 6, PT_DR = 9, PT_RRV = 12, PT_PLO = 15, PT_PUP = 17, PT_RLO = 19, PT_RUP = 21, PT_BP1 = 23, PT_BP2 = 26,
       PT_BR1 = 29, PT_BR2 = 32, PT_CUM = 35, PT_EPMIN = 36, PT_ERMIN = 37, PT_EPMAX = 38, PT_ERMAX = 39, PT_S = 40, PT_LEN = 48 };
// stream state, doubles; the previous solution [N][44] follows at SS_PREV
enum { SS_SECTOR = 0, SS_HASPREV = 1, SS_ERRCNT = 2, SS_PHI = 3, SS_DPHI = 4, SS_DDPHI = 5, SS_DDDPHI = 6, SS_PRREF = 7, SS_IWREF = 10,
       SS_PHIMAX = 13, SS_W = 14 /* 15 weights */, SS_NENT = 29, SS_USINGPREV = 30, SS_VALID = 31, SS_PREV = 32 };
// robot record per tick: the arguments of step(q, dq, ddq, p_lie, v, x_phi_d, jerk)
enum { RB_Q = 0, RB_DQ = 7, RB_DDQ = 14, RB_P = 21, RB_V = 27, RB_XPHID = 33, RB_JERK = 36, RB_LEN = 43 };
// trajectory record: [q dq ddq dddq](7 x N each) [p v a](6 x N each) [phi dphi ddphi dddphi](N each) n_valid
BMPC_HD inline int tr_len(int N) { return 4 * 7 * N + 3 * 6 * N + 4 * N + 4; }
// after the previous solution: Cartesian [pos | vel | acc | jerk] (linear parts, 3 x N each, [c][i]) of the previous plan, then the
// `updated` flag of the reference object
BMPC_HD inline int ss_pc(int N) { return SS_PREV + 44 * N; }
BMPC_HD inline int ss_updated(int N) { return SS_PREV + 56 * N; }
BMPC_HD inline int ss_len(int N) { return SS_PREV + 56 * N + 2; }

BMPC_HD inline double norm3(const double *a) { return BMPC_SQRT(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
BMPC_HD inline double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
BMPC_HD inline void cross3s(const double *a, const double *b, double *c) {
    const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    c[0] = x; c[1] = y; c[2] = z;
}
BMPC_HD inline void mat3_mul(const double *A, const double *B, double *C) {   // row-major 3x3
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    for (int i = 0; i < 9; i++) C[i] = t[i];
}
BMPC_HD inline void mat3_mul_bt(const double *A, const double *B, double *C) {   // A * B^T
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
    for (int i = 0; i < 9; i++) C[i] = t[i];
}
// rotation vector -> unit quaternion (x, y, z, w)
BMPC_HD inline void rotvec_to_quat(const double *v, double *q) {
    const double ang = norm3(v);
    double sc;
    if (ang <= 1e-3) { const double a2 = ang * ang; sc = 0.5 - a2 / 48 + a2 * a2 / 3840; }
    else sc = sin_b(ang / 2) / ang;
    q[0] = sc * v[0]; q[1] = sc * v[1]; q[2] = sc * v[2]; q[3] = cos_b(ang / 2);
}
BMPC_HD inline void quat_to_mat(const double *q, double *M) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w, xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    M[0] = x2 - y2 - z2 + w2; M[3] = 2 * (xy + zw); M[6] = 2 * (xz - yw);
    M[1] = 2 * (xy - zw); M[4] = -x2 + y2 - z2 + w2; M[7] = 2 * (yz + xw);
    M[2] = 2 * (xz + yw); M[5] = 2 * (yz - xw); M[8] = -x2 - y2 + z2 + w2;
}
BMPC_HD inline void rotvec_to_mat(const double *v, double *M) { double q[4]; rotvec_to_quat(v, q); quat_to_mat(q, M); }
// rotation matrix -> unit quaternion (largest of the diagonal / trace decides the branch)
BMPC_HD inline void mat_to_quat(const double *M, double *q) {
    const double d0 = M[0], d1 = M[4], d2 = M[8], tr = d0 + d1 + d2;
    int choice = 0; double best = d0;
    if (d1 > best) { best = d1; choice = 1; }
    if (d2 > best) { best = d2; choice = 2; }
    if (tr > best) { best = tr; choice = 3; }
    if (choice != 3) {
        const int i = choice, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1 - tr + 2 * M[i * 3 + i];
        q[j] = M[j * 3 + i] + M[i * 3 + j];
        q[k] = M[k * 3 + i] + M[i * 3 + k];
        q[3] = M[k * 3 + j] - M[j * 3 + k];
    } else {
        q[0] = M[7] - M[5]; q[1] = M[2] - M[6]; q[2] = M[3] - M[1]; q[3] = 1 + tr;
    }
    const double n = BMPC_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}
BMPC_HD inline void quat_to_rotvec(const double *qin, double *v) {
    double q[4] = {qin[0], qin[1], qin[2], qin[3]};
    if (q[3] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
    const double ang = 2 * BMPC_ATAN2(norm3(q), q[3]);
    double sc;
    if (ang <= 1e-3) { const double a2 = ang * ang; sc = 2 + a2 / 12 + 7 * a2 * a2 / 2880; }
    else sc = ang / sin_b(ang / 2);
    v[0] = sc * q[0]; v[1] = sc * q[1]; v[2] = sc * q[2];
}
BMPC_HD inline void mat_to_rotvec(const double *M, double *v) { double q[4]; mat_to_quat(M, q); quat_to_rotvec(q, v); }
// extrinsic z-y-x Euler angles of R = Rx(e2) Ry(e1) Rz(e0)
BMPC_HD inline void mat_to_euler_zyx(const double *M, double *e) {
    // through the quaternion, as scipy does, so that a slightly non-orthogonal input is treated alike
    double q[4], Rn[9]; mat_to_quat(M, q); quat_to_mat(q, Rn);
    e[1] = BMPC_ATAN2(Rn[2], BMPC_SQRT(Rn[0] * Rn[0] + Rn[1] * Rn[1]));
    e[0] = BMPC_ATAN2(-Rn[1], Rn[0]);
    e[2] = BMPC_ATAN2(-Rn[5], Rn[8]);
}
// J_r^{-1}, J_l^{-1} of SO(3) with the reference's angle = |axis| + 1e-6 (lie_functions.py:41-64), row-major
BMPC_HD inline void jac_so3_inv(const double *a, double sign, double *J) {
    const double th = norm3(a) + 1e-6;
    const double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
    double K2[9]; mat3_mul(K, K, K2);
    const double c = 1 / (th * th) - (1 + cos_b(th)) / (2 * th * sin_b(th));
    for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0 ? 1.0 : 0.0) + sign * 0.5 * K[i] + c * K2[i];
}
BMPC_HD inline void mat3_vec(const double *A, const double *x, double *y) {
    const double a = A[0] * x[0] + A[1] * x[1] + A[2] * x[2], b = A[3] * x[0] + A[4] * x[1] + A[5] * x[2], c = A[6] * x[0] + A[7] * x[1] + A[8] * x[2];
    y[0] = a; y[1] = b; y[2] = c;
}
BMPC_HD inline void unit_or_y(const double *v, double tol, double *out) {
    const double n = norm3(v);
    if (n > tol) { out[0] = v[0] / n; out[1] = v[1] / n; out[2] = v[2] / n; } else { out[0] = 0; out[1] = 1; out[2] = 0; }
}
// rotate the reference by the constant angular velocity om over (phi1 - phi0)  (util_functions.py:88-99)
BMPC_HD inline void integrate_rotation_reference(const double *pr_ref, const double *om, double phi0, double phi1, double *out) {
    double R0[9]; rotvec_to_mat(pr_ref, R0);
    const double n = norm3(om);
    if (n > 1e-4) {
        const double k[3] = {om[0] / n, om[1] / n, om[2] / n};
        const double K[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
        double K2[9]; mat3_mul(K, K, K2);
        const double ang = (phi1 - phi0) * n, s = sin_b(ang), c1 = 1 - cos_b(ang);
        double E[9];
        for (int i = 0; i < 9; i++) E[i] = (i % 4 == 0 ? 1.0 : 0.0) + s * K[i] + c1 * K2[i];
        mat3_mul(E, R0, R0);
    }
    mat_to_rotvec(R0, out);
}
// hat-function jerk integrator, one step (jerk_trajectory_casadi.py:78-175 closed form)
BMPC_HD inline void chain_step(double &x, double &dx, double &ddx, double up, double u, double h) {
    const double xn = x + h * dx + h * h / 2 * ddx + h * h * h / 8 * up + h * h * h / 24 * u;
    const double dxn = dx + h * ddx + h * h / 3 * up + h * h / 6 * u;
    const double ddxn = ddx + h / 2 * (up + u);
    x = xn; dx = dxn; ddx = ddxn;
}

// iiwa14 geometric chain (RobotModel.py:9-16,62-116,254-563): EE pose [pos; rotvec], J (6x7 row-major), dJ
struct Fk { double p[6], J[42], dJ[42]; };
// Second time derivative of the LINEAR rows of J along a motion with joint velocity dq and acceleration ddq (what
// RobotModel.ddjacobian_fk gives in rows 0..2, RobotModel.py:565-1053), for the geometric chain: with a_j the axes, r_j = p - o_j,
// w_j = a_j x r_j (column j of J_v), omega_j = sum_{i<j} dq_i a_i:
//   a_j' = omega_j x a_j,  r_j' = omega_j x r_j + sum_{i>=j} dq_i w_i,  w_j' = a_j' x r_j + a_j x r_j'   (= column j of dJ_v)
//   a_j'' = omega_j' x a_j + omega_j x a_j',  r_j'' = omega_j' x r_j + omega_j x r_j' + sum_{i>=j} (ddq_i w_i + dq_i w_i'),
//   w_j'' = a_j'' x r_j + 2 a_j' x r_j' + a_j x r_j''
BMPC_HD inline void jacobian_lin_ddot(const double *q, const double *dq, const double *ddq, double *ddJ /* [3][7] */) {
    const int ax[7] = {2, 1, 2, -1, 2, 1, 2};
    const double pre[7] = {0.0, 0.1575 + 0.2025, 0.0, 0.2375 + 0.1825, 0.0, 0.2175 + 0.1825, 0.0};
    const double tool = 0.081 + (0.071 + 0.145);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0}, A[7][3], O[7][3], P[3];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        for (int c = 0; c < 3; c++) { o[c] += R[c * 3 + 2] * pre[j]; O[j][c] = o[c]; }
        double cs = cos_b(q[j]), sn = sin_b(q[j]);
        if (ax[j] == 2) {
            for (int c = 0; c < 3; c++) { A[j][c] = R[c * 3 + 2]; const double c0 = R[c * 3], c1 = R[c * 3 + 1]; R[c * 3] = cs * c0 + sn * c1; R[c * 3 + 1] = -sn * c0 + cs * c1; }
        } else {
            const double sg = (double)ax[j]; sn *= sg;
            for (int c = 0; c < 3; c++) { A[j][c] = sg * R[c * 3 + 1]; const double c0 = R[c * 3], c2 = R[c * 3 + 2]; R[c * 3] = cs * c0 - sn * c2; R[c * 3 + 2] = sn * c0 + cs * c2; }
        }
    }
    for (int c = 0; c < 3; c++) P[c] = o[c] + R[c * 3 + 2] * tool;
    double r[7][3], w[7][3], om[7][3], da[7][3], dr[7][3], dw[7][3], suf[3] = {0, 0, 0}, t[3];
#pragma unroll
    for (int j = 0; j < 7; j++) { for (int c = 0; c < 3; c++) r[j][c] = P[c] - O[j][c]; cross3s(A[j], r[j], w[j]); }
    { double acc[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 7; j++) { for (int c = 0; c < 3; c++) { om[j][c] = acc[c]; acc[c] += dq[j] * A[j][c]; } } }
#pragma unroll
    for (int j = 6; j >= 0; j--) {                      // suffix sums of dq_i w_i (inclusive)
        for (int c = 0; c < 3; c++) suf[c] += dq[j] * w[j][c];
        cross3s(om[j], A[j], da[j]);
        cross3s(om[j], r[j], t); for (int c = 0; c < 3; c++) dr[j][c] = t[c] + suf[c];
        double u1[3], u2[3]; cross3s(da[j], r[j], u1); cross3s(A[j], dr[j], u2);
        for (int c = 0; c < 3; c++) dw[j][c] = u1[c] + u2[c];
    }
    double dom[7][3];
    { double acc[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 7; j++) { for (int c = 0; c < 3; c++) { dom[j][c] = acc[c]; acc[c] += ddq[j] * A[j][c] + dq[j] * da[j][c]; } } }
    double suf2[3] = {0, 0, 0};
#pragma unroll
    for (int j = 6; j >= 0; j--) {
        for (int c = 0; c < 3; c++) suf2[c] += ddq[j] * w[j][c] + dq[j] * dw[j][c];
        double dda[3], ddr[3], u1[3], u2[3], u3[3];
        cross3s(dom[j], A[j], u1); cross3s(om[j], da[j], u2); for (int c = 0; c < 3; c++) dda[c] = u1[c] + u2[c];
        cross3s(dom[j], r[j], u1); cross3s(om[j], dr[j], u2); for (int c = 0; c < 3; c++) ddr[c] = u1[c] + u2[c] + suf2[c];
        cross3s(dda, r[j], u1); cross3s(da[j], dr[j], u2); cross3s(A[j], ddr, u3);
        for (int c = 0; c < 3; c++) ddJ[c * 7 + j] = u1[c] + 2.0 * u2[c] + u3[c];
    }
}
BMPC_HD inline void forward_kinematics(const double *q, const double *dq, Fk &F) {
    const int ax[7] = {2, 1, 2, -1, 2, 1, 2};
    const double pre[7] = {0.0, 0.1575 + 0.2025, 0.0, 0.2375 + 0.1825, 0.0, 0.2175 + 0.1825, 0.0};
    const double tool = 0.081 + (0.071 + 0.145);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0}, A[7][3], O[7][3];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        for (int c = 0; c < 3; c++) { o[c] += R[c * 3 + 2] * pre[j]; O[j][c] = o[c]; }
        double cs = cos_b(q[j]), sn = sin_b(q[j]);
        if (ax[j] == 2) {
            for (int c = 0; c < 3; c++) { A[j][c] = R[c * 3 + 2]; const double c0 = R[c * 3], c1 = R[c * 3 + 1]; R[c * 3] = cs * c0 + sn * c1; R[c * 3 + 1] = -sn * c0 + cs * c1; }
        } else {
            const double sg = (double)ax[j]; sn *= sg;
            for (int c = 0; c < 3; c++) { A[j][c] = sg * R[c * 3 + 1]; const double c0 = R[c * 3], c2 = R[c * 3 + 2]; R[c * 3] = cs * c0 - sn * c2; R[c * 3 + 2] = sn * c0 + cs * c2; }
        }
    }
    for (int c = 0; c < 3; c++) F.p[c] = o[c] + R[c * 3 + 2] * tool;
    mat_to_rotvec(R, F.p + 3);
    double Wc[7][3];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        const double r[3] = {F.p[0] - O[j][0], F.p[1] - O[j][1], F.p[2] - O[j][2]};
        cross3s(A[j], r, Wc[j]);
        for (int c = 0; c < 3; c++) { F.J[c * 7 + j] = Wc[j][c]; F.J[(3 + c) * 7 + j] = A[j][c]; }
    }
    for (int i = 0; i < 42; i++) F.dJ[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const double f = dq[i];
#pragma unroll
        for (int j = 0; j < 7; j++) {
            double v[3];
            if (i <= j) cross3s(A[i], Wc[j], v); else cross3s(A[j], Wc[i], v);
            for (int c = 0; c < 3; c++) F.dJ[c * 7 + j] += f * v[c];
            if (i < j) { cross3s(A[i], A[j], v); for (int c = 0; c < 3; c++) F.dJ[(3 + c) * 7 + j] += f * v[c]; }
        }
    }
}

// Pose, Cartesian velocity / acceleration and the linear jerk rows of a planned motion in ONE pass over the chain (round 4): what the
// post-processing needs of forward_kinematics + jacobian_lin_ddot,
//   v = J dq,  a = J ddq + dJ dq  (6 rows),  jk = J u + dJ ddq + ddJ dq  (3 linear rows),
// accumulated joint by joint from the same chain quantities (notation of jacobian_lin_ddot): v_lin, a_lin are the totals of its two suffix
// sums, v_ang, a_ang the totals of its two prefix sums.  No Jacobian array is formed: the two functions together kept ~280 doubles live
// (J, dJ, ddJ and nine 7x3 work arrays: 600-750 B of scratch per lane in the post / fused-tick kernels), this one 84 (axes, lever arms and
// the two prefix sums per joint).  dJ_v of forward_kinematics and dw_j here are the same vector by the Jacobi identity.
struct FkMotion { double p[6], v[6], a[6], jk[3]; };
BMPC_HD inline void fk_motion(const double *q, const double *dq, const double *ddq, const double *u, FkMotion &M) {
    const int ax[7] = {2, 1, 2, -1, 2, 1, 2};
    const double pre[7] = {0.0, 0.1575 + 0.2025, 0.0, 0.2375 + 0.1825, 0.0, 0.2175 + 0.1825, 0.0};
    const double tool = 0.081 + (0.071 + 0.145);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0}, A[7][3], r[7][3];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        for (int c = 0; c < 3; c++) { o[c] += R[c * 3 + 2] * pre[j]; r[j][c] = o[c]; }
        double cs, sn; BMPC_NAMESPACE::bmpc_sincos(q[j], &sn, &cs);      // joint angles are bounded: the wave program's two-constant reduction (bmpc_wave.inl) instead of the library's full-range one
        if (ax[j] == 2) {
            for (int c = 0; c < 3; c++) { A[j][c] = R[c * 3 + 2]; const double c0 = R[c * 3], c1 = R[c * 3 + 1]; R[c * 3] = cs * c0 + sn * c1; R[c * 3 + 1] = -sn * c0 + cs * c1; }
        } else {
            const double sg = (double)ax[j]; sn *= sg;
            for (int c = 0; c < 3; c++) { A[j][c] = sg * R[c * 3 + 1]; const double c0 = R[c * 3], c2 = R[c * 3 + 2]; R[c * 3] = cs * c0 - sn * c2; R[c * 3 + 2] = sn * c0 + cs * c2; }
        }
    }
    for (int c = 0; c < 3; c++) M.p[c] = o[c] + R[c * 3 + 2] * tool;
    mat_to_rotvec(R, M.p + 3);
    // forward: omega_j = sum_{i<j} dq_i a_i,  omega_j' = sum_{i<j} (ddq_i a_i + dq_i a_i'),  a_i' = omega_i x a_i
    double om[7][3], dom[7][3];
    {
        double acc[3] = {0, 0, 0}, dacc[3] = {0, 0, 0}, da[3];
#pragma unroll
        for (int j = 0; j < 7; j++) {
            for (int c = 0; c < 3; c++) { om[j][c] = acc[c]; dom[j][c] = dacc[c]; r[j][c] = M.p[c] - r[j][c]; }
            cross3s(acc, A[j], da);
            for (int c = 0; c < 3; c++) { acc[c] += dq[j] * A[j][c]; dacc[c] += ddq[j] * A[j][c] + dq[j] * da[c]; }
        }
        for (int c = 0; c < 3; c++) { M.v[3 + c] = acc[c]; M.a[3 + c] = dacc[c]; }
    }
    // backward: suffix sums of dq_i w_i and of ddq_i w_i + dq_i w_i'; w_j', w_j'' as in jacobian_lin_ddot
    double suf[3] = {0, 0, 0}, suf2[3] = {0, 0, 0}, jk[3] = {0, 0, 0};
#pragma unroll
    for (int j = 6; j >= 0; j--) {
        double w[3], da[3], dr[3], dw[3], dda[3], ddr[3], t[3], u1[3], u2[3], u3[3];
        cross3s(A[j], r[j], w);
        for (int c = 0; c < 3; c++) suf[c] += dq[j] * w[c];
        cross3s(om[j], A[j], da);
        cross3s(om[j], r[j], t); for (int c = 0; c < 3; c++) dr[c] = t[c] + suf[c];
        cross3s(da, r[j], u1); cross3s(A[j], dr, u2);
        for (int c = 0; c < 3; c++) dw[c] = u1[c] + u2[c];
        for (int c = 0; c < 3; c++) suf2[c] += ddq[j] * w[c] + dq[j] * dw[c];
        cross3s(dom[j], A[j], u1); cross3s(om[j], da, u2); for (int c = 0; c < 3; c++) dda[c] = u1[c] + u2[c];
        cross3s(dom[j], r[j], u1); cross3s(om[j], dr, u2); for (int c = 0; c < 3; c++) ddr[c] = u1[c] + u2[c] + suf2[c];
        cross3s(dda, r[j], u1); cross3s(da, dr, u2); cross3s(A[j], ddr, u3);
        for (int c = 0; c < 3; c++) jk[c] += u[j] * w[c] + ddq[j] * dw[c] + dq[j] * (u1[c] + 2.0 * u2[c] + u3[c]);
    }
    for (int c = 0; c < 3; c++) { M.v[c] = suf[c]; M.a[c] = suf2[c]; M.jk[c] = jk[c]; }
}

// quartic a4..a0 on [0, L] with f(0)=e0, f(L)=e1, f(L/2)=emax, f'(0)=s, f'(L)=-s  (mpc_utils_casadi.py:130-137 at phi0 = 0)
BMPC_HD inline void bound_params(double L, double e0, double e1, double s, double emax, double *a4, double *a3, double *a2, double *a1, double *a0) {
    const double r1 = e1 - e0 - s * L, r2 = emax - e0 - s * L / 2, r3 = -2 * s * L;
    const double A = 16 * r2 - 5 * r1 + r3, B = -32 * r2 + 14 * r1 - 3 * r3, C = 16 * r2 - 8 * r1 + 2 * r3;
    const double L2 = L * L;
    *a4 = C / (L2 * L2); *a3 = B / (L2 * L); *a2 = A / L2; *a1 = s + 0 * e0; *a0 = e0 + 0 * s;
}

// ------------------------------------------------------------------------------------------
// Both functions are written as PHASES over `nl` cooperating lanes (`lane` = 0..nl-1) with BMPCS_SYNC() between phases:
// the device kernels run one 64-lane wave per stream (nl = 64, BMPCS_SYNC = workgroup barrier, `sh` in LDS) so that copies are
// coalesced and the independent pieces (path segments, horizon stages, integrator chains) run side by side; the CPU test
// build runs the same text with nl = 1.  `sh`: shared workspace of SH_LEN doubles.
// ------------------------------------------------------------------------------------------
enum { SH_DTAU = 0, SH_RTAU = 3, SH_JR = 12, SH_JL = 21, SH_PHISW = 30, SH_FLAG = 36 /* 8 */, SH_RED = 44 /* 64 */, SH_LEN = 108 };

// f1 + f3: pack one stream.  path [nent][PT_LEN], ss stream state, rb robot record, p [141+91 S], x0 [N][44],
// dual (may be null): the solver's dual state [57 N + 2], shifted with the plan.
// cap = entries the path table holds: the entry count and the window start read from the state row are clamped to it, so that a
// corrupted state (a NaN, a row written by a racing update) cannot index outside the table.
// xlast (may be null; real-time mode of the fused tick): the solver's iterate of the previous tick.  When the state row says it is
// usable (stream_post sets the word behind the `updated` flag), the warm start shifts IT instead of the last ACCEPTED plan: an iterate
// the acceptance rule rejected is not applied to the plant, but the iterations spent on it are not thrown away either -- the next
// tick continues from it while the plant replays the accepted plan (the reference restarts from the last accepted plan, shifted once,
// however many ticks ago that was: BoundMPC.py:322-375,468-489).
BMPC_HD inline void stream_pack(int N, int S, const double *path, int cap, double *ss, const double *rb, double *p, double *x0, double *dual,
                                const double *xlast, double *sh, int lane, int nl, double lvl_c = 0.0, double lvl_lo = 0.0, double lvl_hi = 0.0) {
    int nent = (int)ss[SS_NENT];
    nent = nent > cap ? cap : nent; nent = nent < S + 1 ? S + 1 : nent;
    const double phi_cur = ss[SS_PHI];
    const double *q0 = rb + RB_Q, *p0 = rb + RB_P;
    const bool has_prev = ss[SS_HASPREV] > 0.5;
    // ReferencePath.update: slide the window while phi passed the first switch (every lane computes the same sector)
    int sector = (int)ss[SS_SECTOR];
    sector = sector > nent - S - 1 ? nent - S - 1 : sector; sector = sector < 0 ? 0 : sector;
    while (sector + S + 1 < nent && phi_cur > path[(sector + 1) * PT_LEN + PT_CUM]) sector++;
    // ---- phase 0 (one lane): initial orientation error, its rotation matrix, SO(3) Jacobians ----
    if (lane == 0) {
        double Ra[9], Rb[9], Rd[9], dtau[3];
        rotvec_to_mat(p0 + 3, Ra); rotvec_to_mat(ss + SS_PRREF, Rb); mat3_mul_bt(Ra, Rb, Rd);
        mat_to_rotvec(Rd, dtau);
        for (int c = 0; c < 3; c++) sh[SH_DTAU + c] = dtau[c];
        rotvec_to_mat(dtau, sh + SH_RTAU);
        jac_so3_inv(dtau, +1.0, sh + SH_JR); jac_so3_inv(dtau, -1.0, sh + SH_JL);
    }
    BMPCS_SYNC();
    // parameter-vector offsets (casadi_ocp_formulation.py:361-376)
    const int o_par = 42, o_o1 = o_par + 3 * S, o_o2 = o_o1 + 3 * S, o_xphid = o_o2 + 3 * S, o_jerk = o_xphid + 3, o_sw = o_jerk + 8, o_jr = o_sw + S + 1,
              o_jl = o_jr + 9, o_pref = o_jl + 9, o_dpref = o_pref + 6 * S, o_dpn = o_dpref + 6 * S, o_b = o_dpn + 3 * S, o_a = o_b + 12 * S,
              o_w = o_a + 45 * (S + 1), o_pm = o_w + 15, o_v1 = o_pm + 2, o_v2 = o_v1 + 3 * S, o_v3 = o_v2 + 3 * S, o_qd = o_v3 + 3 * S;
    const double phi_max = ss[SS_PHIMAX];
    const double phimax_p = BMPC_FMIN(phi_cur + 5.0, phi_max);
    // ---- phase 1a: one lane per path segment: orientation-error split, dual basis, tube quartics ----
    for (int i = lane; i < S; i += nl) {
        const double *e = path + (sector + i) * PT_LEN, *Rtau = sh + SH_RTAU, *jr = sh + SH_JR;
        double dn[3], par[3], o1[3], o2[3];
        unit_or_y(e + PT_DR, 1e-4, dn);
        const double *b1 = e + PT_BR1, *b2 = e + PT_BR2;
        {
            const double F[9] = {b2[0], dn[0], b1[0], b2[1], dn[1], b1[1], b2[2], dn[2], b1[2]};   // columns br2, d, br1
            double T[9], M[9], Ft[9], eul[3];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ft[r * 3 + c] = F[c * 3 + r];
            mat3_mul(Rtau, F, T); mat3_mul(Ft, T, M);
            mat_to_euler_zyx(M, eul);
            for (int c = 0; c < 3; c++) { par[c] = eul[1] * dn[c]; o1[c] = eul[0] * b1[c]; o2[c] = eul[2] * b2[c]; }
        }
        double Ro[9], Rp[9], rest1[9], rest2[9], rv[3], Jt[9], t1[3], t2[3], t3[3];
        rotvec_to_mat(o1, Ro); rotvec_to_mat(par, Rp);
        mat3_mul_bt(Rtau, Ro, rest1); mat3_mul_bt(rest1, Rp, rest2);
        mat3_vec(jr, b1, t1);
        mat_to_rotvec(rest1, rv); jac_so3_inv(rv, +1.0, Jt); mat3_vec(Jt, dn, t2);
        mat_to_rotvec(rest2, rv); jac_so3_inv(rv, +1.0, Jt); mat3_vec(Jt, b2, t3);
        double c23[3], c31[3], c12[3];                     // rows of inv([t1 t2 t3]): cross products over the determinant
        cross3s(t2, t3, c23); cross3s(t3, t1, c31); cross3s(t1, t2, c12);
        const double det = dot3(t1, c23);
        for (int c = 0; c < 3; c++) {
            p[o_par + 3 * i + c] = par[c]; p[o_o1 + 3 * i + c] = o1[c]; p[o_o2 + 3 * i + c] = o2[c];
            p[o_dpn + c * S + i] = dn[c];
            p[o_v1 + c * S + i] = c23[c] / det; p[o_v2 + c * S + i] = c31[c] / det; p[o_v3 + c * S + i] = c12[c] / det;
        }
        // tube quartics a4..a0 [chan 0..8][seg 0..S]; only the first window entry's bound parameters are used (:224-232)
        const double *e0p = path + sector * PT_LEN;
        const double pm = e0p[PT_EPMIN], rmn = e0p[PT_ERMIN], px = e0p[PT_EPMAX], rx = e0p[PT_ERMAX], sl = e0p[PT_S];
        const double e0v[9] = {pm, pm, -pm, -pm, rmn, rmn, -rmn, -rmn, rmn};
        const double sg[9] = {1, 1, -1, -1, 1, 1, -1, -1, 1};
        const double ex[9] = {px, px, px, px, rx, rx, rx, rx, rx};
        const double ab[8] = {e[PT_PUP], e[PT_PUP + 1], -e[PT_PLO], -e[PT_PLO + 1], e[PT_RUP], e[PT_RUP + 1], -e[PT_RLO], -e[PT_RLO + 1]};
        const double Ls = path[(sector + i + 1) * PT_LEN + PT_CUM] - e[PT_CUM];
        double *A4 = p + o_a, *A3 = A4 + 9 * (S + 1), *A2 = A3 + 9 * (S + 1), *A1 = A2 + 9 * (S + 1), *A0 = A1 + 9 * (S + 1);
        for (int ch = 0; ch < 9; ch++) {
            const double scale = ab[ch < 8 ? ch : 7];
            const int id = ch * (S + 1) + i;
            bound_params(Ls, e0v[ch], e0v[ch], sg[ch] * sl * scale, sg[ch] * ex[ch] * scale, A4 + id, A3 + id, A2 + id, A1 + id, A0 + id);
            if (i == S - 1) {                              // row S (np.empty in the reference) := row S-1
                A4[id + 1] = A4[id]; A3[id + 1] = A3[id]; A2[id + 1] = A2[id]; A1[id + 1] = A1[id]; A0[id + 1] = A0[id];
            }
        }
    }
    // ---- phase 1b (all lanes, strided): everything that is a copy ----
    for (int id = lane; id < 21; id += nl) p[id] = rb[RB_Q + id];                       // q0, dq0, ddq0
    for (int id = lane; id < 12; id += nl) p[24 + id] = rb[RB_P + id];                  // p0, v0
    for (int id = lane; id < 3; id += nl) { p[21 + id] = ss[SS_PHI + id]; p[36 + id] = ss[SS_IWREF + id]; p[39 + id] = sh[SH_DTAU + id]; }
    for (int id = lane; id < 7; id += nl) { p[o_jerk + id] = rb[RB_JERK + id]; p[o_qd + id] = (phimax_p - phi_cur < 0.05) ? q0[id] : 0.0; }   // :411-413
    for (int id = lane; id <= S; id += nl) p[o_sw + id] = path[(sector + id) * PT_LEN + PT_CUM];
    for (int id = lane; id < 9; id += nl) { const int col = id / 3, r = id % 3; p[o_jr + id] = sh[SH_JR + r * 3 + col]; p[o_jl + id] = sh[SH_JL + r * 3 + col]; }
    for (int id = lane; id < 6 * S; id += nl) {
        const int k = id / S, i = id % S; const double *e = path + (sector + i) * PT_LEN;
        p[o_pref + id] = e[k < 3 ? PT_P + k : PT_IW + k - 3]; p[o_dpref + id] = e[k < 3 ? PT_DPN + k : PT_DR + k - 3];
    }
    for (int id = lane; id < 12 * S; id += nl) {
        const int f = id / (3 * S), r = id % (3 * S), k = r / S, i = r % S;
        p[o_b + id] = path[(sector + i) * PT_LEN + (f == 0 ? PT_BP1 : (f == 1 ? PT_BP2 : (f == 2 ? PT_BR1 : PT_BR2))) + k];
    }
    for (int id = lane; id < 15; id += nl) {
        double wv = ss[SS_W + id];
        if (id == 6 && rb[RB_XPHID] < 1) wv *= BMPC_FMIN(1 / (phi_max * phi_max), 2.0);      // :400-403
        p[o_w + id] = wv;
    }
    if (lane == 0) {
        p[o_xphid] = BMPC_FMIN(phi_cur + 5.0, rb[RB_XPHID]); p[o_xphid + 1] = rb[RB_XPHID + 1]; p[o_xphid + 2] = rb[RB_XPHID + 2];
        p[o_jerk + 7] = ss[SS_DDDPHI];
        p[o_pm] = phimax_p; p[o_pm + 1] = ss[SS_W + 4];                                     // phi_max (clipped), dphi_max = weights[4]
    }
    // ---- warm-start vector: cold start :316-321, or previous plan with omega unwrap :326-333 and shift :372-375 ----
    if (!has_prev) {
        for (int id = lane; id < 44 * N; id += nl) {
            const int i = id % 44;
            x0[id] = (i >= 8 && i < 15) ? q0[i - 8] : ((i >= 29 && i < 35) ? p0[i - 29] : 0.0);
        }
    } else {
        const bool updated = ss[ss_updated(N)] > 0.5;                      // after update(): no shift, re-projection below (:335-375)
        const double *pv = (xlast && !updated && ss[ss_updated(N) + 1] > 0.5) ? xlast : ss + SS_PREV;
        const double d[3] = {p0[3] - pv[32], p0[4] - pv[33], p0[5] - pv[34]};
        const bool unwrap = norm3(d) > 1.5;
        for (int id = lane; id < 44 * N; id += nl) {
            const int k = id / 44, i = id % 44;
            const int kk = updated ? k : (k + 1 < N ? k + 1 : N - 1);      // shift: row k takes (unwrapped) row k+1
            double v = pv[kk * 44 + i];
            if (unwrap && i >= 32 && i < 35) { const int ks = kk < N - 1 ? kk : N - 2; v = p0[i - 29] + (pv[(ks + 1) * 44 + i] - pv[i]); }
            x0[id] = v;
        }
        if (updated) {
            BMPCS_SYNC();
            // path-parameter states of every stage re-projected onto the first segment of the (new) window from the Cartesian
            // position / velocity / acceleration / jerk of the previous plan (BoundMPC.py:335-369); one lane per stage
            const double *e0 = path + sector * PT_LEN, *pc = ss + ss_pc(N);
            const double sw0 = e0[PT_CUM], sw1 = path[(sector + 1) * PT_LEN + PT_CUM];
            for (int k = lane; k < N; k += nl) {
                double phik = sw0, dphik = 0, ddphik = 0, dddphik = 0;
                for (int c = 0; c < 3; c++) {
                    const double dn = e0[PT_DPN + c];
                    phik += (pc[c * N + k] - e0[PT_P + c]) * dn; dphik += pc[(3 + c) * N + k] * dn;
                    ddphik += pc[(6 + c) * N + k] * dn; dddphik += pc[(9 + c) * N + k] * dn;
                }
                double *z = x0 + k * 44;
                if (phik > sw1 - 0.01) { z[41] = sw1 - 0.01; z[42] = 0.0; z[43] = 0.0; }
                else if (phik < 0) {
                    for (int j = 0; j < 7; j++) z[8 + j] = q0[j];
                    z[41] = 0.0; z[42] = 0.0; z[43] = 0.0;
                    for (int c = 0; c < 6; c++) z[29 + c] = p0[c];
                    for (int c = 0; c < 4; c++) z[35 + c] = 0.0;
                } else { z[41] = phik; z[42] = dphik; z[43] = ddphik; z[7] = dddphik; }
            }
        }
        if (!updated && dual && dual[57 * N] > 0.0)
            for (int i = lane; i < 57; i += nl) for (int k = 0; k < N - 1; k++) dual[k * 57 + i] = dual[(k + 1) * 57 + i];
    }
    BMPCS_SYNC();
    if (lane == 0) ss[SS_SECTOR] = (double)sector;
    // Barrier level of the stream's NEXT solve (bmpc_stream_set_level_rule; handles that hold the level, bmpc_set_barrier_hold): a stream far from the end
    // of its path runs on the robust level lvl_hi, near the end -- where the barrier of phi <= phi_max would stall it short of the goal -- on
    // clamp(lvl_c (phi_max - phi), lvl_lo, lvl_hi).  Written into the mu slot of a WARM dual state (a cold one, mu <= 0, starts on mu_init).
    if (lane == 0 && dual && lvl_hi > 0.0 && dual[57 * N] > 0.0) {
        const double lv = lvl_c * (phi_max - phi_cur);
        dual[57 * N] = lv < lvl_lo ? lvl_lo : (lv > lvl_hi ? lvl_hi : lv);
    }
}

// f2 + f3: post-process one stream.  x [N][44] solver result, g [N][43], status; traj: trajectory record (tr_len(N));
// flags bit 0: advance the robot record rb with the kinematic plant step of the node (util_functions.py:152-161);
// bit 1: real-time-iteration mode (not in the reference): the solver runs a fixed, small number of iterations per tick, so status 1 is the
// normal outcome.  The reference's acceptance rule (BoundMPC.py:460-465: solver success OR summed violation of g beyond 1e-6 below
// 1e-4) still decides -- with the threshold rt_tol in place of 1e-4 (bmpc_stream_set_rt_feasibility_tol; default 1e-4, the
// reference's).  A capped iterate that fails it is NOT applied: the previous plan is replayed from its error count, as the reference
// does after a failed solve (:468-489).  (Round 2 accepted every capped iterate unconditionally; closed loops then ran away.)  Since
// round 4 a plan with any variable outside its bounds (below) is not accepted either (for thresholds below 1): no accepted plan leaves the joint limits.
BMPC_HD inline void stream_post(int N, int S, double h, const double *path, int cap, double *ss, double *rb, const double *x, const double *g, int status,
                                double *traj, int flags, double rt_tol, double *sh, int lane, int nl, double rt_row_cap = 0.0) {
    // ---- phase 0: feasibility rule :460-465 (strided partial sums, fixed-order total) ----
    {
        double part = 0.0;
        for (int id = lane; id < 43 * N; id += nl) {
            const double v = g[id]; const int i = id % 43;
            if (i < 36 && v < -1e-6) part -= v;
            if (v > 1e-6) part += v;
            if (!(v - v == 0.0)) part += 1e6;      // a non-finite constraint value is a violation, not a pass
            // Real-time mode, position tube rows (39, 40: l^2 - w^2 of every stage) held PER ROW (bmpc_stream_set_rt_position_row_cap; 0 = off): the summed rule
            // lets a single row through by up to the threshold, and a stream that later REPLAYS the tail of such a plan (:468-489) carries the plant out of a
            // narrow tube by millimetres (round 6: 5.1 mm at a half width of 18 mm, one sample of 32 636, threshold 1e-2).  A row above the cap vetoes the iterate.
            if ((flags & 2) && rt_row_cap > 0.0 && (i == 39 || i == 40) && v > rt_row_cap) part += 1e6;
        }
        // Real-time mode (flag bit 1) also counts the violation of the VARIABLE bounds lbx <= x <= ubx of the plan (jerks +-35, joint
        // positions and velocities RobotModel.py:20-43, phi >= 0).  The reference's rule looks at g only: an Ipopt iterate satisfies the
        // variable bounds by construction, an iteration-capped iterate of this solver (bounds as slack rows) need not -- round 3's
        // capped closed loops left the joint limits on plans that passed the g rule (profiles/r03_rt_safety_modes.log).
        if (flags & 2) {
            const double qd[7] = {165, 115, 165, 115, 165, 115, 170}, dqd[7] = {85, 85, 100, 75, 130, 135, 135};
            for (int id = lane; id < 44 * N; id += nl) {
                const int i = id % 44; const double v = x[id];
                double lim = -1.0;
                if (i < 8) lim = 35.0; else if (i < 15) lim = qd[i - 8] * (3.14159265358979323846 / 180.0); else if (i < 22) lim = dqd[i - 15] * (3.14159265358979323846 / 180.0);
                // strict: a plan with ANY entry outside its bounds counts as grossly infeasible (+1e6: recognisable in the reported violation).
                // The tests are written so that a NaN entry FAILS them, with the slack of the solver's own bound rows (a converged iterate
                // sits up to ~tol outside an active bound: 1e-9) -- a converged plan with an active jerk or phi bound is not vetoed
                if (lim > 0.0 && !(v <= lim + 1e-9 && v >= -lim - 1e-9)) part += 1e6;
                if (i == 41 && !(v >= -1e-9)) part += 1e6;
                if (lim <= 0.0 && i != 41 && !(v - v == 0.0)) part += 1e6;      // non-finite entries of the unbounded variables
            }
            // ... and the trajectory the plant would actually follow: the joint chains re-integrated from the measured state with the plan's
            // jerks (phase 1 below does the same for the return data).  A plan may satisfy its own bounds and still be off its dynamics rows
            // by up to the threshold: the re-integrated q, dq then leave the limits by that much (round 4, measured: 1e-5 ... 1e-4 rad).
            for (int j = lane; j < 7; j += nl) {
                double a = rb[RB_Q + j], da = rb[RB_DQ + j], dda = rb[RB_DDQ + j], up = rb[RB_JERK + j];
                const double ql = qd[j] * (3.14159265358979323846 / 180.0), dql = dqd[j] * (3.14159265358979323846 / 180.0);
                for (int i = 0; i < N; i++) {
                    const double u = x[i * 44 + j];
                    chain_step(a, da, dda, up, u, h); up = u;
                    if (!(a <= ql && a >= -ql && da <= dql && da >= -dql)) part += 1e6;      // (a NaN fails)
                }
            }
        }
        sh[SH_RED + lane] = part;
    }
    BMPCS_SYNC();
    if (lane == 0) {
        double viol = 0.0;
        for (int l = 0; l < nl; l++) viol += sh[SH_RED + l];
        // (real-time mode: a bound violation of the plan or of the trajectory the plant would follow shows as >= 1e6 in viol and also vetoes an
        // iterate the solver calls converged -- at the loose tolerance of the real-time modes "converged" leaves bound rows open by up to that)
        const bool success = (flags & 2) ? ((status == 0 && viol < 1e6) || viol < rt_tol) : (status == 0 || viol < 1e-4);
        int ec = (int)ss[SS_ERRCNT], using_prev = 0, use_prev_plan = 0;
        if (!success) {
            ec += 1; using_prev = 1;
            if (ss[SS_HASPREV] > 0.5) use_prev_plan = 1; else ec = 0;
        } else ec = 0;
        sh[SH_FLAG + 0] = success ? 1.0 : 0.0; sh[SH_FLAG + 1] = (double)ec; sh[SH_FLAG + 2] = (double)using_prev; sh[SH_FLAG + 3] = (double)use_prev_plan;
        sh[SH_FLAG + 4] = viol;
    }
    BMPCS_SYNC();
    const bool success = sh[SH_FLAG + 0] > 0.5;
    const int ec = (int)sh[SH_FLAG + 1], using_prev = (int)sh[SH_FLAG + 2];
    double *prev = ss + SS_PREV;
    const double *w = sh[SH_FLAG + 3] > 0.5 ? prev : x;          // plan used for the return data
    const int TRN = tr_len(N);
    const int n = N - ec;
    double *Tq = traj, *Tdq = Tq + 7 * N, *Tddq = Tdq + 7 * N, *Tj = Tddq + 7 * N, *Tp = Tj + 7 * N, *Tv = Tp + 6 * N, *Ta = Tv + 6 * N,
           *Tphi = Ta + 6 * N, *Tdphi = Tphi + N, *Tddphi = Tdphi + N, *Tjphi = Tddphi + N;
    const double phi_before = ss[SS_PHI];
    // ---- phase 1: store the plan; one lane per integrator chain re-integrates it from the optimal jerks :536-555 ----
    if (success) for (int id = lane; id < 44 * N; id += nl) prev[id] = x[id];
    if (ec < N) {
        for (int j = lane; j < 8; j += nl) {
            double a, da, dda, up;
            if (j < 7) { a = rb[RB_Q + j]; da = rb[RB_DQ + j]; dda = rb[RB_DDQ + j]; up = rb[RB_JERK + j]; }
            else { a = ss[SS_PHI]; da = ss[SS_DPHI]; dda = ss[SS_DDPHI]; up = ss[SS_DDDPHI]; }
            for (int i = 0; i < n; i++) {
                const double u = w[(ec + i) * 44 + j];
                chain_step(a, da, dda, up, u, h); up = u;
                if (j < 7) { Tq[j * N + i] = a; Tdq[j * N + i] = da; Tddq[j * N + i] = dda; Tj[j * N + i] = u; }
                else { Tphi[i] = a; Tdphi[i] = da; Tddphi[i] = dda; Tjphi[i] = u; }
            }
        }
    }
    BMPCS_SYNC();
    if (lane == 0) {
        ss[SS_ERRCNT] = (double)ec; ss[SS_USINGPREV] = (double)using_prev; ss[SS_VALID] = ec < N ? 1.0 : 0.0;
        ss[ss_updated(N) + 1] = ((flags & 2) && status != 3) ? 1.0 : 0.0;      // real-time mode: the next warm start may continue from this iterate (stream_pack)
        if (success) ss[SS_HASPREV] = 1.0;
        traj[TRN - 4] = ec < N ? (double)n : 0.0; traj[TRN - 3] = (double)using_prev; traj[TRN - 2] = success ? 1.0 : 0.0; traj[TRN - 1] = sh[SH_FLAG + 4];
    }
    if (ec >= N) return;                                          // step() returns None :504-506 (uniform over the lanes)
    // ---- phase 2: roles 0..n-1: Cartesian trajectory of one stage :568-587; role RF: rotation reference and path-parameter
    //      state :594-611; role RF+1: the node's kinematic plant step (util_functions.py:152-161); RF = max(N, 32) ----
    // Roles RF+2..RF+2+ec-1 (only while a previous plan is replayed): the leading, already executed columns of that plan, of which
    // only the Cartesian derivatives for a later re-planning are kept.
    double *pc = ss + ss_pc(N);
    const int RF = N > 32 ? N : 32;
    for (int role = lane; role < RF + 2 + ec; role += nl) {
        if (role < n || role >= RF + 2) {
            const bool lead = role >= RF + 2;
            const int i = lead ? 0 : role, col = lead ? role - (RF + 2) : ec + role;        // trajectory index / column of the plan
            double q[7], dq[7], ddq[7], u[7];
            for (int j = 0; j < 7; j++) {
                if (lead) { q[j] = w[col * 44 + 8 + j]; dq[j] = w[col * 44 + 15 + j]; ddq[j] = w[col * 44 + 22 + j]; }
                else { q[j] = Tq[j * N + i]; dq[j] = Tdq[j * N + i]; ddq[j] = Tddq[j * N + i]; }
                u[j] = w[col * 44 + j];
            }
            FkMotion M; fk_motion(q, dq, ddq, u, M);
            if (!lead) for (int c = 0; c < 6; c++) { Tp[c * N + i] = M.p[c]; Tv[c * N + i] = M.v[c]; Ta[c * N + i] = M.a[c]; }
            // previous-plan arrays of BoundMPC.py:557-566 (linear rows): position of the re-integrated plan (the plan's own entry for
            // a leading column), the SOLVER's velocity variables, J ddq + dJ dq, J u + dJ ddq + ddJ dq
            for (int c = 0; c < 3; c++) {
                // unconditional load (col is a valid column in both roles), then a select; the value is made opaque so that the compiler does not sink
                // the load back under the exec mask of the `lead` lanes (a masked load whose result is read behind the join is what build.py's lint flags)
                double wpos = w[col * 44 + 29 + c]; BMPCS_OPAQUE(wpos);
                pc[c * N + col] = lead ? wpos : M.p[c];
                pc[(3 + c) * N + col] = w[col * 44 + 35 + c];
                pc[(6 + c) * N + col] = M.a[c]; pc[(9 + c) * N + col] = M.jk[c];
            }
        } else if (role == RF) {
            int sector = (int)ss[SS_SECTOR];
            sector = sector > cap - 2 ? cap - 2 : sector; sector = sector < 0 ? 0 : sector;       // clamped to the table (see stream_pack)
            const double *e0 = path + sector * PT_LEN, *e1 = path + (sector + 1) * PT_LEN;
            const double sw0 = e0[PT_CUM], sw1 = e1[PT_CUM], phi0 = Tphi[0];
            double prn[3];
            if (phi0 > sw1) {
                integrate_rotation_reference(e1 + PT_RRV, e1 + PT_DR, sw1, phi0, prn);
                for (int c = 0; c < 3; c++) ss[SS_IWREF + c] = e1[PT_IW + c] + (phi0 - sw1) * e1[PT_DR + c];
            } else {
                integrate_rotation_reference(ss + SS_PRREF, e0 + PT_DR, phi_before, phi0, prn);
                for (int c = 0; c < 3; c++) ss[SS_IWREF + c] = e0[PT_IW + c] + (phi0 - sw0) * e0[PT_DR + c];
            }
            for (int c = 0; c < 3; c++) ss[SS_PRREF + c] = prn[c];
            ss[SS_PHI] = phi0; ss[SS_DPHI] = Tdphi[0]; ss[SS_DDPHI] = Tddphi[0]; ss[SS_DDDPHI] = Tjphi[0];
        } else if (role == RF + 1 && (flags & 1)) {
            // integrate with [jerk_current, first planned jerk], then FK
            double qs[7], dqs[7], ddqs[7];
            for (int j = 0; j < 7; j++) { qs[j] = rb[RB_Q + j]; dqs[j] = rb[RB_DQ + j]; ddqs[j] = rb[RB_DDQ + j]; chain_step(qs[j], dqs[j], ddqs[j], rb[RB_JERK + j], Tj[j * N], h); }
            double u0[7];
            for (int j = 0; j < 7; j++) u0[j] = Tj[j * N];
            FkMotion M; fk_motion(qs, dqs, ddqs, u0, M);      // (pose and v = J dq are what is kept)
            for (int j = 0; j < 7; j++) { rb[RB_Q + j] = qs[j]; rb[RB_DQ + j] = dqs[j]; rb[RB_DDQ + j] = ddqs[j]; rb[RB_JERK + j] = u0[j]; }
            for (int c = 0; c < 6; c++) { rb[RB_P + c] = M.p[c]; rb[RB_V + c] = M.v[c]; }
        }
    }
}

}  // namespace bmpcs
