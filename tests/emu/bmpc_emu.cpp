// TEST-ONLY lane emulator of the wave program (boundmpc_amd/csrc/bmpc_wave.inl).
//
// Compiles the SAME kernel text with g++ and executes each phase as a loop over the 64 lanes
// in a caller-chosen order.  It exists so the kernel's indexing, phase structure and numerics
// can be unit-tested in a container without a GPU (pytest -m "not gpu"), and so that
// intra-phase cross-lane dependences show up as order-dependent results.  It is NOT part of
// the product: boundmpc_amd never builds, loads or falls back to it.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BMPC_EMU 1
#define BMPC_HD
#define BMPC_D
#define BMPC_SINCOS(x, s, c) (*(s) = std::sin(x), *(c) = std::cos(x))
#define BMPC_EXP(x) std::exp(x)
#define BMPC_LOG(x) std::log(x)
#define BMPC_SQRT(x) std::sqrt(x)
#define BMPC_SIN(x) std::sin(x)
#define BMPC_COS(x) std::cos(x)
#define BMPC_ATAN2(y, x) std::atan2(y, x)
#define BMPC_RSQRT(x) (1.0 / std::sqrt(x))
#define BMPC_FABS(x) std::fabs(x)
#define BMPC_FMAX(a, b) std::fmax(a, b)
#define BMPC_FMIN(a, b) std::fmin(a, b)
#define BMPC_POW15(x) ((x) * std::sqrt(x))
#define BMPC_POW(x, y) std::pow(x, y)
#define LANES_BEGIN for (int li_ = 0; li_ < 64; ++li_) { const int lane = W.order[li_]; (void)lane;
#define LANES_END }
#define LIDX lane

#include "../../boundmpc_amd/csrc/bmpc_wave.inl"
#define BMPCS_SYNC()
#include "../../boundmpc_amd/csrc/bmpc_stream.inl"

// CPU build of the stream functions (same text as the device kernels), one call per stream
extern "C" void bmpc_emu_stream_lengths(int N, int *out) { out[0] = bmpcs::PT_LEN; out[1] = bmpcs::ss_len(N); out[2] = bmpcs::RB_LEN; out[3] = bmpcs::tr_len(N); }
extern "C" void bmpc_emu_stream_pack(int N, int S, const double *path, double *ss, const double *rb, double *p, double *x0, double *dual, const double *xlast, double lvl_c, double lvl_lo, double lvl_hi) {
    double sh[bmpcs::SH_LEN];
    bmpcs::stream_pack(N, S, path, (int)ss[bmpcs::SS_NENT], ss, rb, p, x0, dual, xlast, sh, 0, 1, lvl_c, lvl_lo, lvl_hi);
}
extern "C" void bmpc_emu_stream_post(int N, int S, double h, const double *path, double *ss, double *rb, const double *x, const double *g, int status,
                                     double *traj, int simulate, double rt_tol, double rt_row_cap) {
    double sh[bmpcs::SH_LEN];
    bmpcs::stream_post(N, S, h, path, (int)ss[bmpcs::SS_NENT], ss, rb, x, g, status, traj, simulate, rt_tol, sh, 0, 1, rt_row_cap);
}

extern "C" void bmpc_emu_jacobian_lin_ddot(const double *q, const double *dq, const double *ddq, double *out) { bmpcs::jacobian_lin_ddot(q, dq, ddq, out); }
// fk_motion: out = [p 6 | v 6 | a 6 | jk 3]
extern "C" void bmpc_emu_fk_motion(const double *q, const double *dq, const double *ddq, const double *u, double *out) {
    bmpcs::FkMotion M; bmpcs::fk_motion(q, dq, ddq, u, M);
    for (int c = 0; c < 6; c++) { out[c] = M.p[c]; out[6 + c] = M.v[c]; out[12 + c] = M.a[c]; }
    for (int c = 0; c < 3; c++) out[18 + c] = M.jk[c];
}

extern "C" int bmpc_emu_solve(int N, int S, double h, const bmpc::Opts *opts, int B, const double *p, const double *x0, double *state, double *x, double *g,
                              double *lam_g, double *lam_x, double *f, int *iters, int *status, double *kkt, int lane_order, int nthreads) {
    if (S > bmpc::SMAX || S < 2 || N < 1 || N > bmpc::NMAX) return 1;
    const bmpc::Scr sc = bmpc::make_scr(N);
    const int np = 141 + 91 * S, nw = N * bmpc::NZ, ng = N * bmpc::NG;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const bool poison = getenv("BMPC_EMU_POISON") != nullptr;
    const bool inkernel = getenv("BMPC_EMU_INKERNEL") != nullptr;
#pragma omp parallel
    {
        std::vector<double> lds(bmpc::L_SIZE, 0.0), scr(sc.size, 0.0);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; b++) {
            // BMPC_EMU_POISON=1: LDS and workspace are filled with NaN before every problem -- a read of something this solve has not
            // written (what a reused slab or LDS holds on the GPU) then shows up in the outputs
            if (poison) { std::fill(lds.begin(), lds.end(), std::nan("")); std::fill(scr.begin(), scr.end(), std::nan("")); }
            bmpc::Wave W; W.N = N; W.S = S; W.h = h; W.o = *opts; W.L = lds.data(); W.G = bmpc::make_gptr(scr.data()); W.it_base = 0;
            for (int i = 0; i < 64; i++) W.order[i] = lane_order == 0 ? i : (lane_order == 1 ? 63 - i : (i * 37 + 11) % 64);
            bmpc::Problem pr;
            pr.p = p + (size_t)b * np; pr.x0 = x0 + (size_t)b * nw;
            pr.x = x ? x + (size_t)b * nw : nullptr; pr.g = g ? g + (size_t)b * ng : nullptr;
            pr.lam_g = lam_g ? lam_g + (size_t)b * ng : nullptr; pr.lam_x = lam_x ? lam_x + (size_t)b * nw : nullptr;
            pr.f = f ? f + b : nullptr; pr.kkt = kkt ? kkt + b : nullptr; pr.iters = iters ? iters + b : nullptr; pr.status = status ? status + b : nullptr;
            pr.state = state ? state + (size_t)b * (N * bmpc::NI + 2) : nullptr;
            pr.resto_from = -1;
            const bool zl = N <= 11 && S <= bmpc::SMAX_ZLDS;
            if (inkernel) {      // the restoration phase inside the kernel (what the fused closed-loop ticks run)
                if (zl) bmpc::wave_solve<true, false, true>(W, pr); else bmpc::wave_solve<false, false, true>(W, pr);
            } else {             // the batch kernels: main phase only; a jammed problem (internal status 4) is continued by the restoration kernel from its iterate
                std::vector<double> xb(nw); int it_ = 0, st_ = 0;
                bmpc::Problem q = pr; q.x = xb.data(); q.iters = &it_; q.status = &st_;
                if (zl) bmpc::wave_solve_retry<true>(W, q); else bmpc::wave_solve_retry<false>(W, q);      // (the batch kernels' call: with the second attempt of a status-2 solve)
                if (st_ == 4) {
                    std::vector<double> x0b(xb);
                    q.x0 = x0b.data(); q.resto_from = it_;
                    if (zl) bmpc::wave_solve_retry<true, false, true>(W, q, pr.x0); else bmpc::wave_solve_retry<false, false, true>(W, q, pr.x0);      // (the restoration kernel's call)
                }
                if (pr.x) memcpy(pr.x, xb.data(), sizeof(double) * nw);
                if (pr.iters) *pr.iters = it_;
                if (pr.status) *pr.status = st_;
            }
        }
    }
    return 0;
}
// debug: one Newton direction at (x, t, nu, mu); dumps the scratch slab
extern "C" int bmpc_emu_newton(int N, int S, double h, const bmpc::Opts *opts, const double *p, const double *x, const double *t, const double *nu,
                               double mu, double delta, double *scratch_out, double *lds_out) {
    using namespace bmpc;
    const Scr sc = make_scr(N); const POff po = make_poff_lds(S, L_ZL);
    std::vector<double> lds(L_SIZE, 0.0), scr(sc.size, 0.0);
    Wave W; W.N = N; W.S = S; W.h = h; W.o = *opts; W.L = lds.data(); W.G = bmpc::make_gptr(scr.data()); W.it_base = 0;
    for (int i = 0; i < 64; i++) W.order[i] = i;
    for (int i = 0; i < po.size; i++) W.L[L_PAR + lds_index_of_p(S, i, L_ZL)] = p[i];
    wave_init_tables(W, po);
    const bool zl = N <= 11 && S <= bmpc::SMAX_ZLDS; W.Zc = zl ? W.L + L_ZL : (W.G + sc.Z).ptr(); W.Zt = zl ? W.L + L_PB : (W.G + sc.ZT).ptr(); W.Dz = zl ? W.L + L_PB + 512 : (W.G + sc.DZ).ptr();
    for (int i = 0; i < N * NZ; i++) W.Zc[i] = x[i];
    for (int i = 0; i < N * NI; i++) { W.G[sc.T + i] = t[i]; W.G[sc.NUm + i] = nu[i]; }
    wave_eval(W, po, sc, W.Zc, sc.G, sc.HIN, false);
    LaneRegs LRs[64];
    wave_adjoint(W, po, sc, sc.NUm, false, 0.0, LRs);
    for (int i = 0; i < N * NI; i++) { const double tt = W.G[sc.T + i], nn = W.G[sc.NUm + i]; W.G[sc.SG + i] = nn / tt; W.G[sc.TI + i] = 1.0 / tt; W.G[sc.SR + i] = nn / tt * (W.G[sc.HIN + i] + tt); }
    wave_adjoint(W, po, sc, sc.NUm, true, mu, LRs);
    wave_stage_data_wide(W, po, sc);
    bool ok = wave_backward_blk(W, po, sc, mu, delta, LRs);
    if (ok) wave_forward(W, sc, LRs);
    if (ok) for (int i = 0; i < N * NZ; i++) W.G[sc.DZ + i] = W.Dz[i];
    memcpy(scratch_out, scr.data(), sizeof(double) * sc.size);
    memcpy(lds_out, lds.data(), sizeof(double) * L_SIZE);
    return ok ? 0 : 3;
}
extern "C" void bmpc_emu_scr_offsets(int N, int *out) {
    const bmpc::Scr s = bmpc::make_scr(N);
    int v[] = {s.Z, s.ZT, s.T, s.TT, s.NUm, s.LAM, s.G, s.GT, s.HIN, s.HT, s.DZ, s.DT, s.DNU, s.GH, s.GVP, s.RJ, s.KIN, s.REF, s.KT, s.KF, s.RDY, s.AES, s.RLV, s.SG, s.TI, s.SR, s.size};
    for (unsigned i = 0; i < sizeof(v) / sizeof(int); i++) out[i] = v[i];
}
extern "C" void bmpc_emu_sincos(int n, const double *x, double *s, double *c) { for (int i = 0; i < n; i++) bmpc::bmpc_sincos(x[i], s + i, c + i); }
extern "C" int bmpc_emu_lds_doubles() { return bmpc::L_SIZE; }
extern "C" int bmpc_emu_scratch_doubles(int N) { return bmpc::make_scr(N).size; }
