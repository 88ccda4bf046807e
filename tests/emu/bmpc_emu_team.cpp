// TEST-ONLY lane emulator of the TEAM variant of the wave program (boundmpc_amd/csrc/bmpc_wave.inl with BMPC_NW waves per problem).
//
// Same idea as bmpc_emu.cpp: the kernel text compiled by g++, a phase = a loop over lanes.  A wide phase (WIDE_BEGIN ... WIDE_END) is a
// loop over the NW waves of the team (in a caller-chosen order) times their 64 lanes; a solo region runs with the wave index it names.
// The waves of the GPU run a wide phase concurrently and meet at the barrier behind it: any order of the waves inside a phase must give the
// same result, which is what running forward / reverse / scrambled wave AND lane orders checks.  What it cannot see are missing
// barriers BETWEEN phases (a wave racing ahead): those are argued in the kernel text (TEAM_SYNC comments) and tested on the GPU.
// Never built or loaded by the product.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef BMPC_NW
#define BMPC_NW 4
#endif
#define BMPC_NAMESPACE bmpct
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
#define LIDXW wl
#define TEAM_SYNC()
#define TEAM_SYNC_LDS()
#define WIDE_BEGIN for (int wi_ = 0; wi_ < BMPC_NW; ++wi_) { W.wv = W.worder[wi_]; LANES_BEGIN const int wl = W.wv * 64 + lane; (void)wl;
#define WIDE_END LANES_END } W.wv = 0;
#define SOLO_BEGIN(w) { W.wv = (w);
#define SOLO_END W.wv = 0; }

#include "../../boundmpc_amd/csrc/bmpc_wave.inl"

extern "C" int bmpc_emu_team_waves() { return BMPC_NW; }
extern "C" int bmpc_emu_team_lds_doubles() { return bmpct::L_SIZE; }
extern "C" int bmpc_emu_team_solve(int N, int S, double h, const bmpct::Opts *opts, int B, const double *p, const double *x0, double *state, double *x, double *g,
                                   double *lam_g, double *lam_x, double *f, int *iters, int *status, double *kkt, int lane_order, int wave_order, int nthreads) {
    if (S > bmpct::SMAX || S < 2 || N < 1 || N > bmpct::NMAX) return 1;
    const bmpct::Scr sc = bmpct::make_scr(N);
    const int np = 141 + 91 * S, nw = N * bmpct::NZ, ng = N * bmpct::NG;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const bool poison = getenv("BMPC_EMU_POISON") != nullptr;
#pragma omp parallel
    {
        std::vector<double> lds(bmpct::L_SIZE, 0.0), scr(sc.size, 0.0);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; b++) {
            if (poison) { std::fill(lds.begin(), lds.end(), std::nan("")); std::fill(scr.begin(), scr.end(), std::nan("")); }
            bmpct::Wave W; W.N = N; W.S = S; W.h = h; W.o = *opts; W.L = lds.data(); W.G = bmpct::make_gptr(scr.data()); W.wv = 0; W.it_base = 0;
            for (int i = 0; i < 64; i++) W.order[i] = lane_order == 0 ? i : (lane_order == 1 ? 63 - i : (i * 37 + 11) % 64);
            for (int i = 0; i < BMPC_NW; i++) W.worder[i] = wave_order == 0 ? i : (wave_order == 1 ? BMPC_NW - 1 - i : (i * 3 + 1) % BMPC_NW);
            bmpct::Problem pr;
            pr.p = p + (size_t)b * np; pr.x0 = x0 + (size_t)b * nw;
            pr.x = x ? x + (size_t)b * nw : nullptr; pr.g = g ? g + (size_t)b * ng : nullptr;
            pr.lam_g = lam_g ? lam_g + (size_t)b * ng : nullptr; pr.lam_x = lam_x ? lam_x + (size_t)b * nw : nullptr;
            pr.f = f ? f + b : nullptr; pr.kkt = kkt ? kkt + b : nullptr; pr.iters = iters ? iters + b : nullptr; pr.status = status ? status + b : nullptr;
            pr.state = state ? state + (size_t)b * (N * bmpct::NI + 2) : nullptr;
            pr.resto_from = -1;      // (the team text with the restoration phase inside, as the fused team tick runs it; the team BATCH kernel hands jammed problems to the one-wave restoration kernel)
            if (N <= 11 && S <= bmpct::SMAX_ZLDS) bmpct::wave_solve<true, false, true>(W, pr); else bmpct::wave_solve<false, false, true>(W, pr);
        }
    }
    return 0;
}
