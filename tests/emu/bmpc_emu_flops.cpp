// TEST-ONLY (never built or loaded by the product): flop-counting build of the lane emulator.  The kernel text
// (boundmpc_amd/csrc/bmpc_wave.inl) is compiled by g++ with `double` replaced by a one-word class whose operators count what they
// execute, summed over the 64 lanes of every phase: the fp64 operations the HIP kernel's OWN algorithm executes per interior-point
// iteration (lanes that a phase switches off by `if (lane < n)` do not count; predicated phases, which evaluate every role in every
// lane, do).  Convention as oracle/flopcount.cpp: add/sub/mul 1, a*b+c 2, division / root / transcendental 1 (tallied separately).
// Phases: the BMPC_PROF stamps of the wave program (the ids tests/gpu_profile_phases.py names) close a phase.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>
namespace fc {
static unsigned long long g_flops = 0, g_special = 0, g_phase[32], g_mark = 0;
struct Real {
    double v;
    Real() = default;
    constexpr Real(double x) : v(x) {}
    constexpr Real(int x) : v((double)x) {}
    explicit operator double() const { return v; }
    explicit operator int() const { return (int)v; }
};
static inline void c1() { g_flops++; }
static inline void cs() { g_flops++; g_special++; }
static inline Real operator+(Real a, Real b) { c1(); return Real(a.v + b.v); }
static inline Real operator-(Real a, Real b) { c1(); return Real(a.v - b.v); }
static inline Real operator*(Real a, Real b) { c1(); return Real(a.v * b.v); }
static inline Real operator/(Real a, Real b) { cs(); return Real(a.v / b.v); }
static inline Real operator-(Real a) { return Real(-a.v); }
static inline Real &operator+=(Real &a, Real b) { c1(); a.v += b.v; return a; }
static inline Real &operator-=(Real &a, Real b) { c1(); a.v -= b.v; return a; }
static inline Real &operator*=(Real &a, Real b) { c1(); a.v *= b.v; return a; }
static inline bool operator<(Real a, Real b) { return a.v < b.v; }
static inline bool operator>(Real a, Real b) { return a.v > b.v; }
static inline bool operator<=(Real a, Real b) { return a.v <= b.v; }
static inline bool operator>=(Real a, Real b) { return a.v >= b.v; }
static inline bool operator==(Real a, Real b) { return a.v == b.v; }
static inline bool operator!=(Real a, Real b) { return a.v != b.v; }
}
using fc::Real;
using namespace fc;
#define BMPC_EMU 1
#define BMPC_HD
#define BMPC_D
#define BMPC_SINCOS(x, s, c) (fc::cs(), fc::cs(), *(s) = Real(std::sin((x).v)), *(c) = Real(std::cos((x).v)))
#define BMPC_EXP(x) (fc::cs(), Real(std::exp(Real(x).v)))
#define BMPC_LOG(x) (fc::cs(), Real(std::log(Real(x).v)))
#define BMPC_SQRT(x) (fc::cs(), Real(std::sqrt(Real(x).v)))
#define BMPC_SIN(x) (fc::cs(), Real(std::sin(Real(x).v)))
#define BMPC_COS(x) (fc::cs(), Real(std::cos(Real(x).v)))
#define BMPC_ATAN2(y, x) (fc::cs(), Real(std::atan2(Real(y).v, Real(x).v)))
#define BMPC_RSQRT(x) (fc::cs(), Real(1.0 / std::sqrt(Real(x).v)))
#define BMPC_FABS(x) Real(std::fabs(Real(x).v))
#define BMPC_FMAX(a, b) Real(std::fmax(Real(a).v, Real(b).v))
#define BMPC_FMIN(a, b) Real(std::fmin(Real(a).v, Real(b).v))
#define BMPC_POW15(x) ((x) * BMPC_SQRT(x))
#define BMPC_RINT(x) Real(__builtin_rint(Real(x).v))
#define BMPC_POW(x, y) (fc::cs(), Real(std::pow(Real(x).v, Real(y).v)))
#define LANES_BEGIN for (int li_ = 0; li_ < 64; ++li_) { const int lane = W.order[li_]; (void)lane;
#define LANES_END }
#define LIDX lane
#define BMPC_PROF(W, id) { fc::g_phase[id] += fc::g_flops - fc::g_mark; fc::g_mark = fc::g_flops; }
#define double Real
#include "../../boundmpc_amd/csrc/bmpc_wave.inl"
#undef double

// out[0] = total iterations, out[1] = converged solves, out[2] = flops, out[3] = divisions/roots/transcendentals among them, out[4..35] = flops per phase slot
extern "C" int bmpc_emu_count_flops(int N, int S, double h, const bmpc::Opts *opts, int B, const double *p, const double *x0, unsigned long long *out) {
    if (S > bmpc::SMAX || S < 2 || N < 1 || N > bmpc::NMAX) return 1;
    const bmpc::Scr sc = bmpc::make_scr(N);
    const int np = 141 + 91 * S, nw = N * bmpc::NZ;
    fc::g_flops = fc::g_special = fc::g_mark = 0; for (int i = 0; i < 32; i++) fc::g_phase[i] = 0;
    std::vector<Real> lds(bmpc::L_SIZE, Real(0.0)), scr(sc.size, Real(0.0)), x(nw);
    unsigned long long its = 0, okc = 0;
    for (int b = 0; b < B; b++) {
        bmpc::Wave W; W.N = N; W.S = S; W.h = Real(h); W.o = *opts; W.L = lds.data(); W.G = bmpc::make_gptr(scr.data()); W.it_base = 0;
        for (int i = 0; i < 64; i++) W.order[i] = i;
        bmpc::Problem pr; int it = 0, st = 0;
        pr.p = (const Real *)p + (size_t)b * np; pr.x0 = (const Real *)x0 + (size_t)b * nw;
        pr.x = x.data(); pr.g = nullptr; pr.lam_g = nullptr; pr.lam_x = nullptr; pr.f = nullptr; pr.kkt = nullptr; pr.iters = &it; pr.status = &st; pr.state = nullptr; pr.resto_from = -1;
        if (N <= 11 && S <= bmpc::SMAX_ZLDS) bmpc::wave_solve_retry<true>(W, pr); else bmpc::wave_solve_retry<false>(W, pr);
        its += (unsigned long long)it; okc += st == 0;
    }
    out[0] = its; out[1] = okc; out[2] = fc::g_flops; out[3] = fc::g_special;
    for (int i = 0; i < 32; i++) out[4 + i] = fc::g_phase[i];
    return 0;
}
