/*
 * ORACLE -- test infrastructure only (see bmpc_oracle.c).  Flop-counting build of the CPU oracle:
 * the SAME source text (bmpc_oracle.c is #included below, unmodified) compiled as C++ with `double`
 * replaced by a one-word class whose arithmetic operators count what they execute.  It answers
 * "the builder must replace [the 2.67 MFLOP/iteration estimate] by an exact count from
 * the CPU oracle's instrumented build" -- bench.py's roofline_fp64 is priced with the number this
 * library measures on a sample of the bench batch, not with the survey's dense-matrix model.
 *
 * Counting convention (stated in DESIGN.md 5): one count per executed fp64 add, subtract, multiply,
 * compare-free fmax/fmin/fabs excluded; a division, a square root and each transcendental call
 * (sin, cos, exp, log, pow, atan2) are counted ONCE in `flops` and also tallied separately in
 * `special`, so that a reader can re-price them (a fp64 division is ~25 instructions on gfx950).
 * a*b+c counts 2 (what an FMA instruction is credited with in a peak-FLOP/s figure).
 * Regions: bmpc_oracle.c marks its phases with ORACLE_REGION(id) (a no-op in the normal build).
 *
 * Single-threaded by construction (global counters): the entry points below run the batch in one thread.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

namespace flopcount {
enum { NREG = 8 };
static unsigned long long g_flops[NREG], g_special[NREG];
static int g_region = 0;
struct Real {
    double v;
    Real() = default;
    Real(double x) : v(x) {}
    Real(int x) : v((double)x) {}
    explicit operator double() const { return v; }
    explicit operator int() const { return (int)v; }
    explicit operator bool() const { return v != 0.0; }
};
static_assert(sizeof(Real) == sizeof(double), "Real must be layout-compatible with double (the C ABI passes plain double arrays)");
static inline void c1() { g_flops[g_region]++; }
static inline void cs() { g_flops[g_region]++; g_special[g_region]++; }
static inline Real operator+(Real a, Real b) { c1(); return Real(a.v + b.v); }
static inline Real operator-(Real a, Real b) { c1(); return Real(a.v - b.v); }
static inline Real operator*(Real a, Real b) { c1(); return Real(a.v * b.v); }
static inline Real operator/(Real a, Real b) { cs(); return Real(a.v / b.v); }
static inline Real operator-(Real a) { return Real(-a.v); }
static inline Real operator+(Real a) { return a; }
static inline Real &operator+=(Real &a, Real b) { c1(); a.v += b.v; return a; }
static inline Real &operator-=(Real &a, Real b) { c1(); a.v -= b.v; return a; }
static inline Real &operator*=(Real &a, Real b) { c1(); a.v *= b.v; return a; }
static inline Real &operator/=(Real &a, Real b) { cs(); a.v /= b.v; return a; }
static inline bool operator<(Real a, Real b) { return a.v < b.v; }
static inline bool operator>(Real a, Real b) { return a.v > b.v; }
static inline bool operator<=(Real a, Real b) { return a.v <= b.v; }
static inline bool operator>=(Real a, Real b) { return a.v >= b.v; }
static inline bool operator==(Real a, Real b) { return a.v == b.v; }
static inline bool operator!=(Real a, Real b) { return a.v != b.v; }
static inline bool operator!(Real a) { return a.v == 0.0; }
static inline Real r_sqrt(Real a) { cs(); return Real(::sqrt(a.v)); }
static inline Real r_sin(Real a) { cs(); return Real(::sin(a.v)); }
static inline Real r_cos(Real a) { cs(); return Real(::cos(a.v)); }
static inline Real r_exp(Real a) { cs(); return Real(::exp(a.v)); }
static inline Real r_log(Real a) { cs(); return Real(::log(a.v)); }
static inline Real r_pow(Real a, Real b) { cs(); return Real(::pow(a.v, b.v)); }
static inline Real r_atan2(Real a, Real b) { cs(); return Real(::atan2(a.v, b.v)); }
static inline Real r_fabs(Real a) { return Real(::fabs(a.v)); }
static inline Real r_fmax(Real a, Real b) { return Real(::fmax(a.v, b.v)); }
static inline Real r_fmin(Real a, Real b) { return Real(::fmin(a.v, b.v)); }
static inline bool r_isfinite(Real a) { return ::isfinite(a.v); }
}  // namespace flopcount

using flopcount::Real;
using namespace flopcount;
#define ORACLE_REGION(id) (flopcount::g_region = (id))
#define ORACLE_FLOPCOUNT 1
#define double Real
#define sqrt r_sqrt
#define sin r_sin
#define cos r_cos
#define exp r_exp
#define log r_log
#define pow r_pow
#define atan2 r_atan2
#define fabs r_fabs
#define fmax r_fmax
#define fmin r_fmin
#undef isfinite
#define isfinite r_isfinite
#undef _OPENMP
extern "C" {
#include "bmpc_oracle.c"
}
#undef double
#undef sqrt
#undef sin
#undef cos
#undef exp
#undef log
#undef pow
#undef atan2
#undef fabs
#undef fmax
#undef fmin
#undef isfinite

/* Solve B problems (one thread) and report what was executed: out[0] = total interior-point iterations, out[1] = solves that
 * converged, out[2 + r] = flops of region r, out[2 + NREG + r] = divisions / roots / transcendentals among them (r = 0..7). */
extern "C" int bmpc_oracle_count_flops(int N, int S, double h, const bmpc_oracle_opts *opts, int B, const double *p, const double *x0,
                                       unsigned long long *out) {
    for (int r = 0; r < flopcount::NREG; r++) flopcount::g_flops[r] = flopcount::g_special[r] = 0;
    flopcount::g_region = 0;
    const int nw = N * NZ, np = 141 + 91 * S;
    Real *x = (Real *)malloc(sizeof(Real) * (size_t)nw * B), *f = (Real *)malloc(sizeof(Real) * B), *kkt = (Real *)malloc(sizeof(Real) * B);
    int *iters = (int *)malloc(sizeof(int) * B), *status = (int *)malloc(sizeof(int) * B);
    (void)np;
    const int rc = bmpc_oracle_solve(N, S, Real(h), opts, B, (const Real *)p, (const Real *)x0, x, NULL, NULL, NULL, f, iters, status, kkt, 1);
    unsigned long long its = 0, okc = 0;
    for (int b = 0; b < B; b++) { its += (unsigned long long)iters[b]; okc += status[b] == 0; }
    out[0] = its; out[1] = okc;
    for (int r = 0; r < flopcount::NREG; r++) { out[2 + r] = flopcount::g_flops[r]; out[2 + flopcount::NREG + r] = flopcount::g_special[r]; }
    free(x); free(f); free(kkt); free(iters); free(status);
    return rc;
}
