// TEST-ONLY (never built or loaded by the product): MASK-AWARE flop count of the kernel text.
//
// bmpc_emu_flops.cpp counts every fp64 operation a lane executes.  The kernel text is predicated straight-line code: every lane of a phase
// evaluates every role on clamped indices and only the STORES are conditional (a lane without an item stores to a dummy word, a lane past the end
// of a pass recomputes the last item and stores the same value again).  Those operations are issued on the GPU and counted as "executed" -- they
// are not work.  This build separates the two tallies by data flow: `double` is replaced by a class that carries, beside the value, the id of the
// operation that produced it; every operation is a node {operands, cost, phase slot}.  An operation is USEFUL when its result reaches
//   * a store to a word of LDS / the workspace / an output that is (a) not the dummy word and (b) not written a second time in the same phase
//     (the clamped duplicates), or
//   * a decision: a comparison, or a conversion to an integer (table look-ups, segment selects).
// Reaching is transitive through registers; a stored value starts a new chain (what is computed FROM it later is judged by where THAT goes).
// Output per phase slot (the BMPC_PROF stamps): executed flops, useful flops.  Same conventions as bmpc_emu_flops.cpp (a*b+c = 2).
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <unordered_set>
#include <vector>
namespace fc {
struct Node { uint32_t a, b; uint8_t cost, slot, useful, pad; };
static std::vector<Node> g_nodes(1);                 // node 0 = a leaf (a loaded value, a constant)
static std::vector<uint32_t> g_stack;
static int g_slot = 31;                              // phase slot the operations executed now will be stamped into (set at the PREVIOUS stamp: see BMPC_PROF)
static const char *g_mem_lo[4], *g_mem_hi[4]; static int g_nmem = 0;
static const void *g_dummy = nullptr;
static std::unordered_set<const void *> g_written;  // words stored to in the current phase
static unsigned long long g_dup_stores = 0, g_dummy_stores = 0, g_stores = 0;
static inline bool in_mem(const void *p) { for (int i = 0; i < g_nmem; i++) if ((const char *)p >= g_mem_lo[i] && (const char *)p < g_mem_hi[i]) return true; return false; }
static inline void mark(uint32_t id) {
    if (!id || g_nodes[id].useful) return;
    g_stack.clear(); g_stack.push_back(id);
    while (!g_stack.empty()) {
        const uint32_t n = g_stack.back(); g_stack.pop_back();
        Node &nd = g_nodes[n];
        if (nd.useful) continue;
        nd.useful = 1;
        if (nd.a && !g_nodes[nd.a].useful) g_stack.push_back(nd.a);
        if (nd.b && !g_nodes[nd.b].useful) g_stack.push_back(nd.b);
    }
}
static inline uint32_t op(uint32_t a, uint32_t b, int cost) { g_nodes.push_back(Node{a, b, (uint8_t)cost, (uint8_t)g_slot, 0, 0}); return (uint32_t)g_nodes.size() - 1; }
struct Real {
    double v; uint32_t id;
    Real() = default;
    constexpr Real(double x) : v(x), id(0) {}
    constexpr Real(int x) : v((double)x), id(0) {}
    Real(double x, uint32_t i) : v(x), id(i) {}
    Real(const Real &o) = default;
    Real &operator=(const Real &o) {
        if (in_mem(this)) {       // a store to LDS / workspace / output
            g_stores++;
            if ((const void *)this == g_dummy) { g_dummy_stores++; v = o.v; id = 0; return *this; }
            if (!g_written.insert((const void *)this).second) { g_dup_stores++; v = o.v; id = 0; return *this; }      // written before in this phase: a clamped duplicate
            mark(o.id); v = o.v; id = 0;
        } else { v = o.v; id = o.id; }
        return *this;
    }
    explicit operator double() const { mark(id); return v; }
    explicit operator int() const { mark(id); return (int)v; }
};
static inline Real operator+(const Real &a, const Real &b) { return Real(a.v + b.v, op(a.id, b.id, 1)); }
static inline Real operator-(const Real &a, const Real &b) { return Real(a.v - b.v, op(a.id, b.id, 1)); }
static inline Real operator*(const Real &a, const Real &b) { return Real(a.v * b.v, op(a.id, b.id, 1)); }
static inline Real operator/(const Real &a, const Real &b) { return Real(a.v / b.v, op(a.id, b.id, 1)); }
static inline Real operator-(const Real &a) { return Real(-a.v, a.id); }
static inline Real &operator+=(Real &a, const Real &b) { a = a + b; return a; }
static inline Real &operator-=(Real &a, const Real &b) { a = a - b; return a; }
static inline Real &operator*=(Real &a, const Real &b) { a = a * b; return a; }
static inline bool dec(const Real &a, const Real &b) { mark(a.id); mark(b.id); return true; }
static inline bool operator<(const Real &a, const Real &b) { dec(a, b); return a.v < b.v; }
static inline bool operator>(const Real &a, const Real &b) { dec(a, b); return a.v > b.v; }
static inline bool operator<=(const Real &a, const Real &b) { dec(a, b); return a.v <= b.v; }
static inline bool operator>=(const Real &a, const Real &b) { dec(a, b); return a.v >= b.v; }
static inline bool operator==(const Real &a, const Real &b) { dec(a, b); return a.v == b.v; }
static inline bool operator!=(const Real &a, const Real &b) { dec(a, b); return a.v != b.v; }
static inline Real un(const Real &a, double v, int cost) { return Real(v, cost ? op(a.id, 0, cost) : a.id); }
static inline Real bi(const Real &a, const Real &b, double v, int cost) { return Real(v, op(a.id, b.id, cost)); }
}
using fc::Real;
using namespace fc;
#define BMPC_EMU 1
#define BMPC_HD
#define BMPC_D
#define BMPC_SINCOS(x, s, c) (*(s) = fc::un(Real(x), std::sin(Real(x).v), 1), *(c) = fc::un(Real(x), std::cos(Real(x).v), 1))
#define BMPC_EXP(x) fc::un(Real(x), std::exp(Real(x).v), 1)
#define BMPC_LOG(x) fc::un(Real(x), std::log(Real(x).v), 1)
#define BMPC_SQRT(x) fc::un(Real(x), std::sqrt(Real(x).v), 1)
#define BMPC_SIN(x) fc::un(Real(x), std::sin(Real(x).v), 1)
#define BMPC_COS(x) fc::un(Real(x), std::cos(Real(x).v), 1)
#define BMPC_ATAN2(y, x) fc::bi(Real(y), Real(x), std::atan2(Real(y).v, Real(x).v), 1)
#define BMPC_RSQRT(x) fc::un(Real(x), 1.0 / std::sqrt(Real(x).v), 1)
#define BMPC_FABS(x) fc::un(Real(x), std::fabs(Real(x).v), 0)
#define BMPC_FMAX(a, b) fc::bi(Real(a), Real(b), std::fmax(Real(a).v, Real(b).v), 0)
#define BMPC_FMIN(a, b) fc::bi(Real(a), Real(b), std::fmin(Real(a).v, Real(b).v), 0)
#define BMPC_POW15(x) ((x) * BMPC_SQRT(x))
#define BMPC_RINT(x) fc::un(Real(x), __builtin_rint(Real(x).v), 0)
#define BMPC_POW(x, y) fc::bi(Real(x), Real(y), std::pow(Real(x).v, Real(y).v), 1)
#define LANES_BEGIN for (int li_ = 0; li_ < 64; ++li_) { const int lane = W.order[li_]; (void)lane;
#define LANES_END } fc::g_written.clear();
#define LIDX lane
// the stamp `id` closes the phase whose operations were executed since the previous stamp: they were created with g_slot = a placeholder and are
// re-stamped here (the nodes since the last stamp form a contiguous tail of the arena)
static size_t g_tail = 1;
#define BMPC_PROF(W, id) { for (size_t n_ = g_tail; n_ < fc::g_nodes.size(); n_++) fc::g_nodes[n_].slot = (uint8_t)(id); g_tail = fc::g_nodes.size(); }
// the caller's option record (plain doubles; layout of bmpc::Opts in the other builds), declared before `double` changes its meaning
struct PlainOpts { double tol; int max_iter; double mu_init; double mu_min_fac; double slack_push; int exact_hessian; int verbose; double mu_warm; int stall_window;
                   double bound_margin; int restoration; int resto_short; int resto_cap; int start_rollout; int hold_mu; int retry_cap; };
#define double Real
#include "../../boundmpc_amd/csrc/bmpc_wave.inl"
#undef double

// out[0] = iterations, out[1] = converged, out[2] = executed flops, out[3] = useful flops, out[4] = stores, out[5] = duplicate stores, out[6] = dummy stores,
// out[8 + 2 s], out[9 + 2 s] = executed / useful flops of phase slot s (0..31)
extern "C" int bmpc_emu_count_useful(int N, int S, double h, const PlainOpts *po_, const double *p, const double *x0, unsigned long long *out) {
    bmpc::Opts oo; oo.tol = Real(po_->tol); oo.max_iter = po_->max_iter; oo.mu_init = Real(po_->mu_init); oo.mu_min_fac = Real(po_->mu_min_fac); oo.slack_push = Real(po_->slack_push);
    oo.exact_hessian = po_->exact_hessian; oo.verbose = 0; oo.mu_warm = Real(po_->mu_warm); oo.stall_window = po_->stall_window; oo.bound_margin = Real(po_->bound_margin);
    oo.restoration = po_->restoration; oo.resto_short = po_->resto_short; oo.resto_cap = po_->resto_cap; oo.start_rollout = po_->start_rollout; oo.hold_mu = po_->hold_mu; oo.retry_cap = po_->retry_cap;
    const bmpc::Opts *opts = &oo;
    if (S > bmpc::SMAX || S < 2 || N < 1 || N > bmpc::NMAX) return 1;
    const bmpc::Scr sc = bmpc::make_scr(N);
    const int np = 141 + 91 * S, nw = N * bmpc::NZ, ng = N * bmpc::NG;
    std::vector<Real> lds(bmpc::L_SIZE, Real(0.0)), scr(sc.size, Real(0.0)), x(nw), pp(np), xx(nw), g(ng), lg(ng);
    for (int i = 0; i < np; i++) pp[i] = Real(p[i]);
    for (int i = 0; i < nw; i++) xx[i] = Real(x0[i]);
    fc::g_nodes.assign(1, fc::Node{0, 0, 0, 31, 1, 0}); g_tail = 1; fc::g_written.clear(); fc::g_dup_stores = fc::g_dummy_stores = fc::g_stores = 0;
    fc::g_nmem = 0;
    auto reg = [](std::vector<Real> &v) { fc::g_mem_lo[fc::g_nmem] = (const char *)v.data(); fc::g_mem_hi[fc::g_nmem] = (const char *)(v.data() + v.size()); fc::g_nmem++; };
    reg(lds); reg(scr); reg(x); reg(g);
    fc::g_dummy = (const void *)(lds.data() + bmpc::L_DUMMY);
    bmpc::Wave W; W.N = N; W.S = S; W.h = Real(h); W.o = *opts; W.L = lds.data(); W.G = bmpc::make_gptr(scr.data()); W.it_base = 0;
    for (int i = 0; i < 64; i++) W.order[i] = i;
    bmpc::Problem pr; int it = 0, st = 0;
    pr.p = pp.data(); pr.x0 = xx.data();
    pr.x = x.data(); pr.g = g.data(); pr.lam_g = nullptr; pr.lam_x = nullptr; pr.f = nullptr; pr.kkt = nullptr; pr.iters = &it; pr.status = &st; pr.state = nullptr; pr.resto_from = -1;
    if (N <= 11 && S <= bmpc::SMAX_ZLDS) bmpc::wave_solve_retry<true>(W, pr); else bmpc::wave_solve_retry<false>(W, pr);
    for (int i = 0; i < 72; i++) out[i] = 0;
    out[0] = (unsigned long long)it; out[1] = st == 0; out[4] = fc::g_stores; out[5] = fc::g_dup_stores; out[6] = fc::g_dummy_stores;
    for (size_t n = 1; n < fc::g_nodes.size(); n++) {
        const fc::Node &nd = fc::g_nodes[n];
        out[2] += nd.cost; out[8 + 2 * (nd.slot & 31)] += nd.cost;
        if (nd.useful) { out[3] += nd.cost; out[9 + 2 * (nd.slot & 31)] += nd.cost; }
    }
    return 0;
}
