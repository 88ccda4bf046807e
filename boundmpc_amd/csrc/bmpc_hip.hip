// bmpc_hip.hip -- gfx950 kernel + C ABI (include/boundmpc_hip.h) of the batched BoundMPC OCP solver.
// One problem per 64-lane wavefront (one wave per workgroup); persistent workgroups pull problems
// from an atomic work queue so per-problem iteration counts load-balance; per-wave LDS working
// set + per-wave global scratch slab (see bmpc_wave.inl for the algorithm and the lane maps).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/boundmpc_hip.h"

#include "bmpc_gpu_common.h"
#define LANES_BEGIN { int lane_ = threadIdx.x; asm volatile("" : "+v"(lane_)); const int lane = lane_; (void)lane;   // opaque per phase: stops LICM from hoisting per-lane address arithmetic out of the solver loops (register pressure)
// The workgroup is ONE wave: its LDS and vector-memory instructions execute in program order, so a phase boundary needs no
// s_barrier and no s_waitcnt drain (what __syncthreads() would emit: vmcnt(0) lgkmcnt(0), i.e. a full stall on every
// outstanding prefetch / store).  A wavefront-scope fence keeps the COMPILER from moving memory operations across it.
#define LANES_END } __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();

#ifdef BMPC_MARKS
// diagnostic compile only (-S): textual markers in the ISA at the phase stamps, to count static instructions per phase
#define BMPC_PROF(W, id) asm volatile("s_nop 0 ; BMPCMARK " #id ::: "memory");
#endif
#ifdef BMPC_PROFILE
// diagnostic build only (libboundmpc_hip_prof.so): per-phase cycle stamps of lane 0, never in the product library
#undef BMPC_PROF
#define BMPC_PROF(W, id) { long long now_ = clock64(); if (threadIdx.x == 0) { ((long long *)((W).L + bmpc::L_PROF))[id] += now_ - (W).tprev; } (W).tprev = now_; }
#endif

#include "bmpc_wave.inl"
#define BMPCS_SYNC() __syncthreads()
#include "bmpc_stream.inl"

typedef KArgsT<bmpc::Opts> KArgs;
#ifndef BMPC_TEAM_NW
#define BMPC_TEAM_NW 4      // waves per problem of the team kernels compiled into the library (bmpc_team.hip)
#endif

#ifndef BMPC_WAVES_PER_EU
#define BMPC_WAVES_PER_EU 1
#endif
template <bool ZLDS>
__global__ void __launch_bounds__(64, BMPC_WAVES_PER_EU) bmpc_solve_kernel(KArgs a) {
    __shared__ double lds[bmpc::L_SIZE];
    bmpc::Wave W; W.N = a.N; W.S = a.S; W.h = a.h; W.o = a.o; W.L = lds; W.G = bmpc::make_gptr(a.scratch + (long long)blockIdx.x * a.scr_stride); W.wv = 0;
    W.deadline = 0; W.it_base = 0;
    const int np = 141 + 91 * a.S, nw = a.N * bmpc::NZ, ng = a.N * bmpc::NG;
#ifdef BMPC_PROFILE
    if (threadIdx.x < 32) ((long long *)(lds + bmpc::L_PROF))[threadIdx.x] = 0;
    __syncthreads();
    W.tprev = clock64();
#endif
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = atomicAdd(a.counter, 1);
        b = __builtin_amdgcn_readfirstlane(b);
        if ((unsigned)b >= (unsigned)a.B) break;      // every wave reaches this exit: the queue is finite (unsigned: a queue word nobody reset ends the wave, it never becomes an address)
        if (a.order) b = __builtin_amdgcn_readfirstlane(a.order[b]);      // longest-expected-first order of a batch larger than the resident waves (queue_order_kernel)
        bmpc::Problem pr;
        pr.p = a.p + (long long)b * np; pr.x0 = a.x0 + (long long)b * nw;
        pr.x = a.x ? a.x + (long long)b * nw : nullptr; pr.g = a.g ? a.g + (long long)b * ng : nullptr;
        pr.lam_g = a.lam_g ? a.lam_g + (long long)b * ng : nullptr; pr.lam_x = a.lam_x ? a.lam_x + (long long)b * nw : nullptr;
        pr.f = a.f ? a.f + b : nullptr; pr.kkt = a.kkt ? a.kkt + b : nullptr;
        pr.iters = a.iters ? a.iters + b : nullptr; pr.status = a.status ? a.status + b : nullptr;
        pr.state = a.state ? a.state + (long long)b * (a.N * bmpc::NI + 2) : nullptr;
        pr.resto_from = -1;
        const long long t0_ = a.latency_us ? (long long)wall_clock64() : 0;
        bmpc::wave_solve_retry<ZLDS>(W, pr);      // (+ the second attempt of a long-horizon solve that ends with status 2)
        __syncthreads();
        if (a.rcount && threadIdx.x == 0 && *pr.status == 4) atomicAdd(a.rcount, 1);      // jammed: the restoration kernel continues it (bmpc_resto.hip)
        if (a.latency_us && threadIdx.x == 0) a.latency_us[b] = (double)((long long)wall_clock64() - t0_) * 0.01;   // constant 100 MHz counter
    }
#ifdef BMPC_PROFILE
    if (threadIdx.x < 32 && a.prof) atomicAdd(a.prof + threadIdx.x, (unsigned long long)((long long *)(lds + bmpc::L_PROF))[threadIdx.x]);
#endif
}

// Work-queue order of a batch that takes several rounds of the resident waves (bmpc_set_queue_order): rank of every problem by DECREASING key
// (ties: by index) -> order[rank] = problem.  B threads, each walks the B keys (B <= a few 10^4: microseconds).  A launch lasts until its last
// wave is done, so problems that are expected to take long should start first; the key is the objective at the start point (evaluation pass).
#define BMPC_QUEUE_ORDER_MAX 65536      // the ranking is O(B^2 / lanes): beyond this many problems a batch keeps its natural order
__global__ void __launch_bounds__(256) queue_order_kernel(int B, const double *key, int *order) {
    __shared__ double tile[256];
    const int i = blockIdx.x * 256 + threadIdx.x, ic = i < B ? i : B - 1;
    const double ki = key[ic] == key[ic] ? key[ic] : INFINITY;      // (a NaN key -- f of a garbage x0 -- counts as the largest: the ranks must stay a permutation)
    int r = 0;
    for (int j0 = 0; j0 < B; j0 += 256) {      // keys staged through LDS a tile at a time: every thread of the block compares against the same 256 keys
        const int jj = j0 + threadIdx.x;
        __syncthreads();
        tile[threadIdx.x] = jj < B ? (key[jj] == key[jj] ? key[jj] : INFINITY) : -INFINITY;
        __syncthreads();
        const int n = B - j0 < 256 ? B - j0 : 256;
        for (int t = 0; t < n; t++) { const double kj = tile[t]; r += (kj > ki || (kj == ki && j0 + t < i)) ? 1 : 0; }
    }
    if (i < B) order[r] = i;
}

struct bmpc_handle {
    int N, S; double h; bmpc_options o;
    int dev;                 // device the handle was created on: workspace, work queue, events and streams of the handle live there
    int refs; bool closed;   // one reference for the creator, one per captured graph (their kernels carry the workspace addresses);
                             // bmpc_destroy closes the handle, the memory goes when the last reference does
    // launches of one handle share its workspace and work queue, so they are ordered against each other whatever streams the
    // caller uses: every launch records order_ev, a launch on another stream waits for it first
    hipEvent_t order_ev, bridge_ev; bool order_valid; hipStream_t order_stream;
    double rt_row_cap;       // real-time mode: a position tube row (l^2 - w^2, any stage) above this vetoes the iterate (bmpc_stream_set_rt_position_row_cap); 0 = off
    double rt_viol_tol;      // acceptance threshold of stream_post in real-time mode (flag bit 1); default = the reference's 1e-4
    double rt_budget_us;     // time budget of a fused tick (bmpc_stream_set_time_budget); 0 = none
    hipStream_t own_stream;  // graph replays requested on the legacy null stream run here, bracketed by events (bmpc_graph_launch)
    int grid; long long scr_stride; double *scratch; int scr_waves; int graphs_alive; int *counter; unsigned long long *prof;
    int team_grid;           // resident TEAMS (workgroups of BMPC_TEAM_NW waves, bmpc_team.hip) of the device; 0: no team kernel for this handle (N > 10 or S > 4)
    int pair_grid;           // resident PAIRS (workgroups of 2 waves at two waves per SIMD, bmpc_pair.hip); 0: no pair kernel for this handle (N > 11 or S > 4)
    int *aux_int; int aux_cap;      // [2][aux_cap] status / iters of a batch whose caller passed NULL (the restoration kernel reads them)
    int hold_mu;            // bmpc_set_barrier_hold: 1 = a solve holds the barrier level it starts on
    int retry_cap;          // bmpc_set_second_attempt: iterations of the second attempt of a stateless solve that ends with status 2 (0 = none; default 100 for N > 11)
    double level_c, level_lo, level_hi;      // bmpc_stream_set_level_rule: stream_pack sets the level of a stream's next tick (level_hi <= 0: off)
    int start_rollout;      // 1 (default): a stateless solve whose x0 is far off its own dynamics starts from the rollout of x0's jerks (bmpc_set_start_rollout)
    int resto_on, resto_short, resto_cap;      // restoration phase (bmpc_set_restoration): mode 0 off / 1 full (default N <= 11) / 2 after a numerical breakdown only (default N > 11); jam = resto_short consecutive short steps; iterations per phase
    int queue_order;         // bmpc_set_queue_order: 1 = a batch beyond the resident waves is solved in the order of decreasing f(x0) (default for N > 11), 0 = natural order
    double *qkey; int *qorder; int q_cap;      // [q_cap] keys and order of the last such batch
    int team_mode;           // bmpc_set_team_waves: 0 automatic (teams when the batch fits into the resident teams), 1 never, BMPC_TEAM_NW whenever possible
    int timing; hipEvent_t *ev; int nev; long long n_timed;   // timing = number of launches whose {start, stop} event pairs are kept (ring)
    double *latency_us;
    double *stage_d, *stage_h; int stage_cap;   // device and pinned host staging of the host-buffer path (bmpc_solve_batch_host)
};

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "boundmpc_hip: %s failed: %s\n", #x, hipGetErrorString(e_)); return BMPC_ERR_HIP; } } while (0)

// makes the handle's device current for the scope (allocation, free and synchronisation act on the CURRENT device)
struct DevGuard {
    int prev; bool changed;
    explicit DevGuard(int dev) : prev(dev), changed(false) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess; }
    ~DevGuard() { if (changed) hipSetDevice(prev); }
};
// A stream the CALLER is capturing (e.g. a torch CUDA graph around solve_batch / stream_tick) takes neither: waiting there on an event
// recorded outside the capture invalidates the capture, and recording order_ev inside it would turn it into a captured event that later
// direct launches then wait on.  Ordering of such launches against the handle's other work is the caller's (include/boundmpc_hip.h).
static bool caller_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
static int order_before(bmpc_handle *h, hipStream_t st) {
    if (caller_is_capturing(st)) return BMPC_OK;
    if (h->order_valid && st != h->order_stream) HIPCHK(hipStreamWaitEvent(st, h->order_ev, 0));
    return BMPC_OK;
}
static int order_after(bmpc_handle *h, hipStream_t st) {
    if (caller_is_capturing(st)) return BMPC_OK;
    HIPCHK(hipEventRecord(h->order_ev, st));
    h->order_stream = st; h->order_valid = true;
    return BMPC_OK;
}
// host wait for the last launch of THIS handle (every launch records order_ev): what teardown and workspace growth need -- not a
// device-wide drain, which would stall every other stream and handle of the process
static void wait_for_handle(bmpc_handle *h) {
    if (h->order_valid) hipEventSynchronize(h->order_ev);
}
static void handle_release(bmpc_handle *h) {
    if (--h->refs > 0) return;
    DevGuard dg(h->dev);
    for (int i = 0; i < 2 * h->nev; i++) hipEventDestroy(h->ev[i]);
    delete[] h->ev;
    if (h->order_ev) hipEventDestroy(h->order_ev);
    if (h->bridge_ev) hipEventDestroy(h->bridge_ev);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    hipFree(h->scratch); hipFree(h->counter); hipFree(h->aux_int); hipFree(h->qkey); hipFree(h->qorder); hipFree(h->prof); hipFree(h->stage_d); if (h->stage_h) hipHostFree(h->stage_h);
    delete h;
}

extern "C" int bmpc_default_options(bmpc_options *o) {
    if (!o) return BMPC_ERR_ARG;
    o->tol = 1e-8; o->max_iter = 500; o->mu_init = 0.1; o->mu_min_fac = 0.1; o->slack_push = 1e-2; o->exact_hessian = 1; o->verbose = 0; o->mu_warm = 1e-2; o->stall_window = 40; o->bound_margin = 0.0;
    return BMPC_OK;
}
extern "C" int bmpc_default_options_for(int N, bmpc_options *o) {
    const int rc = bmpc_default_options(o);
    if (rc == BMPC_OK && N > 11) { o->mu_init = 3.0; o->slack_push = 0.1; o->stall_window = 20; }   // long horizons: a cold start far from the solution wants a more central first
                                                                               // barrier level and roomier slacks (the values of the barrier restart; 3-15 % fewer iterations than 0.3 / 1e-2 at N = 16..40,
                                                                               // DESIGN.md 2); stalls are met by barrier restarts (bmpc_wave.inl), so they are looked for earlier
    return rc;
}
extern "C" const char *bmpc_error_string(int c) {
    switch (c) { case BMPC_OK: return "ok"; case BMPC_ERR_ARG: return "invalid argument"; case BMPC_ERR_HIP: return "HIP runtime error";
                 case BMPC_ERR_NOGPU: return "no HIP device available"; default: return "unknown error"; }
}
extern "C" int bmpc_create(int N, int S, double dt, const bmpc_options *opts, bmpc_handle **out) {
    if (!out || N < 1 || N > bmpc::NMAX || S < 2 || S > bmpc::SMAX || !(dt > 0)) return BMPC_ERR_ARG;
    if (opts && (!(opts->tol > 0) || opts->max_iter < 0 || opts->stall_window < 0 || (opts->stall_window & 1) || !(opts->mu_init > 0) || !(opts->slack_push > 0) || !(opts->bound_margin >= 0) || opts->bound_margin > 0.5)) return BMPC_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return BMPC_ERR_NOGPU;
    bmpc_handle *h = new (std::nothrow) bmpc_handle();
    if (!h) return BMPC_ERR_ARG;
    h->start_rollout = 1; h->hold_mu = 0; h->retry_cap = N > 11 ? 100 : 0; h->level_c = 0.0; h->level_lo = 0.0; h->level_hi = 0.0;
    h->resto_on = N <= 11 ? 1 : 2; h->resto_short = 6; h->resto_cap = 40;      // restoration phase: full for short horizons, after a numerical breakdown only for long ones (bmpc_set_restoration)
    h->N = N; h->S = S; h->h = dt; h->timing = 0; h->ev = nullptr; h->nev = 0; h->n_timed = 0; h->latency_us = nullptr;
    h->scratch = nullptr; h->scr_waves = 0; h->graphs_alive = 0; h->counter = nullptr; h->aux_int = nullptr; h->aux_cap = 0; h->prof = nullptr; h->stage_d = nullptr; h->stage_h = nullptr; h->stage_cap = 0;
    h->team_grid = 0; h->pair_grid = 0; h->team_mode = 0;
    h->queue_order = N > 11 ? 1 : 0; h->qkey = nullptr; h->qorder = nullptr; h->q_cap = 0;
    h->rt_viol_tol = 1e-4; h->rt_row_cap = 0.0; h->rt_budget_us = 0.0; h->dev = 0; h->refs = 1; h->closed = false; h->order_ev = nullptr; h->bridge_ev = nullptr; h->order_valid = false; h->order_stream = nullptr; h->own_stream = nullptr;
    if (opts) h->o = *opts; else bmpc_default_options_for(N, &h->o);
    int dev = 0, per_cu = 0; hipDeviceProp_t prop;
    bool ok = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess;
    h->dev = dev;
    if (ok) ok = ((N <= 11 && S <= bmpc::SMAX_ZLDS) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bmpc_solve_kernel<true>, 64, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bmpc_solve_kernel<false>, 64, 0)) == hipSuccess;
    if (ok) {
        if (per_cu < 1) per_cu = 1;
        h->grid = per_cu * prop.multiProcessorCount;
        h->scr_stride = bmpc::make_scr(N).size;
        // teams: a workgroup of BMPC_TEAM_NW waves per problem (bmpc_team.hip; iterate-in-LDS instantiation only)
        h->team_grid = (N <= bmpc_team_nmax(BMPC_TEAM_NW) && S <= bmpc::SMAX_ZLDS) ? bmpc_team_blocks_per_cu(BMPC_TEAM_NW) * prop.multiProcessorCount : 0;
        // pairs: two waves per problem at two waves per SIMD (bmpc_pair.hip); same workspace layout and LDS budget as one wave per problem
        h->pair_grid = (N <= bmpc_pair_nmax() && S <= bmpc::SMAX_ZLDS && bmpc_pair_scr_stride(N) == h->scr_stride) ? bmpc_pair_blocks_per_cu() * prop.multiProcessorCount : 0;
        // the per-wave workspace slabs (148 KB at N=10, 444 KB at N=30) are allocated on the first solve, for min(B, grid) waves, and grow on demand:
        // a single-problem handle (the nlpsol shim of one BoundMPC object) holds one slab, not 1024
        ok = hipMalloc(&h->counter, 4 * sizeof(int)) == hipSuccess      /* work queue, work queue of the restoration kernel, jam count */
          && hipMemset(h->counter, 0, 4 * sizeof(int)) == hipSuccess
          && hipMalloc(&h->prof, 32 * sizeof(unsigned long long)) == hipSuccess
          && hipMemset(h->prof, 0, 32 * sizeof(unsigned long long)) == hipSuccess
          && hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming) == hipSuccess
          && hipEventCreateWithFlags(&h->bridge_ev, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {   // nothing half-built survives a failed create
        fprintf(stderr, "boundmpc_hip: bmpc_create failed: %s\n", hipGetErrorString(hipGetLastError()));
        if (h->order_ev) hipEventDestroy(h->order_ev);
        if (h->bridge_ev) hipEventDestroy(h->bridge_ev);
        hipFree(h->scratch); hipFree(h->counter); hipFree(h->prof); delete h;
        return BMPC_ERR_HIP;
    }
    *out = h;
    return BMPC_OK;
}
extern "C" int bmpc_destroy(bmpc_handle *h) {
    if (!h || h->closed) return BMPC_ERR_ARG;
    {
        DevGuard dg(h->dev);
        wait_for_handle(h);        // launches are asynchronous: nothing of this handle may still be running on its workspace
    }
    // captured graphs carry the addresses of the workspace and of the work queue: while one is alive the memory stays, the handle
    // only stops accepting work (bmpc_graph_launch of such a graph returns BMPC_ERR_ARG); the last bmpc_graph_destroy frees it
    h->closed = true;
    handle_release(h);
    return BMPC_OK;
}
extern "C" int bmpc_num_vars(const bmpc_handle *h) { return h ? h->N * bmpc::NZ : -1; }
extern "C" int bmpc_num_cons(const bmpc_handle *h) { return h ? h->N * bmpc::NG : -1; }
extern "C" int bmpc_num_params(const bmpc_handle *h) { return h ? 141 + 91 * h->S : -1; }

extern "C" int bmpc_get_bounds(const bmpc_handle *h, double *lbx, double *ubx, double *lbg, double *ubg) {
    if (!h) return BMPC_ERR_ARG;
    // casadi_ocp_formulation.py:92-153 (variables) and :272-349 (constraints); limits RobotModel.py:20-43
    const double qd[7] = {165, 115, 165, 115, 165, 115, 170}, dqd[7] = {85, 85, 100, 75, 130, 135, 135};
    const double inf = INFINITY, pi = 3.14159265358979323846;
    for (int k = 0; k < h->N; k++) {
        double *l = lbx ? lbx + k * 44 : nullptr, *u = ubx ? ubx + k * 44 : nullptr;
        for (int i = 0; i < 44; i++) {
            double lo = -inf, hi = inf;
            if (i < 8) { lo = -35.0; hi = 35.0; }
            else if (i < 15) { hi = qd[i - 8] * pi / 180; lo = -hi; }
            else if (i < 22) { hi = dqd[i - 15] * pi / 180; lo = -hi; }
            else if (i == 41) { lo = 0.0; }
            if (l) l[i] = lo; if (u) u[i] = hi;
        }
        for (int i = 0; i < 43; i++) { if (lbg) lbg[k * 43 + i] = i < 36 ? 0.0 : -inf; if (ubg) ubg[k * 43 + i] = 0.0; }
    }
    return BMPC_OK;
}

// workspace for `waves` resident waves; growing frees the old slabs, which no launch may still be using: the handle's last launch is
// waited for first, and captured graphs (which carry the old address) forbid growth -- size the first solve / capture for the largest batch
static int ensure_scratch(bmpc_handle *h, int waves) {
    if (waves <= h->scr_waves) return BMPC_OK;
    DevGuard dg(h->dev);
    if (h->graphs_alive > 0 && h->scratch) {
        fprintf(stderr, "boundmpc_hip: a larger batch needs a larger workspace, but %d captured graph(s) hold the current one\n", h->graphs_alive);
        return BMPC_ERR_ARG;
    }
    if (h->scratch) { wait_for_handle(h); HIPCHK(hipFree(h->scratch)); h->scratch = nullptr; h->scr_waves = 0; }
    HIPCHK(hipMalloc(&h->scratch, sizeof(double) * (size_t)h->scr_stride * waves));
    h->scr_waves = waves;
    return BMPC_OK;
}

// event pair of the next timed launch (ring of h->timing pairs, created on first use)
static int timing_slot(bmpc_handle *h, hipEvent_t **pair) {
    if (h->nev < h->timing) {
        hipEvent_t *ne = new hipEvent_t[2 * h->timing];
        for (int i = 0; i < 2 * h->nev; i++) ne[i] = h->ev[i];
        for (int i = 2 * h->nev; i < 2 * h->timing; i++) HIPCHK(hipEventCreate(&ne[i]));
        delete[] h->ev; h->ev = ne; h->nev = h->timing;
    }
    *pair = h->ev + 2 * (h->n_timed % h->timing);
    return BMPC_OK;
}
// Which kernel solves a batch of B (waves per problem): a team of BMPC_TEAM_NW waves when the batch fits into the resident teams of the device
// (256 on an MI355X: a team owns a CU) or when the caller asked for teams; a pair (2 waves on the one-wave budget: 512 resident, bmpc_pair.hip)
// when it fits into the resident pairs; else one wave per problem.  bmpc_set_team_waves: 1 = always one wave, 2 = pairs whatever the batch,
// BMPC_TEAM_NW = teams whatever the batch.
static int solve_waves(const bmpc_handle *h, int B) {
    if (h->team_mode == 1) return 1;
    if (h->team_mode == 2) return h->pair_grid > 0 ? 2 : 1;
    if (h->team_grid > 0 && (h->team_mode == BMPC_TEAM_NW || B <= h->team_grid)) return BMPC_TEAM_NW;
    return (h->pair_grid > 0 && B <= h->pair_grid) ? 2 : 1;
}
static bool use_team(const bmpc_handle *h, int B) { return solve_waves(h, B) == BMPC_TEAM_NW; }
static int launch_grid(const bmpc_handle *h, int B) { const int w = solve_waves(h, B), g = w == BMPC_TEAM_NW ? h->team_grid : (w == 2 ? h->pair_grid : h->grid); return B < g ? B : g; }
// the restoration kernel (bmpc_resto.hip) runs one wave per problem whatever kernel solved the batch
static int resto_grid(const bmpc_handle *h, int B) { return B < h->grid ? B : h->grid; }
// what a batch of B needs besides the caller's buffers: workspace slabs for the resident waves of both kernels, and -- the hand-over to the
// restoration kernel goes through status[] and iters[] -- handle-owned stand-ins for a caller that passes NULL there.  Never inside a capture.
static int reserve_for_batch(bmpc_handle *h, int B) {
    const int lg = launch_grid(h, B), rg = h->resto_on ? resto_grid(h, B) : 0;
    const int rc = ensure_scratch(h, lg > rg ? lg : rg);
    if (rc != BMPC_OK) return rc;
    if (B > h->aux_cap) {
        DevGuard dg(h->dev);
        if (h->graphs_alive > 0 && h->aux_int) { fprintf(stderr, "boundmpc_hip: a larger batch needs larger status buffers, but captured graphs hold the current ones\n"); return BMPC_ERR_ARG; }
        if (h->aux_int) { wait_for_handle(h); HIPCHK(hipFree(h->aux_int)); h->aux_int = nullptr; h->aux_cap = 0; }
        HIPCHK(hipMalloc(&h->aux_int, sizeof(int) * 2 * (size_t)B));
        h->aux_cap = B;
    }
    if (h->queue_order && B > h->q_cap && B <= BMPC_QUEUE_ORDER_MAX && solve_waves(h, B) == 1 && B > h->grid) {
        DevGuard dg(h->dev);
        if (h->graphs_alive > 0 && h->qkey) { fprintf(stderr, "boundmpc_hip: a larger batch needs larger queue-order buffers, but captured graphs hold the current ones\n"); return BMPC_ERR_ARG; }
        if (h->qkey) { wait_for_handle(h); hipFree(h->qkey); hipFree(h->qorder); h->qkey = nullptr; h->qorder = nullptr; h->q_cap = 0; }
        HIPCHK(hipMalloc(&h->qkey, sizeof(double) * (size_t)B)); HIPCHK(hipMalloc(&h->qorder, sizeof(int) * (size_t)B));
        h->q_cap = B;
    }
    return BMPC_OK;
}
extern "C" int bmpc_set_queue_order(bmpc_handle *h, int mode) {
    if (!h || mode < 0 || mode > 1) return BMPC_ERR_ARG;
    h->queue_order = mode;
    return BMPC_OK;
}
extern "C" int bmpc_get_queue_order(const bmpc_handle *h) { return h ? h->queue_order : -1; }
extern "C" int bmpc_set_restoration(bmpc_handle *h, int enabled, int short_steps, int cap) {
    if (!h || short_steps > 1000 || cap > 100000 || cap == 0 || enabled > 2) return BMPC_ERR_ARG;      // (every argument is checked before any is applied)
    if (enabled >= 0) h->resto_on = enabled;
    if (short_steps >= 0) h->resto_short = short_steps;
    if (cap >= 1) h->resto_cap = cap;
    return BMPC_OK;
}
extern "C" int bmpc_get_restoration(const bmpc_handle *h, int *enabled, int *short_steps, int *cap) {
    if (!h) return BMPC_ERR_ARG;
    if (enabled) *enabled = h->resto_on; if (short_steps) *short_steps = h->resto_short; if (cap) *cap = h->resto_cap;
    return BMPC_OK;
}
extern "C" int bmpc_set_start_rollout(bmpc_handle *h, int enabled) {
    if (!h || enabled < 0 || enabled > 1) return BMPC_ERR_ARG;
    h->start_rollout = enabled;
    return BMPC_OK;
}
extern "C" int bmpc_set_barrier_hold(bmpc_handle *h, int enabled) {
    if (!h || enabled < 0 || enabled > 1) return BMPC_ERR_ARG;
    h->hold_mu = enabled;
    return BMPC_OK;
}
extern "C" int bmpc_set_second_attempt(bmpc_handle *h, int cap) {
    if (!h || cap < 0 || cap > 100000) return BMPC_ERR_ARG;
    h->retry_cap = cap;
    return BMPC_OK;
}
extern "C" int bmpc_get_second_attempt(const bmpc_handle *h) { return h ? h->retry_cap : -1; }
extern "C" int bmpc_stream_set_level_rule(bmpc_handle *h, double c, double lo, double hi) {
    if (!h || !(c >= 0.0) || !(lo >= 0.0) || !(hi >= 0.0) || (hi > 0.0 && !(lo > 0.0 && lo <= hi))) return BMPC_ERR_ARG;
    h->level_c = c; h->level_lo = lo; h->level_hi = hi;
    return BMPC_OK;
}
extern "C" int bmpc_get_start_rollout(const bmpc_handle *h) { return h ? h->start_rollout : -1; }
extern "C" int bmpc_options_size(void) { return (int)sizeof(bmpc_options); }
#ifndef BMPC_BUILD_HASH_STR
#define BMPC_BUILD_HASH_STR "0000000000000000"
#endif
// hash of the source text and compiler flags this library was built from (boundmpc_amd/build.py source_hash); the marker is also found in the file's bytes
static const char bmpc_build_hash_marker[] = "BMPC_BUILD_HASH=" BMPC_BUILD_HASH_STR;
extern "C" const char *bmpc_build_hash(void) { return bmpc_build_hash_marker + 16; }
extern "C" int bmpc_set_team_waves(bmpc_handle *h, int waves) {
    if (!h || (waves != 0 && waves != 1 && waves != 2 && waves != BMPC_TEAM_NW)) return BMPC_ERR_ARG;
    if (waves == BMPC_TEAM_NW && h->team_grid <= 0) return BMPC_ERR_ARG;      // no team instantiation for this horizon / window
    if (waves == 2 && h->pair_grid <= 0) return BMPC_ERR_ARG;                // no pair instantiation
    h->team_mode = waves;
    return BMPC_OK;
}
extern "C" int bmpc_team_info(const bmpc_handle *h, int B, int *waves, int *resident_teams, int *lds_bytes) {
    if (!h) return BMPC_ERR_ARG;
    const int w = solve_waves(h, B);
    if (waves) *waves = w;
    if (resident_teams) *resident_teams = w == 2 ? h->pair_grid : h->team_grid;
    if (lds_bytes) *lds_bytes = w == 2 ? bmpc_pair_lds_bytes() : bmpc_team_lds_bytes(BMPC_TEAM_NW);
    return BMPC_OK;
}
// Reset of the work-queue words (queue, restoration queue, jam count) ahead of a batch kernel.  A KERNEL, not hipMemsetAsync: inside a captured graph
// the runtime (ROCm 7.2) does not reliably order a memset node before the kernel node that follows it when the replay comes behind a cross-stream
// event wait -- a replayed pair kernel drew its first problem index from a queue word that had not been reset yet (round 6: wrong results, then a
// memory fault on a negative index; profiles/r06_e_graph_memset_node.txt).  Kernel after kernel is ordered by the queue itself.
__global__ void __launch_bounds__(64) queue_reset_kernel(int *c) { if (threadIdx.x < 3) c[threadIdx.x] = 0; }
static hipError_t reset_queue(bmpc_handle *h, hipStream_t st) {
    hipLaunchKernelGGL(queue_reset_kernel, dim3(1), dim3(64), 0, st, h->counter);
    return hipGetLastError();
}
// fills the kernel arguments and enqueues {reset of the work-queue counter, solver kernel} on `st`
static int enqueue_solve(bmpc_handle *h, int B, const double *p, const double *x0, double *state, int max_iter, double *x, double *g, double *lam_g,
                         double *lam_x, double *f, int *iters, int *status, double *kkt, hipStream_t st, bool timed, bool capturing = false) {
    if (h->closed) return BMPC_ERR_ARG;
    if (!capturing) { const int rc_ = order_before(h, st); if (rc_ != BMPC_OK) return rc_; }
    KArgs a; a.N = h->N; a.S = h->S; a.B = B; a.h = h->h;
    a.o.tol = h->o.tol; a.o.max_iter = max_iter > 0 ? max_iter : h->o.max_iter; a.o.mu_init = h->o.mu_init; a.o.mu_min_fac = h->o.mu_min_fac;
    a.o.slack_push = h->o.slack_push; a.o.exact_hessian = h->o.exact_hessian; a.o.verbose = 0; a.o.mu_warm = h->o.mu_warm; a.o.stall_window = h->o.stall_window; a.o.bound_margin = h->o.bound_margin;
    a.o.restoration = h->resto_on; a.o.resto_short = h->resto_short; a.o.resto_cap = h->resto_cap; a.o.start_rollout = h->start_rollout; a.o.hold_mu = h->hold_mu; a.o.retry_cap = state ? 0 : h->retry_cap;
    a.p = p; a.x0 = x0; a.x = x; a.g = g; a.lam_g = lam_g; a.lam_x = lam_x; a.f = f; a.kkt = kkt; a.iters = iters; a.status = status;
    a.state = state; a.latency_us = h->latency_us; a.budget_ticks = 0;
    const int grid = launch_grid(h, B);
    if (grid > h->scr_waves) return BMPC_ERR_ARG;      // callers reserve the workspace first (never inside a stream capture)
    a.scratch = h->scratch; a.scr_stride = h->scr_stride; a.counter = h->counter; a.prof = h->prof;
    // restoration phase: the batch kernels hand a jammed problem over through status[] / iters[] (handle-owned when the caller wants neither)
    const bool resto = h->resto_on != 0;
    a.counter2 = h->counter + 1; a.rcount = resto ? h->counter + 2 : nullptr;
    if (resto && (!a.status || !a.iters)) {
        if (B > h->aux_cap) return BMPC_ERR_ARG;
        if (!a.status) a.status = h->aux_int; if (!a.iters) a.iters = h->aux_int + h->aux_cap;
    }
    const int rgrid = resto ? resto_grid(h, B) : 0;
    if (rgrid > h->scr_waves) return BMPC_ERR_ARG;
    const bool zlds = h->N <= 11 && h->S <= bmpc::SMAX_ZLDS;
    hipEvent_t *pair = nullptr;
    if (timed) { int rc = timing_slot(h, &pair); if (rc != BMPC_OK) return rc; HIPCHK(hipEventRecord(pair[0], st)); }
    a.order = nullptr;
    if (h->queue_order && !state && solve_waves(h, B) == 1 && B > h->grid && B <= h->q_cap && B <= BMPC_QUEUE_ORDER_MAX) {
        // Longest-expected-first: an evaluation pass (the same kernel with max_iter = 0: f at x0, nothing else written), the ranking, then the solve
        // hands the problems out in that order.  Inside the timed region: it is part of what the batch costs.  A result does not depend on which
        // wave solves it or when (bitwise invariance under permutation of the batch is a test), so the outputs are those of the natural order.
        KArgs e = a; e.o.max_iter = 0; e.o.start_rollout = 0; e.x = nullptr; e.g = nullptr; e.lam_g = nullptr; e.lam_x = nullptr; e.kkt = nullptr; e.iters = nullptr; e.status = nullptr;
        e.f = h->qkey; e.state = nullptr; e.latency_us = nullptr; e.rcount = nullptr; e.order = nullptr;
        HIPCHK(reset_queue(h, st));
        if (zlds) hipLaunchKernelGGL(bmpc_solve_kernel<true>, dim3(grid), dim3(64), 0, st, e);
        else hipLaunchKernelGGL(bmpc_solve_kernel<false>, dim3(grid), dim3(64), 0, st, e);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(queue_order_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, (const double *)h->qkey, h->qorder);
        HIPCHK(hipGetLastError());
        a.order = h->qorder;
    }
    HIPCHK(reset_queue(h, st));
    // long horizons with the FULL restoration phase (mode 1 is not their default): the whole batch runs in the instantiation that holds the phase, so the
    // continuations of the few problems that need it sit in the work queue instead of following the batch on a handful of waves (same results)
    const bool whole_in_resto = h->resto_on == 1 && !zlds && solve_waves(h, B) == 1;
    if (whole_in_resto) { KArgs f = a; f.rcount = nullptr; HIPCHK(bmpc_resto_launch(zlds, &f, grid, st)); }
    else if (use_team(h, B)) HIPCHK(bmpc_team_launch_solve(BMPC_TEAM_NW, &a, grid, st));      // a workgroup of waves per problem (bmpc_team.hip)
    else if (solve_waves(h, B) == 2) HIPCHK(bmpc_pair_launch_solve(&a, grid, st));           // two waves per problem at two waves per SIMD (bmpc_pair.hip)
    else if (zlds) hipLaunchKernelGGL(bmpc_solve_kernel<true>, dim3(grid), dim3(64), 0, st, a);      // iterate in LDS; else in the workspace (long horizons, S > 4)
    else hipLaunchKernelGGL(bmpc_solve_kernel<false>, dim3(grid), dim3(64), 0, st, a);
    HIPCHK(hipGetLastError());
    if (resto && !whole_in_resto) HIPCHK(bmpc_resto_launch(zlds, &a, rgrid, st));      // continues what the batch kernel left jammed; returns at once when nothing did (bmpc_resto.hip)
    if (timed) { HIPCHK(hipEventRecord(pair[1], st)); h->n_timed++; }
    if (!capturing) return order_after(h, st);
    return BMPC_OK;
}

extern "C" int bmpc_solve_batch(bmpc_handle *h, int B, const double *p, const double *x0, double *x, double *g, double *lam_g, double *lam_x,
                                double *f, int *iters, int *status, double *kkt, void *hip_stream) {
    if (!h || B < 0 || (B > 0 && (!p || !x0 || !x))) return BMPC_ERR_ARG;
    if (B == 0) return BMPC_OK;
    { const int rc_ = reserve_for_batch(h, B); if (rc_ != BMPC_OK) return rc_; }
    return enqueue_solve(h, B, p, x0, nullptr, 0, x, g, lam_g, lam_x, f, iters, status, kkt, (hipStream_t)hip_stream, h->timing != 0);
}

extern "C" int bmpc_state_len(const bmpc_handle *h) { return h ? h->N * bmpc::NI + 2 : -1; }

extern "C" int bmpc_solve_batch_warm(bmpc_handle *h, int B, const double *p, const double *x0, double *state, int max_iter, double *x, double *g,
                                     double *lam_g, double *lam_x, double *f, int *iters, int *status, double *kkt, void *hip_stream) {
    if (!h || B < 0 || max_iter < 0 || (B > 0 && (!p || !x0 || !x || !state))) return BMPC_ERR_ARG;
    if (B == 0) return BMPC_OK;
    { const int rc_ = reserve_for_batch(h, B); if (rc_ != BMPC_OK) return rc_; }
    return enqueue_solve(h, B, p, x0, state, max_iter, x, g, lam_g, lam_x, f, iters, status, kkt, (hipStream_t)hip_stream, h->timing != 0);
}

// ---- hipGraph-captured step: {queue reset, solver kernel, restoration kernel} of one (warm-started) solve, instantiated once, replayed per tick ----
struct bmpc_graph { bmpc_handle *h; hipGraph_t graph; hipGraphExec_t exec; };

extern "C" int bmpc_graph_create(bmpc_handle *h, int B, const double *p, const double *x0, double *state, int max_iter, double *x, double *g,
                                 double *lam_g, double *lam_x, double *f, int *iters, int *status, double *kkt, bmpc_graph **out) {
    if (!h || !out || B < 1 || max_iter < 0 || !p || !x0 || !x) return BMPC_ERR_ARG;
    { const int rc_ = reserve_for_batch(h, B); if (rc_ != BMPC_OK) return rc_; }
    hipStream_t cs;
    HIPCHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    bmpc_graph *gr = new (std::nothrow) bmpc_graph();
    if (!gr) { hipStreamDestroy(cs); return BMPC_ERR_ARG; }
    gr->h = h; gr->graph = nullptr; gr->exec = nullptr;
    int rc = BMPC_OK;
    if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) rc = BMPC_ERR_HIP;
    if (rc == BMPC_OK) {
        rc = enqueue_solve(h, B, p, x0, state, max_iter, x, g, lam_g, lam_x, f, iters, status, kkt, cs, false, true);
        hipError_t e = hipStreamEndCapture(cs, &gr->graph);       // always end the capture, also after an enqueue error
        if (rc == BMPC_OK && e != hipSuccess) rc = BMPC_ERR_HIP;
    }
    if (rc == BMPC_OK && hipGraphInstantiate(&gr->exec, gr->graph, nullptr, nullptr, 0) != hipSuccess) rc = BMPC_ERR_HIP;
    hipStreamDestroy(cs);
    if (rc != BMPC_OK) { if (gr->exec) hipGraphExecDestroy(gr->exec); if (gr->graph) hipGraphDestroy(gr->graph); delete gr; return rc; }
    *out = gr; h->graphs_alive++; h->refs++;
    return BMPC_OK;
}
extern "C" int bmpc_graph_launch(bmpc_graph *gr, void *hip_stream) {
    if (!gr || !gr->h || gr->h->closed) return BMPC_ERR_ARG;
    bmpc_handle *h = gr->h; hipStream_t st = (hipStream_t)hip_stream;
    // A replay requested on the LEGACY NULL STREAM does not run there: on ROCm 7.2 a graph replayed on the null stream, followed by
    // further null-stream launches without a host synchronisation, ended in a GPU memory fault (tests/cabi/graph_nullstream.cpp).
    // It runs on a non-blocking stream of the handle, bracketed by events: after everything the null stream holds so far, and the
    // null stream's later work after it -- the ordering a caller expects from "launch on the null stream".
    const bool bridged = st == nullptr;
    if (bridged) {
        DevGuard dg(h->dev);
        if (!h->own_stream) HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        HIPCHK(hipEventRecord(h->bridge_ev, nullptr));
        HIPCHK(hipStreamWaitEvent(h->own_stream, h->bridge_ev, 0));
        st = h->own_stream;
    }
    { const int rc_ = order_before(h, st); if (rc_ != BMPC_OK) return rc_; }
    hipEvent_t *pair = nullptr;
    if (h->timing) { int rc = timing_slot(h, &pair); if (rc != BMPC_OK) return rc; HIPCHK(hipEventRecord(pair[0], st)); }
    HIPCHK(hipGraphLaunch(gr->exec, st));
    if (h->timing) { HIPCHK(hipEventRecord(pair[1], st)); h->n_timed++; }
    { const int rc_ = order_after(h, st); if (rc_ != BMPC_OK) return rc_; }
    if (bridged) HIPCHK(hipStreamWaitEvent(nullptr, h->order_ev, 0));
    return BMPC_OK;
}
extern "C" int bmpc_graph_destroy(bmpc_graph *gr) {
    if (!gr) return BMPC_ERR_ARG;
    bmpc_handle *h = gr->h;
    if (h) {
        DevGuard dg(h->dev);
        wait_for_handle(h);            // a replay of this graph may still be in flight (bmpc_graph_launch records the handle's event behind it)
    }
    hipGraphExecDestroy(gr->exec); hipGraphDestroy(gr->graph); delete gr;
    if (h) { if (h->graphs_alive > 0) h->graphs_alive--; handle_release(h); }     // the last reference of a closed handle frees it
    return BMPC_OK;
}

// host-buffer path: device and pinned host staging buffers are owned by the handle and grow on demand (no hipMalloc per call: the
// single-problem solver(...) call of the drop-in shim runs every tick).  One staging record holds everything that crosses PCIe for a
// call, inputs first: [p | x0 | x | lam_x | g | lam_g | f | kkt] doubles, then [iters | status] ints -- so a call is ONE host-to-device
// copy of the inputs, the launch, ONE device-to-host copy of the outputs and one stream synchronisation (until round 3: two blocking
// copies in, a device synchronisation and eight blocking copies out, ~190 us around a 1.1 ms single-problem launch).
static int host_stage_reserve(bmpc_handle *h, int B) {
    if (B <= h->stage_cap) return BMPC_OK;
    const size_t np = 141 + 91 * h->S, nw = (size_t)h->N * 44, ng = (size_t)h->N * 43;
    const size_t bytes = ((size_t)B * (np + 3 * nw + 2 * ng + 2)) * sizeof(double) + (size_t)B * 2 * sizeof(int);
    if (h->stage_d) { hipFree(h->stage_d); h->stage_d = nullptr; }
    if (h->stage_h) { hipHostFree(h->stage_h); h->stage_h = nullptr; }
    h->stage_cap = 0;
    if (hipMalloc(&h->stage_d, bytes) != hipSuccess || hipHostMalloc(&h->stage_h, bytes, hipHostMallocDefault) != hipSuccess) {
        hipFree(h->stage_d); if (h->stage_h) hipHostFree(h->stage_h); h->stage_d = nullptr; h->stage_h = nullptr;
        return BMPC_ERR_HIP;
    }
    h->stage_cap = B;
    return BMPC_OK;
}
extern "C" int bmpc_solve_batch_host(bmpc_handle *h, int B, const double *p, const double *x0, double *x, double *g, double *lam_g, double *lam_x,
                                     double *f, int *iters, int *status, double *kkt) {
    if (!h || B < 0 || (B > 0 && (!p || !x0 || !x))) return BMPC_ERR_ARG;
    if (B == 0) return BMPC_OK;
    int rc = host_stage_reserve(h, B);
    if (rc != BMPC_OK) return rc;
    const size_t np = 141 + 91 * h->S, nw = (size_t)h->N * 44, ng = (size_t)h->N * 43, b = (size_t)B;
    const size_t n_in = b * (np + nw), n_out = b * (2 * nw + 2 * ng + 2);
    double *dp = h->stage_d, *dx0 = dp + b * np, *dx = dx0 + b * nw, *dlx = dx + b * nw, *dg = dlx + b * nw, *dlg = dg + b * ng, *df = dlg + b * ng, *dk = df + b;
    int *dit = (int *)(dk + b), *dst = dit + b;
    double *hp = h->stage_h, *hx0 = hp + b * np, *hx = hx0 + b * nw, *hlx = hx + b * nw, *hg = hlx + b * nw, *hlg = hg + b * ng, *hf = hlg + b * ng, *hk = hf + b;
    const int *hit = (const int *)(hk + b), *hst = hit + b;
#define TRY(x) do { if (rc == BMPC_OK && (x) != hipSuccess) rc = BMPC_ERR_HIP; } while (0)
    // on a non-blocking stream of the handle (round 5; it was the legacy null stream, on which the call serialised against every blocking stream
    // of a torch process): the staged copies and the launch touch only the handle's own buffers
    hipStream_t hs = nullptr;
    {
        DevGuard dg(h->dev);
        if (!h->own_stream && hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) return BMPC_ERR_HIP;
        hs = h->own_stream;
    }
    memcpy(hp, p, b * np * sizeof(double)); memcpy(hx0, x0, b * nw * sizeof(double));
    TRY(hipMemcpyAsync(dp, hp, n_in * sizeof(double), hipMemcpyHostToDevice, hs));
    if (rc == BMPC_OK) rc = bmpc_solve_batch(h, B, dp, dx0, dx, dg, dlg, dlx, df, dit, dst, dk, hs);
    TRY(hipMemcpyAsync(hx, dx, n_out * sizeof(double) + b * 2 * sizeof(int), hipMemcpyDeviceToHost, hs));
    TRY(hipStreamSynchronize(hs));
    if (rc == BMPC_OK) {
        memcpy(x, hx, b * nw * sizeof(double));
        if (g) memcpy(g, hg, b * ng * sizeof(double));
        if (lam_g) memcpy(lam_g, hlg, b * ng * sizeof(double));
        if (lam_x) memcpy(lam_x, hlx, b * nw * sizeof(double));
        if (f) memcpy(f, hf, b * sizeof(double));
        if (kkt) memcpy(kkt, hk, b * sizeof(double));
        if (iters) memcpy(iters, hit, b * sizeof(int));
        if (status) memcpy(status, hst, b * sizeof(int));
    }
#undef TRY
    return rc;
}

extern "C" int bmpc_set_latency_buffer(bmpc_handle *h, double *latency_us) { if (!h) return BMPC_ERR_ARG; h->latency_us = latency_us; return BMPC_OK; }
extern "C" int bmpc_set_timing(bmpc_handle *h, int keep) {
    if (!h || keep < 0 || keep > 65536) return BMPC_ERR_ARG;
    h->timing = keep; h->n_timed = 0;
    return BMPC_OK;
}
extern "C" int bmpc_kernel_ms(bmpc_handle *h, int back, float *ms) {
    if (!h || !ms || back < 0 || h->timing <= 0 || back >= h->timing || back >= h->n_timed) return BMPC_ERR_ARG;
    hipEvent_t *pair = h->ev + 2 * ((h->n_timed - 1 - back) % h->timing);
    HIPCHK(hipEventSynchronize(pair[1]));
    HIPCHK(hipEventElapsedTime(ms, pair[0], pair[1]));
    return BMPC_OK;
}
extern "C" int bmpc_last_kernel_ms(bmpc_handle *h, float *ms) { return bmpc_kernel_ms(h, 0, ms); }
// ---- receding-horizon streams: device-side packing / post-processing, one 64-lane wave per stream ----
__global__ void __launch_bounds__(64) bmpc_stream_pack_kernel(int N, int S, int B, const double *path, int path_stride, double *ss, const double *rb,
                                                             double *p, double *x0, double *dual, const double *xlast, double lvl_c, double lvl_lo, double lvl_hi) {
    __shared__ double sh[bmpcs::SH_LEN];
    const int b = blockIdx.x;
    bmpcs::stream_pack(N, S, path + (long long)b * path_stride, path_stride / bmpcs::PT_LEN, ss + (long long)b * bmpcs::ss_len(N), rb + (long long)b * bmpcs::RB_LEN,
                       p + (long long)b * (141 + 91 * S), x0 + (long long)b * 44 * N, dual ? dual + (long long)b * (57 * N + 2) : nullptr,
                       xlast ? xlast + (long long)b * 44 * N : nullptr, sh, threadIdx.x, 64, lvl_c, lvl_lo, lvl_hi);
}
__global__ void __launch_bounds__(64) bmpc_stream_post_kernel(int N, int S, int B, double h, const double *path, int path_stride, double *ss, double *rb,
                                                             const double *x, const double *g, const int *status, double *traj, int flags, double rt_tol, double rt_row_cap) {
    __shared__ double sh[bmpcs::SH_LEN];
    const int b = blockIdx.x;
    bmpcs::stream_post(N, S, h, path + (long long)b * path_stride, path_stride / bmpcs::PT_LEN, ss + (long long)b * bmpcs::ss_len(N), rb + (long long)b * bmpcs::RB_LEN,
                       x + (long long)b * 44 * N, g + (long long)b * 43 * N, status[b], traj + (long long)b * bmpcs::tr_len(N), flags, rt_tol, sh, threadIdx.x, 64, rt_row_cap);
}
// (the fused one-launch tick kernels of one wave per stream live in bmpc_tick.hip, those of the teams in bmpc_team.hip)
// enqueues the fused tick on `st` (direct launch or inside a capture)
static int enqueue_tick(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, double *robot, double *p, double *x0, double *dual_state,
                        int max_iter, double *x, double *g, int *iters, int *status, double *kkt, double *traj, int flags, hipStream_t st, bool capturing) {
    if (h->closed) return BMPC_ERR_ARG;
    if (!capturing) { const int rc_ = order_before(h, st); if (rc_ != BMPC_OK) return rc_; }
    KArgs a; a.N = h->N; a.S = h->S; a.B = B; a.h = h->h;
    a.o.tol = h->o.tol; a.o.max_iter = max_iter > 0 ? max_iter : h->o.max_iter; a.o.mu_init = h->o.mu_init; a.o.mu_min_fac = h->o.mu_min_fac;
    a.o.slack_push = h->o.slack_push; a.o.exact_hessian = h->o.exact_hessian; a.o.verbose = 0; a.o.mu_warm = h->o.mu_warm; a.o.stall_window = h->o.stall_window; a.o.bound_margin = h->o.bound_margin;
    a.o.restoration = h->resto_on; a.o.resto_short = h->resto_short; a.o.resto_cap = h->resto_cap; a.o.start_rollout = h->start_rollout; a.o.hold_mu = h->hold_mu; a.o.retry_cap = 0;
    a.p = p; a.x0 = x0; a.x = x; a.g = g; a.lam_g = nullptr; a.lam_x = nullptr; a.f = nullptr; a.kkt = kkt; a.iters = iters; a.status = status;
    a.state = dual_state; a.latency_us = h->latency_us; a.budget_ticks = (long long)(h->rt_budget_us * 100.0);
    if (B > h->scr_waves) return BMPC_ERR_ARG;
    a.scratch = h->scratch; a.scr_stride = h->scr_stride; a.counter = h->counter; a.prof = h->prof;
    SArgs s; s.path = path; s.path_stride = path_entries * bmpcs::PT_LEN; s.ss = sstate; s.rb = robot; s.traj = traj; s.flags = flags; s.rt_tol = h->rt_viol_tol; s.rt_row_cap = h->rt_row_cap; s.lvl_c = h->level_c; s.lvl_lo = h->level_lo; s.lvl_hi = h->level_hi;
    const bool timed = !capturing && h->timing != 0;
    hipEvent_t *pair = nullptr;
    if (timed) { int rc = timing_slot(h, &pair); if (rc != BMPC_OK) return rc; HIPCHK(hipEventRecord(pair[0], st)); }
    // The post-processing of a fused tick needs the final solution, so here the restoration phase runs INSIDE the kernel (instantiations with
    // RESTO); a time-budgeted real-time tick never gets as far as a jam (six short steps) and runs the lean instantiation with the phase off.
    const bool resto = h->resto_on != 0 && a.budget_ticks == 0;
    a.o.restoration = resto ? h->resto_on : 0; a.counter2 = nullptr; a.rcount = nullptr; a.order = nullptr;      // (the handle's MODE, not a flag: 2 = after a numerical breakdown only, as every other launch shape runs it)
    if (use_team(h, B)) HIPCHK(bmpc_team_launch_tick(BMPC_TEAM_NW, resto, &a, &s, B, st));
    else HIPCHK(bmpc_tick_launch(h->N <= 11 && h->S <= bmpc::SMAX_ZLDS, resto, &a, &s, B, st));      // (long horizons, 5 or 6 path segments: iterate in the workspace)
    if (timed) { HIPCHK(hipEventRecord(pair[1], st)); h->n_timed++; }
    if (!capturing) return order_after(h, st);
    return BMPC_OK;
}
static bool tick_fusable(const bmpc_handle *h, int B) { return B <= (use_team(h, B) ? h->team_grid : h->grid); }      // stream b = workgroup b: every stream needs a resident workgroup
extern "C" int bmpc_stream_tick(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, double *robot, double *p, double *x0,
                                double *dual_state, int max_iter, double *x, double *g, int *iters, int *status, double *kkt, double *traj, int flags,
                                void *hip_stream) {
    if (!h || B < 0 || max_iter < 0 || path_entries < h->S + 1 || (B > 0 && (!path || !sstate || !robot || !p || !x0 || !x || !g || !status || !traj))) return BMPC_ERR_ARG;
    if (h->N > bmpcs::STREAM_NMAX) return BMPC_ERR_ARG;      // (the solver's own limit)
    if (B == 0) return BMPC_OK;
    { const int rc_ = reserve_for_batch(h, B); if (rc_ != BMPC_OK) return rc_; }
    hipStream_t st = (hipStream_t)hip_stream;
    if (tick_fusable(h, B)) return enqueue_tick(h, B, path, path_entries, sstate, robot, p, x0, dual_state, max_iter, x, g, iters, status, kkt, traj, flags, st, false);
    // real-time mode: the warm start continues from the iterate of the previous tick on every launch shape (fused or not)
    int rc = bmpc_stream_pack_rt(h, B, path, path_entries, sstate, robot, p, x0, dual_state, (flags & 2) ? x : nullptr, st);
    if (rc == BMPC_OK) rc = enqueue_solve(h, B, p, x0, dual_state, max_iter, x, g, nullptr, nullptr, nullptr, iters, status, kkt, st, h->timing != 0);
    if (rc == BMPC_OK) rc = bmpc_stream_post(h, B, path, path_entries, sstate, robot, x, g, status, traj, flags, st);
    return rc;
}

extern "C" int bmpc_stream_set_rt_feasibility_tol(bmpc_handle *h, double tol) {
    if (!h || !(tol > 0)) return BMPC_ERR_ARG;
    h->rt_viol_tol = tol;      // read at launch / capture time: re-capture a tick graph after changing it
    return BMPC_OK;
}
extern "C" int bmpc_stream_set_rt_position_row_cap(bmpc_handle *h, double cap_m2) {
    if (!h || !(cap_m2 >= 0.0)) return BMPC_ERR_ARG;
    h->rt_row_cap = cap_m2;      // read at launch / capture time: re-capture a tick graph after changing it
    return BMPC_OK;
}
extern "C" int bmpc_stream_set_time_budget(bmpc_handle *h, double microseconds) {
    if (!h || !(microseconds >= 0.0) || microseconds > 1e7) return BMPC_ERR_ARG;
    h->rt_budget_us = microseconds;      // read at launch / capture time: re-capture a tick graph after changing it
    return BMPC_OK;
}
extern "C" int bmpc_stream_lengths(const bmpc_handle *h, int *path_entry, int *state, int *robot, int *traj) {
    if (!h) return BMPC_ERR_ARG;
    if (path_entry) *path_entry = bmpcs::PT_LEN; if (state) *state = bmpcs::ss_len(h->N); if (robot) *robot = bmpcs::RB_LEN; if (traj) *traj = bmpcs::tr_len(h->N);
    return BMPC_OK;
}
extern "C" int bmpc_stream_pack(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, const double *robot, double *p, double *x0,
                                double *dual_state, void *hip_stream) {
    return bmpc_stream_pack_rt(h, B, path, path_entries, sstate, robot, p, x0, dual_state, nullptr, hip_stream);
}
extern "C" int bmpc_stream_pack_rt(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, const double *robot, double *p, double *x0,
                                   double *dual_state, const double *xlast, void *hip_stream) {
    if (!h || B < 0 || path_entries < h->S + 1 || (B > 0 && (!path || !sstate || !robot || !p || !x0))) return BMPC_ERR_ARG;
    if (h->N > bmpcs::STREAM_NMAX) return BMPC_ERR_ARG;      // (the solver's own limit)
    if (B == 0) return BMPC_OK;
    if (h->closed) return BMPC_ERR_ARG;
    hipLaunchKernelGGL(bmpc_stream_pack_kernel, dim3(B), dim3(64), 0, (hipStream_t)hip_stream, h->N, h->S, B, path, path_entries * bmpcs::PT_LEN,
                       sstate, robot, p, x0, dual_state, xlast, h->level_c, h->level_lo, h->level_hi);
    HIPCHK(hipGetLastError());
    return BMPC_OK;
}
extern "C" int bmpc_stream_post(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, double *robot, const double *x, const double *g,
                                const int *status, double *traj, int flags, void *hip_stream) {
    if (!h || B < 0 || path_entries < h->S + 1 || (B > 0 && (!path || !sstate || !robot || !x || !g || !status || !traj))) return BMPC_ERR_ARG;
    if (h->N > bmpcs::STREAM_NMAX) return BMPC_ERR_ARG;      // (the solver's own limit)
    if (B == 0) return BMPC_OK;
    hipLaunchKernelGGL(bmpc_stream_post_kernel, dim3(B), dim3(64), 0, (hipStream_t)hip_stream, h->N, h->S, B, h->h, path,
                       path_entries * bmpcs::PT_LEN, sstate, robot, x, g, status, traj, flags, h->rt_viol_tol, h->rt_row_cap);
    HIPCHK(hipGetLastError());
    return BMPC_OK;
}
// one closed-loop tick {pack, solve (warm-started, max_iter), post} captured into a hipGraph
extern "C" int bmpc_stream_graph_create(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, double *robot, double *p, double *x0,
                                        double *dual_state, int max_iter, double *x, double *g, int *iters, int *status, double *kkt, double *traj,
                                        int flags, bmpc_graph **out) {
    if (!h || !out || B < 1 || max_iter < 0 || path_entries < h->S + 1 || !path || !sstate || !robot || !p || !x0 || !x || !g || !status || !traj) return BMPC_ERR_ARG;
    if (h->N > bmpcs::STREAM_NMAX) return BMPC_ERR_ARG;      // (the solver's own limit)
    { const int rc_ = reserve_for_batch(h, B); if (rc_ != BMPC_OK) return rc_; }
    hipStream_t cs;
    HIPCHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    bmpc_graph *gr = new (std::nothrow) bmpc_graph();
    if (!gr) { hipStreamDestroy(cs); return BMPC_ERR_ARG; }
    gr->h = h; gr->graph = nullptr; gr->exec = nullptr;
    int rc = BMPC_OK;
    if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) rc = BMPC_ERR_HIP;
    if (rc == BMPC_OK) {
        if (tick_fusable(h, B)) rc = enqueue_tick(h, B, path, path_entries, sstate, robot, p, x0, dual_state, max_iter, x, g, iters, status, kkt, traj, flags, cs, true);
        else {
            rc = bmpc_stream_pack_rt(h, B, path, path_entries, sstate, robot, p, x0, dual_state, (flags & 2) ? x : nullptr, cs);
            if (rc == BMPC_OK) rc = enqueue_solve(h, B, p, x0, dual_state, max_iter, x, g, nullptr, nullptr, nullptr, iters, status, kkt, cs, false, true);
            if (rc == BMPC_OK) rc = bmpc_stream_post(h, B, path, path_entries, sstate, robot, x, g, status, traj, flags, cs);
        }
        hipError_t e = hipStreamEndCapture(cs, &gr->graph);
        if (rc == BMPC_OK && e != hipSuccess) rc = BMPC_ERR_HIP;
    }
    if (rc == BMPC_OK && hipGraphInstantiate(&gr->exec, gr->graph, nullptr, nullptr, 0) != hipSuccess) rc = BMPC_ERR_HIP;
    hipStreamDestroy(cs);
    if (rc != BMPC_OK) { if (gr->exec) hipGraphExecDestroy(gr->exec); if (gr->graph) hipGraphDestroy(gr->graph); delete gr; return rc; }
    *out = gr; h->graphs_alive++; h->refs++;
    return BMPC_OK;
}

#ifdef BMPC_MARKS
// diagnostic compile only (-S): textual markers in the ISA at the phase stamps, to count static instructions per phase
#define BMPC_PROF(W, id) asm volatile("s_nop 0 ; BMPCMARK " #id ::: "memory");
#endif
#ifdef BMPC_PROFILE
// diagnostic build only: accumulated lane-0 cycle counts per phase (16 slots), then reset
extern "C" int bmpc_get_profile(bmpc_handle *h, unsigned long long *out) {
    if (!h || !out) return BMPC_ERR_ARG;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, h->prof, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(h->prof, 0, 32 * sizeof(unsigned long long)));
    return BMPC_OK;
}
#endif
extern "C" int bmpc_launch_info(const bmpc_handle *h, int *grid, int *lds_bytes, long long *scratch_bytes) {
    if (!h) return BMPC_ERR_ARG;
    if (grid) *grid = h->grid; if (lds_bytes) *lds_bytes = (int)(bmpc::L_SIZE * sizeof(double)); if (scratch_bytes) *scratch_bytes = h->scr_stride * 8;
    return BMPC_OK;
}
