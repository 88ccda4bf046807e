// bmpc_tick.hip -- gfx950 fused closed-loop tick kernels (one wave per stream) of the batched BoundMPC OCP solver: {pack, solve, post} of a
// stream in ONE launch.  A translation unit of its own since round 5: the kernel exists with and without the restoration phase in the solver
// (RESTO), for both placements of the iterate (ZLDS) -- four instantiations of the whole wave program.  The launch function is called from the
// C ABI in bmpc_hip.hip; the team version lives in bmpc_team.hip.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "bmpc_gpu_common.h"
#define LANES_BEGIN { int lane_ = threadIdx.x; asm volatile("" : "+v"(lane_)); const int lane = lane_; (void)lane;
#define LANES_END } __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#include "bmpc_wave.inl"
#define BMPCS_SYNC() __syncthreads()
#include "bmpc_stream.inl"

typedef KArgsT<bmpc::Opts> KArgs;

// ---- one closed-loop tick of a stream in ONE launch: {pack, solve, post} by the wave that owns the stream (both instantiations of the solver) ----
// The three steps of a tick are each "one wave per stream" and strictly sequential per stream, so they need no grid-wide boundary
// between them: as three kernels + the work-queue reset they cost three launch ramps, three drains and ~130 us of launch overhead
// per tick at 1 kHz (profiles/r03_*_stream_trace.txt); here stream b is block b (B <= resident waves: no work queue, no reset node),
// the stream functions use the reduction area of the solver's LDS, and the hand-over of p, x0 -> solver -> x, g, status goes through
// global memory of the same wave in program order.
template <bool ZLDS, bool RESTO>
__global__ void __launch_bounds__(64, 1) bmpc_stream_tick_kernel(KArgs a, SArgs s) {
    __shared__ double lds[bmpc::L_SIZE];
    const long long tk0_ = a.budget_ticks ? BMPC_NOW() : 0;
    const int b = blockIdx.x;
    if (b >= a.B) return;
    const int np = 141 + 91 * a.S, nw = a.N * bmpc::NZ, ng = a.N * bmpc::NG;
    double *sh = lds + bmpc::L_RED;
    static_assert(bmpcs::SH_LEN <= 6 * 64, "the stream functions' LDS words must fit into the solver's reduction area");
    const double *path = s.path + (long long)b * s.path_stride;
    double *ss = s.ss + (long long)b * bmpcs::ss_len(a.N), *rb = s.rb + (long long)b * bmpcs::RB_LEN;
    double *p = const_cast<double *>(a.p) + (long long)b * np, *x0 = const_cast<double *>(a.x0) + (long long)b * nw;
    double *dual = a.state ? a.state + (long long)b * (a.N * bmpc::NI + 2) : nullptr;
    // A stream that has lost its plan (N consecutive ticks without an accepted solution: BoundMPC.step() returns five Nones there and the
    // reference node stops, BoundMPC.py:498-506, bound_mpc_node.py:318) is not ticked any further: its problems are the ones nobody could
    // solve (tests/golden/g13_hard_ticks.npz), each would run to the stall test or the iteration cap, and a tick lasts as long as its slowest stream.
    if (ss[bmpcs::SS_ERRCNT] >= (double)a.N) {
        if (threadIdx.x == 0) { a.status[b] = 3; if (a.iters) a.iters[b] = 0; if (a.kkt) a.kkt[b] = 0.0; if (a.latency_us) a.latency_us[b] = 0.0; }
        return;
    }
    bmpcs::stream_pack(a.N, a.S, path, s.path_stride / bmpcs::PT_LEN, ss, rb, p, x0, dual, (s.flags & 2) ? a.x + (long long)b * nw : nullptr, sh, threadIdx.x, 64, s.lvl_c, s.lvl_lo, s.lvl_hi);
    __syncthreads();
    bmpc::Wave W; W.N = a.N; W.S = a.S; W.h = a.h; W.o = a.o; W.L = lds; W.G = bmpc::make_gptr(a.scratch + (long long)b * a.scr_stride); W.wv = 0;
    bmpc::Problem pr;
    pr.p = p; pr.x0 = x0; pr.x = a.x + (long long)b * nw; pr.g = a.g + (long long)b * ng; pr.lam_g = nullptr; pr.lam_x = nullptr;
    pr.f = nullptr; pr.kkt = a.kkt ? a.kkt + b : nullptr; pr.iters = a.iters ? a.iters + b : nullptr; pr.status = a.status + b; pr.state = dual;
    const long long t0_ = a.latency_us ? (long long)wall_clock64() : 0;
    W.deadline = a.budget_ticks ? tk0_ + a.budget_ticks : 0; W.it_base = 0;
    pr.resto_from = -1;
    bmpc::wave_solve<ZLDS, true, RESTO>(W, pr);
    __syncthreads();
    if (a.latency_us && threadIdx.x == 0) a.latency_us[b] = (double)((long long)wall_clock64() - t0_) * 0.01;
    bmpcs::stream_post(a.N, a.S, a.h, path, s.path_stride / bmpcs::PT_LEN, ss, rb, pr.x, pr.g, a.status[b], s.traj + (long long)b * bmpcs::tr_len(a.N), s.flags, s.rt_tol,
                       sh, threadIdx.x, 64, s.rt_row_cap);
}

hipError_t bmpc_tick_launch(bool zlds, bool resto, const void *kargs, const SArgs *s, int B, hipStream_t st) {
    KArgs a; memcpy(&a, kargs, sizeof(a));
    if (zlds && resto) hipLaunchKernelGGL((bmpc_stream_tick_kernel<true, true>), dim3(B), dim3(64), 0, st, a, *s);
    else if (zlds) hipLaunchKernelGGL((bmpc_stream_tick_kernel<true, false>), dim3(B), dim3(64), 0, st, a, *s);
    else if (resto) hipLaunchKernelGGL((bmpc_stream_tick_kernel<false, true>), dim3(B), dim3(64), 0, st, a, *s);
    else hipLaunchKernelGGL((bmpc_stream_tick_kernel<false, false>), dim3(B), dim3(64), 0, st, a, *s);
    return hipGetLastError();
}
