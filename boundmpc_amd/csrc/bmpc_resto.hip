// bmpc_resto.hip -- gfx950 RESTORATION kernels of the batched BoundMPC OCP solver (round 5).
//
// The batch kernels (bmpc_hip.hip: one wave per problem; bmpc_team.hip: teams) are compiled without the restoration phase: carrying the
// elastic-row code costs their hot path 6 % through register allocation alone (measured, profiles/r05_a_ab_restoration_in_kernel.txt).  A
// problem whose main phase is jammed or stalled leaves them with the internal status 4 and its iterate in x.  The library starts the kernel
// below right behind every batch kernel of a handle with the restoration phase on: a wave returns at once when nothing jammed (one word),
// otherwise the waves walk the batch, and a problem with status 4 is continued by the same wave program compiled WITH the phase
// (wave_solve<ZLDS, false, true>), which starts in the restoration phase from that iterate (Problem::resto_from) and runs the solve to its
// end -- feasible point and main phase again, or status 2.  Entering the restoration phase discards everything but the iterate, so the
// hand-over computes the same numbers as a kernel that carries the phase (the fused closed-loop ticks do; tests/emu runs both).
// Always one wave per problem: the jammed problems of a batch are few.  The launch function is called from the C ABI in bmpc_hip.hip.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "bmpc_gpu_common.h"
#define LANES_BEGIN { int lane_ = threadIdx.x; asm volatile("" : "+v"(lane_)); const int lane = lane_; (void)lane;
#define LANES_END } __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#include "bmpc_wave.inl"

typedef KArgsT<bmpc::Opts> KArgs;

template <bool ZLDS>
__global__ void __launch_bounds__(64, 1) bmpc_resto_kernel(KArgs a) {
    __shared__ double lds[bmpc::L_SIZE];
    // Two uses.  Behind a batch kernel (a.rcount set): continue what it left with status 4.  As THE batch kernel (a.rcount == NULL; round 6, long horizons
    // with the full restoration phase, bmpc_hip.hip enqueue_solve): every problem from its x0, the phase inside the solve -- the same results (entering the
    // phase discards everything but the iterate: the hand-over loses nothing), but the continuations sit in the work queue instead of running on a
    // handful of waves behind the batch (configs[3]: 218 -> 1xx ms with mode 1; profiles/r06_h_configs3_failures.txt).
    const bool fresh = a.rcount == nullptr;
    if (!fresh && *(volatile int *)a.rcount == 0) return;      // nothing jammed in this batch (wave-uniform: one word)
    bmpc::Wave W; W.N = a.N; W.S = a.S; W.h = a.h; W.o = a.o; W.L = lds; W.G = bmpc::make_gptr(a.scratch + (long long)blockIdx.x * a.scr_stride); W.wv = 0;
    W.deadline = 0; W.tprev = 0; W.it_base = 0;
    const int np = 141 + 91 * a.S, nw = a.N * bmpc::NZ, ng = a.N * bmpc::NG;
    for (;;) {
        int b = 0, st = 0;
        if (threadIdx.x == 0) { b = atomicAdd(fresh ? a.counter : a.counter2, 1); st = (!fresh && (unsigned)b < (unsigned)a.B) ? a.status[b] : 0; }
        b = __builtin_amdgcn_readfirstlane(b); st = __builtin_amdgcn_readfirstlane(st);
        if ((unsigned)b >= (unsigned)a.B) break;             // every wave reaches this exit: the queue is finite
        if (!fresh && st != 4) continue;
        if (fresh && a.order) b = __builtin_amdgcn_readfirstlane(a.order[b]);      // longest-expected-first order (queue_order_kernel)
        bmpc::Problem pr;
        pr.p = a.p + (long long)b * np; pr.x0 = (fresh ? a.x0 : a.x) + (long long)b * nw;      // continuation: the iterate the batch kernel left (read before x is rewritten)
        pr.x = a.x + (long long)b * nw; pr.g = a.g ? a.g + (long long)b * ng : nullptr;
        pr.lam_g = a.lam_g ? a.lam_g + (long long)b * ng : nullptr; pr.lam_x = a.lam_x ? a.lam_x + (long long)b * nw : nullptr;
        pr.f = a.f ? a.f + b : nullptr; pr.kkt = a.kkt ? a.kkt + b : nullptr;
        pr.iters = a.iters + b; pr.status = a.status + b;
        pr.state = a.state ? a.state + (long long)b * (a.N * bmpc::NI + 2) : nullptr;
        pr.resto_from = fresh ? -1 : a.iters[b];
        const long long t0_ = a.latency_us ? (long long)wall_clock64() : 0;
        bmpc::wave_solve_retry<ZLDS, false, true>(W, pr, fresh ? nullptr : a.x0 + (long long)b * nw);      // (a continuation's second attempt is a fresh solve from the caller's x0)
        __syncthreads();
        if (a.latency_us && threadIdx.x == 0) a.latency_us[b] = (fresh ? 0.0 : a.latency_us[b]) + (double)((long long)wall_clock64() - t0_) * 0.01;   // continuation: on top of the batch kernel's share
    }
}

hipError_t bmpc_resto_launch(bool zlds, const void *kargs, int grid, hipStream_t st) {
    KArgs a; memcpy(&a, kargs, sizeof(a));
    if (zlds) hipLaunchKernelGGL(bmpc_resto_kernel<true>, dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(bmpc_resto_kernel<false>, dim3(grid), dim3(64), 0, st, a);
    return hipGetLastError();
}
