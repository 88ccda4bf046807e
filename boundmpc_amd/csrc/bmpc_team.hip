// bmpc_team.hip -- gfx950 TEAM kernels of the batched BoundMPC OCP solver: a workgroup of NW cooperating waves per problem.
//
// For batches that leave SIMDs idle (B <= resident workgroups / NW: the 256 closed-loop streams of BASELINE configs[4], the single
// solver(...) call per tick that is the reference's own use, BoundMPC.py:446-453) one wave per problem keeps 1 of the 4 SIMDs of a CU
// busy.  Here the same wave program (bmpc_wave.inl compiled with BMPC_NW waves, namespace bmpct) runs on a 64 NW-thread workgroup: the
// item-parallel passes of an interior-point iteration run over all 64 NW lanes, independent sequential pieces run side by side on
// different waves, the recursions (adjoint / Riccati / forward sweep) stay on wave 0.  At 512 registers per wave one team owns a CU
// (one wave per SIMD): 256 problems resident.  The launch functions are called from the C ABI in bmpc_hip.hip.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

#ifndef BMPC_NW
#define BMPC_NW 4
#endif
#define BMPC_NAMESPACE bmpct
#include "bmpc_gpu_common.h"

#define BMPC_LANE_ID (threadIdx.x & 63)
// a phase of ONE wave of the team (lane = 0..63); opaque lane id as in the one-wave build
#define LANES_BEGIN { int lane_ = threadIdx.x & 63; asm volatile("" : "+v"(lane_)); const int lane = lane_; (void)lane;
#define LANES_END } __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
// workgroup barrier: s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier with workgroup-scope release / acquire.  The waves of a workgroup sit on
// one CU and share its L1, so workspace words one wave stored are visible to the others behind it.
#define TEAM_SYNC() __syncthreads()
// the same for hand-overs that go through LDS only: LDS operations complete (lgkmcnt(0)), the waves meet, but outstanding vector-memory
// loads -- a sweep's register prefetch of a later stage -- are NOT waited for (what the vmcnt(0) of __syncthreads() would do: a full
// round trip to the slab per barrier)
#define TEAM_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define WIDE_BEGIN LANES_BEGIN const int wl = W.wv * 64 + lane; (void)wl;
#define WIDE_END LANES_END TEAM_SYNC();
#define SOLO_BEGIN(w) if (W.wv == (w)) {
#define SOLO_END }

#ifdef BMPC_PROFILE
// diagnostic build only: per-phase cycle stamps of lane 0 of wave 0
#define BMPC_PROF(W, id) { long long now_ = clock64(); if (threadIdx.x == 0) { ((long long *)((W).L + bmpct::L_PROF))[id] += now_ - (W).tprev; } (W).tprev = now_; }
#endif

#include "bmpc_wave.inl"
// the stream functions run on wave 0 of the team (64 cooperating lanes, as in the one-wave build): their phase boundary is a wavefront fence
#define BMPCS_SYNC() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#include "bmpc_stream.inl"

typedef KArgsT<bmpct::Opts> KArgsTeam;
static_assert(bmpct::NW == BMPC_NW, "team size");

__global__ void __launch_bounds__(64 * BMPC_NW, 1) bmpc_team_solve_kernel(KArgsTeam a) {
    __shared__ double lds[bmpct::L_SIZE];
    bmpct::Wave W; W.N = a.N; W.S = a.S; W.h = a.h; W.o = a.o; W.L = lds; W.G = bmpct::make_gptr(a.scratch + (long long)blockIdx.x * a.scr_stride);
    W.wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    W.deadline = 0; W.it_base = 0;
    const int np = 141 + 91 * a.S, nw = a.N * bmpct::NZ, ng = a.N * bmpct::NG;
#ifdef BMPC_PROFILE
    if (threadIdx.x < 32) ((long long *)(lds + bmpct::L_PROF))[threadIdx.x] = 0;
    __syncthreads();
    W.tprev = clock64();
#endif
    for (;;) {
        // one lane takes the next problem off the queue for the whole team
        if (threadIdx.x == 0) lds[bmpct::L_TFLAG + 1] = (double)atomicAdd(a.counter, 1);
        __syncthreads();
        const int b = __builtin_amdgcn_readfirstlane((int)lds[bmpct::L_TFLAG + 1]);
        __syncthreads();                 // everyone has read the word before the next round rewrites it
        if ((unsigned)b >= (unsigned)a.B) break;             // every wave of every team reaches this exit: the queue is finite
        bmpct::Problem pr;
        pr.p = a.p + (long long)b * np; pr.x0 = a.x0 + (long long)b * nw;
        pr.x = a.x ? a.x + (long long)b * nw : nullptr; pr.g = a.g ? a.g + (long long)b * ng : nullptr;
        pr.lam_g = a.lam_g ? a.lam_g + (long long)b * ng : nullptr; pr.lam_x = a.lam_x ? a.lam_x + (long long)b * nw : nullptr;
        pr.f = a.f ? a.f + b : nullptr; pr.kkt = a.kkt ? a.kkt + b : nullptr;
        pr.iters = a.iters ? a.iters + b : nullptr; pr.status = a.status ? a.status + b : nullptr;
        pr.state = a.state ? a.state + (long long)b * (a.N * bmpct::NI + 2) : nullptr;
        pr.resto_from = -1;
        const long long t0_ = a.latency_us ? (long long)wall_clock64() : 0;
        bmpct::wave_solve<true>(W, pr);
        __syncthreads();
        if (a.rcount && threadIdx.x == 0 && *pr.status == 4) atomicAdd(a.rcount, 1);      // jammed: the (one-wave) restoration kernel continues it (bmpc_resto.hip)
        if (a.latency_us && threadIdx.x == 0) a.latency_us[b] = (double)((long long)wall_clock64() - t0_) * 0.01;   // constant 100 MHz counter
    }
#ifdef BMPC_PROFILE
    if (threadIdx.x < 32 && a.prof) atomicAdd(a.prof + threadIdx.x, (unsigned long long)((long long *)(lds + bmpct::L_PROF))[threadIdx.x]);
#endif
}

// one closed-loop tick of a stream in ONE launch by a team: wave 0 packs, the team solves, wave 0 post-processes (stream b = block b)
template <bool RESTO>
__global__ void __launch_bounds__(64 * BMPC_NW, 1) bmpc_team_tick_kernel(KArgsTeam a, SArgs s) {
    __shared__ double lds[bmpct::L_SIZE];
    const long long tk0_ = a.budget_ticks ? BMPC_NOW() : 0;
    const int b = blockIdx.x;
    if (b >= a.B) return;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int np = 141 + 91 * a.S, nw = a.N * bmpct::NZ, ng = a.N * bmpct::NG;
    double *sh = lds + bmpct::L_RED;
    static_assert(bmpcs::SH_LEN <= 6 * 64, "the stream functions' LDS words must fit into the solver's reduction area");
    const double *path = s.path + (long long)b * s.path_stride;
    double *ss = s.ss + (long long)b * bmpcs::ss_len(a.N), *rb = s.rb + (long long)b * bmpcs::RB_LEN;
    double *p = const_cast<double *>(a.p) + (long long)b * np, *x0 = const_cast<double *>(a.x0) + (long long)b * nw;
    double *dual = a.state ? a.state + (long long)b * (a.N * bmpct::NI + 2) : nullptr;
    // A stream that has lost its plan (N consecutive ticks without an accepted solution: BoundMPC.step() returns five Nones there and the
    // reference node stops, BoundMPC.py:498-506, bound_mpc_node.py:318) is not ticked any further: its problems are the ones nobody could
    // solve (tests/golden/g13_hard_ticks.npz), each would run to the stall test or the iteration cap, and a tick lasts as long as its slowest stream.
    if (ss[bmpcs::SS_ERRCNT] >= (double)a.N) {
        if (threadIdx.x == 0) { a.status[b] = 3; if (a.iters) a.iters[b] = 0; if (a.kkt) a.kkt[b] = 0.0; if (a.latency_us) a.latency_us[b] = 0.0; }
        return;
    }
    if (wv == 0) bmpcs::stream_pack(a.N, a.S, path, s.path_stride / bmpcs::PT_LEN, ss, rb, p, x0, dual, (s.flags & 2) ? a.x + (long long)b * nw : nullptr, sh, threadIdx.x, 64, s.lvl_c, s.lvl_lo, s.lvl_hi);
    __syncthreads();
    bmpct::Wave W; W.N = a.N; W.S = a.S; W.h = a.h; W.o = a.o; W.L = lds; W.G = bmpct::make_gptr(a.scratch + (long long)b * a.scr_stride); W.wv = wv;
    bmpct::Problem pr;
    pr.p = p; pr.x0 = x0; pr.x = a.x + (long long)b * nw; pr.g = a.g + (long long)b * ng; pr.lam_g = nullptr; pr.lam_x = nullptr;
    pr.f = nullptr; pr.kkt = a.kkt ? a.kkt + b : nullptr; pr.iters = a.iters ? a.iters + b : nullptr; pr.status = a.status + b; pr.state = dual;
    const long long t0_ = a.latency_us ? (long long)wall_clock64() : 0;
    W.deadline = a.budget_ticks ? tk0_ + a.budget_ticks : 0; W.it_base = 0;
    pr.resto_from = -1;
    bmpct::wave_solve<true, true, RESTO>(W, pr);
    __syncthreads();
    if (a.latency_us && threadIdx.x == 0) a.latency_us[b] = (double)((long long)wall_clock64() - t0_) * 0.01;
    if (wv == 0) bmpcs::stream_post(a.N, a.S, a.h, path, s.path_stride / bmpcs::PT_LEN, ss, rb, pr.x, pr.g, a.status[b], s.traj + (long long)b * bmpcs::tr_len(a.N), s.flags, s.rt_tol,
                                    sh, threadIdx.x, 64, s.rt_row_cap);
}

int bmpc_team_blocks_per_cu(int nw) {
    if (nw != BMPC_NW) return 0;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bmpc_team_solve_kernel, 64 * BMPC_NW, 0) != hipSuccess) return 0;
    return per_cu;
}
int bmpc_team_nmax(int nw) { return nw == BMPC_NW ? bmpct::TEAM_NMAX : 0; }
int bmpc_team_lds_bytes(int nw) { return nw == BMPC_NW ? (int)(bmpct::L_SIZE * sizeof(double)) : 0; }
hipError_t bmpc_team_launch_solve(int nw, const void *kargs, int grid, hipStream_t st) {
    if (nw != BMPC_NW) return hipErrorInvalidValue;
    KArgsTeam a; memcpy(&a, kargs, sizeof(a));
    hipLaunchKernelGGL(bmpc_team_solve_kernel, dim3(grid), dim3(64 * BMPC_NW), 0, st, a);
    return hipGetLastError();
}
hipError_t bmpc_team_launch_tick(int nw, bool resto, const void *kargs, const SArgs *s, int B, hipStream_t st) {
    if (nw != BMPC_NW) return hipErrorInvalidValue;
    KArgsTeam a; memcpy(&a, kargs, sizeof(a));
    if (resto) hipLaunchKernelGGL(bmpc_team_tick_kernel<true>, dim3(B), dim3(64 * BMPC_NW), 0, st, a, *s);
    else hipLaunchKernelGGL(bmpc_team_tick_kernel<false>, dim3(B), dim3(64 * BMPC_NW), 0, st, a, *s);
    return hipGetLastError();
}
