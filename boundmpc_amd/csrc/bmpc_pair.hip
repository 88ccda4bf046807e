// bmpc_pair.hip -- gfx950 PAIR kernel of the batched BoundMPC OCP solver: two cooperating waves per problem on the one-wave budget.
//
// The one-wave kernel (bmpc_hip.hip) runs one problem per 64-lane wave; the 4-wave teams (bmpc_team.hip) own a whole CU per problem (their
// workspace rows live in its 160 KB of LDS), so only 256 problems are resident.  Here the same wave program (bmpc_wave.inl compiled with
// BMPC_NW = 2 and BMPC_WSG, namespace bmpcp) runs on a 128-thread workgroup that keeps the ONE-WAVE budget per problem -- 40 KB of LDS, the
// workspace in the global slab -- so two pairs share a CU: 512 problems resident.  Roles: wave 0 runs the recursions (adjoint, Riccati,
// forward); wave 1 the references / objective half of every evaluation beside the kinematics and, inside the Riccati sweep, the staging (its
// own register prefetch) and the recursion-independent half of the next stage's node-cost add, the q~ rows and t6; the item-parallel row
// passes run over all 128 lanes.  Same reduction orders as the team text: results equal the one-wave kernel's up to the order of a few sums
// behind discrete decisions (bit-equal on the bench batches; 2 of 8192 problems differ by 1e-12).  Launched from the C ABI in bmpc_hip.hip
// for batches between the resident teams and the resident pairs (256 < B <= 512): 1.18-1.20x the one-wave kernel there.
//
// Built as the answer to "a second wave on every SIMD" (round 6): compiled for TWO waves per SIMD (-DBMPC_PAIR_EU=2: 256 registers per wave,
// four pairs per CU, 1024 problems resident) the same text needs 744 B of scratch per lane and runs 0.58x the one-wave kernel at B = 1024
// (4.68 vs 2.70 ms; profiles/r06_a_pair_occupancy_ab.txt has the A/B and the resource table per wave role): the LDS budget holds (40 896 B),
// the register budget does not -- at 512 registers per wave the pair is 1.2x the one-wave kernel per problem.  The product build is the
// 512-register one (BMPC_PAIR_EU = 1).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

#define BMPC_NW 2
#define BMPC_WSG 1
#define BMPC_NAMESPACE bmpcp
#include "bmpc_gpu_common.h"

#define BMPC_LANE_ID (threadIdx.x & 63)
// a phase of ONE wave of the pair (lane = 0..63); opaque lane id as in the one-wave build
#define LANES_BEGIN { int lane_ = threadIdx.x & 63; asm volatile("" : "+v"(lane_)); const int lane = lane_; (void)lane;
#define LANES_END } __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
// workgroup barrier with workgroup-scope release / acquire (hand-overs through the workspace slab: the two waves sit on one CU and share its L1)
#define TEAM_SYNC() __syncthreads()
// hand-overs through LDS only: LDS operations complete, the waves meet; outstanding vector-memory loads (the register prefetch of a sweep,
// of the helper) stay in flight
#define TEAM_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define WIDE_BEGIN LANES_BEGIN const int wl = W.wv * 64 + lane; (void)wl;
#define WIDE_END LANES_END TEAM_SYNC();
#ifdef BMPC_PAIR_ROLE
// diagnostic compile only (resource table per wave role, tests/kernel_resources.py --pair-roles): the solo regions of the OTHER role are compiled out
#define SOLO_BEGIN(w) if (W.wv == (w) && (w) == BMPC_PAIR_ROLE) {
#else
#define SOLO_BEGIN(w) if (W.wv == (w)) {
#endif
#define SOLO_END }

#ifdef BMPC_MARKS
// diagnostic compile only (-S): textual markers in the ISA at the phase stamps (tests/isa_phase_stats.py)
#define BMPC_PROF(W, id) asm volatile("s_nop 0 ; BMPCMARK " #id ::: "memory");
#endif
#ifdef BMPC_PROFILE
// diagnostic build only: per-phase cycle stamps of lane 0 of wave 0
#define BMPC_PROF(W, id) { long long now_ = clock64(); if (threadIdx.x == 0) { ((long long *)((W).L + bmpcp::L_PROF))[id] += now_ - (W).tprev; } (W).tprev = now_; }
#endif

#include "bmpc_wave.inl"

typedef KArgsT<bmpcp::Opts> KArgsPair;
static_assert(bmpcp::NW == 2, "pair size");

#ifndef BMPC_PAIR_EU
#define BMPC_PAIR_EU 1      // waves per SIMD the kernel is compiled for (2 = the 256-register experiment of the header)
#endif
__global__ void __launch_bounds__(128, BMPC_PAIR_EU) bmpc_pair_solve_kernel(KArgsPair a) {
    __shared__ double lds[bmpcp::L_SIZE];
    bmpcp::Wave W; W.N = a.N; W.S = a.S; W.h = a.h; W.o = a.o; W.L = lds; W.G = bmpcp::make_gptr(a.scratch + (long long)blockIdx.x * a.scr_stride);
    W.wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    W.deadline = 0; W.it_base = 0;
    const int np = 141 + 91 * a.S, nw = a.N * bmpcp::NZ, ng = a.N * bmpcp::NG;
#ifdef BMPC_PROFILE
    if (threadIdx.x < 32) ((long long *)(lds + bmpcp::L_PROF))[threadIdx.x] = 0;
    __syncthreads();
    W.tprev = clock64();
#endif
    for (;;) {
        // one lane takes the next problem off the queue for the pair
        if (threadIdx.x == 0) lds[bmpcp::L_TFLAG + 1] = (double)atomicAdd(a.counter, 1);
        __syncthreads();
        const int b = __builtin_amdgcn_readfirstlane((int)lds[bmpcp::L_TFLAG + 1]);
        __syncthreads();                 // both waves have read the word before the next round rewrites it
        if ((unsigned)b >= (unsigned)a.B) break;             // both waves of every pair reach this exit: the queue is finite
        bmpcp::Problem pr;
        pr.p = a.p + (long long)b * np; pr.x0 = a.x0 + (long long)b * nw;
        pr.x = a.x ? a.x + (long long)b * nw : nullptr; pr.g = a.g ? a.g + (long long)b * ng : nullptr;
        pr.lam_g = a.lam_g ? a.lam_g + (long long)b * ng : nullptr; pr.lam_x = a.lam_x ? a.lam_x + (long long)b * nw : nullptr;
        pr.f = a.f ? a.f + b : nullptr; pr.kkt = a.kkt ? a.kkt + b : nullptr;
        pr.iters = a.iters ? a.iters + b : nullptr; pr.status = a.status ? a.status + b : nullptr;
        pr.state = a.state ? a.state + (long long)b * (a.N * bmpcp::NI + 2) : nullptr;
        pr.resto_from = -1;
        const long long t0_ = a.latency_us ? (long long)wall_clock64() : 0;
        bmpcp::wave_solve<true>(W, pr);
        __syncthreads();
        if (a.rcount && threadIdx.x == 0 && *pr.status == 4) atomicAdd(a.rcount, 1);      // jammed: the (one-wave) restoration kernel continues it (bmpc_resto.hip)
        if (a.latency_us && threadIdx.x == 0) a.latency_us[b] = (double)((long long)wall_clock64() - t0_) * 0.01;   // constant 100 MHz counter
    }
#ifdef BMPC_PROFILE
    if (threadIdx.x < 32 && a.prof) atomicAdd(a.prof + threadIdx.x, (unsigned long long)((long long *)(lds + bmpcp::L_PROF))[threadIdx.x]);
#endif
}

int bmpc_pair_blocks_per_cu(void) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bmpc_pair_solve_kernel, 128, 0) != hipSuccess) return 0;
    return per_cu;
}
int bmpc_pair_nmax(void) { return bmpcp::TEAM_NMAX; }
int bmpc_pair_lds_bytes(void) { return (int)(bmpcp::L_SIZE * sizeof(double)); }
long long bmpc_pair_scr_stride(int N) { return bmpcp::make_scr(N).size; }
hipError_t bmpc_pair_launch_solve(const void *kargs, int grid, hipStream_t st) {
    KArgsPair a; memcpy(&a, kargs, sizeof(a));
    hipLaunchKernelGGL(bmpc_pair_solve_kernel, dim3(grid), dim3(128), 0, st, a);
    return hipGetLastError();
}
