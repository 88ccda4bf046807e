// bmpc_gpu_common.h -- what the translation units of libboundmpc_hip.so share: the device math macros the wave program
// (bmpc_wave.inl) is written in, and the kernel argument records.  bmpc_hip.hip holds the one-wave-per-problem kernels and the C ABI,
// bmpc_team.hip the team kernels (NW cooperating waves per problem) and their launchers.
#pragma once
#include <hip/hip_runtime.h>

#define BMPC_HD __host__ __device__ __forceinline__
#define BMPC_D __device__ __forceinline__
#define BMPC_SINCOS(x, s, c) sincos(x, s, c)
#define BMPC_EXP(x) exp(x)
#define BMPC_LOG(x) log(x)
#define BMPC_SQRT(x) sqrt(x)
#define BMPC_SIN(x) sin(x)
#define BMPC_COS(x) cos(x)
#define BMPC_ATAN2(y, x) atan2(y, x)
#define BMPC_RSQRT(x) rsqrt(x)
#define BMPC_FABS(x) fabs(x)
#define BMPC_FMAX(a, b) fmax(a, b)
#define BMPC_FMIN(a, b) fmin(a, b)
#define BMPC_POW15(x) ((x) * sqrt(x))
#define BMPC_POW(x, y) pow(x, y)
#define LIDX 0
#define BMPC_WAVE_RED 1
#ifndef BMPC_NO_MFMA
#define BMPC_MFMA 1       // Schur update of the Riccati stage on the matrix cores (v_mfma_f64_16x16x4_f64)
#endif
#define BMPC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define BMPC_NOW() ((long long)wall_clock64())      // constant 100 MHz counter
#define BMPCS_OPAQUE(x) asm volatile("" : "+v"(x))      // stream functions: keeps a loaded value out of the optimiser's reach (no re-sinking of the load under a branch)

// kernel arguments of a solve; OPTS = the Opts type of the wave program's namespace (same layout in every instantiation)
template <class OPTS>
struct KArgsT {
    int N, S, B; double h; OPTS o;
    const double *p, *x0; double *x, *g, *lam_g, *lam_x, *f, *kkt; int *iters, *status;
    double *state;           // optional [B][57 N + 2] dual state of a receding-horizon stream (bmpc_solve_batch_warm)
    double *latency_us;      // optional [B]: in-kernel duration of each solve (bmpc_set_latency_buffer)
    double *scratch; long long scr_stride; int *counter; unsigned long long *prof;
    long long budget_ticks;  // fused closed-loop tick only: time budget of a tick in counts of the 100 MHz wall clock, from kernel entry (0 = none)
    int *counter2, *rcount;  // restoration kernel (bmpc_resto.hip): its work queue; number of problems the batch kernel left with status 4 (NULL: phase off)
    const int *order;        // one-wave batch kernel: the work queue hands out order[0], order[1], ... instead of 0, 1, ... (NULL: natural order; bmpc_set_queue_order)
};
// stream arguments of a fused tick
struct SArgs {
    const double *path; int path_stride; double *ss, *rb, *traj; int flags; double rt_tol; double rt_row_cap; double lvl_c, lvl_lo, lvl_hi;
};

// ---- restoration kernel (bmpc_resto.hip): continues the problems a batch kernel left with the internal status 4 ----
hipError_t bmpc_resto_launch(bool zlds, const void *kargs, int grid, hipStream_t st);

// ---- fused closed-loop tick, one wave per stream (bmpc_tick.hip); resto: the instantiation that carries the restoration phase ----
hipError_t bmpc_tick_launch(bool zlds, bool resto, const void *kargs, const SArgs *s, int B, hipStream_t st);

// ---- team kernels (bmpc_team.hip); `kargs` points at a KArgsT<...> record (the layouts are identical across instantiations) ----
// resident workgroups per CU of the team solver kernel with `nw` waves (0: no such instantiation)
int bmpc_team_blocks_per_cu(int nw);
hipError_t bmpc_team_launch_solve(int nw, const void *kargs, int grid, hipStream_t st);
hipError_t bmpc_team_launch_tick(int nw, bool resto, const void *kargs, const SArgs *s, int B, hipStream_t st);
int bmpc_team_lds_bytes(int nw);
int bmpc_team_nmax(int nw);      // longest horizon the team kernels take (their LDS holds most of the workspace)

// ---- pair kernel (bmpc_pair.hip): two cooperating waves per problem at two waves per SIMD, one-wave LDS / workspace budget per problem ----
int bmpc_pair_blocks_per_cu(void);      // resident pairs per CU (4 on an MI355X; 0: the launch configuration does not fit)
int bmpc_pair_nmax(void);               // longest horizon the pair kernel takes (the iterate-in-LDS instantiation)
int bmpc_pair_lds_bytes(void);
long long bmpc_pair_scr_stride(int N);  // doubles of workspace per pair (the one-wave layout)
hipError_t bmpc_pair_launch_solve(const void *kargs, int grid, hipStream_t st);
