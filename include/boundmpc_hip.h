/*
 * boundmpc_hip.h -- C ABI of the MI355X-native batched BoundMPC OCP solver (measurements and design notes: DESIGN.md).
 *
 * Drop-in boundary.  In the reference (Thieso/BoundMPC) the per-step optimisation is one CasADi `nlpsol('solver','ipopt',...)` Function object,
 * created at bound_mpc/bound_mpc/BoundMPC/casadi_ocp_formulation.py:389 and invoked only at BoundMPC/BoundMPC.py:446-453 (sol = solver(x0=, lbx=,
 * ubx=, lbg=, ubg=, p=)) and :456 (solver.stats()).  The reference defines no C ABI for it; these entry points carry what that call carries, batched:
 *   bmpc_create       <-> setup_optimization_problem(N, 7, nr_segs, dt, ...) (casadi_ocp_formulation.py:9-11) + the Ipopt option dict (BoundMPC.py:120-148)
 *   bmpc_get_bounds   <-> the lbu, ubu, lbg, ubg lists returned there (casadi_ocp_formulation.py:384-391)
 *   bmpc_solve_batch  <-> solver(x0=, ..., p=) -> {'x','f','g','lam_x','lam_g'} (BoundMPC.py:446-453) and solver.stats() (:456-457), one record per problem
 *   bmpc_destroy      <-> garbage collection of the Function object
 * All arithmetic is fp64.  Layouts (row-major, one problem per row):
 *   p      [B][n_p]   n_p = 141 + 91 S, parameter vector in the order of casadi_ocp_formulation.py:361-376
 *   x0, x  [B][44 N]  stage variables z_k = [u(7) u_phi | q dq ddq | p(6) | v(6) | phi dphi ddphi] (:90-153)
 *   g      [B][43 N]  constraints in the reference's order and form (:272-349); lam_g [B][43 N] (sign: L = f + lam_g . g); lam_x [B][44 N]
 *   f, kkt [B]; iters [B]; status [B]: 0 converged, 1 max_iter reached, 2 locally infeasible (restoration phase, or a stall that survived the
 *   barrier restarts of a long horizon), 3 numerical failure.  Per-problem failure is data, never an error code (BoundMPC.py:465-489).
 * lbx / ubx / lbg / ubg are structural constants of the formulation and do not cross the ABI per call (bmpc_get_bounds).
 * Ownership: the caller owns every buffer; the library owns the handle and its device workspace (allocated by the first solve or capture for
 * min(B, resident workgroups), grown on demand after a host wait for the handle's own last launch; it cannot grow while captured graphs of the
 * handle exist: capture for the largest batch first).  Errors are integer return codes.  One in-flight call per handle from the host's side.
 */
#ifndef BOUNDMPC_HIP_H
#define BOUNDMPC_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct bmpc_handle bmpc_handle;

typedef struct {
    double tol;         /* scaled KKT tolerance (Ipopt's error measure; reference 'tol': 10e-6, BoundMPC.py:121); default 1e-8 */
    int max_iter;       /* default 500 (BoundMPC.py:122) */
    double mu_init;     /* first barrier level: 0.1 for N <= 11, 3.0 for longer horizons (bmpc_default_options_for) */
    double mu_min_fac;  /* last barrier level = tol * mu_min_fac; default 0.1 */
    double slack_push;  /* smallest initial slack of an inequality row: 1e-2 for N <= 11, 0.1 for longer horizons */
    int exact_hessian;  /* 1 (default): exact Lagrangian Hessian, as the reference; 0: Gauss-Newton */
    int verbose;
    double mu_warm;     /* warm start (bmpc_solve_batch_warm): the barrier restarts at clamp(stored mu, mu_warm, mu_init); default 1e-2 */
    int stall_window;   /* stalled = the primal infeasibility has not halved over this many iterations (even; checked every half window; 0 = never);
                           default 40 for N <= 11, 20 beyond; 16 recommended for warm-started streams.  A stall starts the restoration phase
                           (N <= 11) or a barrier restart, then status 2 (longer horizons, or restoration off) */
    double bound_margin;/* joint position / velocity limits tightened by this much (rad, rad/s) inside the solver; default 0 = the reference's
                           limits (RobotModel.py:20-39); for real-time loops solved to a loose tolerance or a budget */
} bmpc_options;

enum { BMPC_OK = 0, BMPC_ERR_ARG = 1, BMPC_ERR_HIP = 2, BMPC_ERR_NOGPU = 4 };

/* sizeof(bmpc_options) of the library (the struct grew by bound_margin once; later additions are handle setters, not fields) */
int bmpc_options_size(void);
/* 16 hex digits: hash of the source text and compiler flags the library was built from (boundmpc_amd/build.py source_hash) */
const char *bmpc_build_hash(void);
int bmpc_default_options(bmpc_options *o);                 /* the N <= 11 defaults */
int bmpc_default_options_for(int N, bmpc_options *o);      /* defaults for horizon N; what bmpc_create(..., NULL, ...) uses */
const char *bmpc_error_string(int code);

/* N horizon (1..40), S path segments in the window (2..6), dt sampling time */
int bmpc_create(int N, int S, double dt, const bmpc_options *opts, bmpc_handle **out);
int bmpc_destroy(bmpc_handle *h);

/* Restoration phase -- what stands where Ipopt's filter line search falls back to its restoration phase (BoundMPC.py:135; Waechter & Biegler 2006,
 * 3.3).  A main phase that is jammed (`short_steps` consecutive steps shorter than 10 % with the primal infeasibility open; default 6), stalled or
 * numerically broken (dual residual beyond 1e12) switches to  min rho sum e  s.t. dynamics, h(x) - e <= 0, e >= 0  (rho = 1000, objective weights
 * zero) on the same Riccati recursion, and returns to the main phase from the first strictly feasible iterate, or ends the solve with status 2: converged
 * to a point of non-zero violation, `cap` iterations (default 40) without a feasible point, or the fourth call within one solve.
 * enabled: 0 never; 1 full (jam, stall, breakdown; long horizons: behind their three barrier restarts, never on a jam); 2 after a numerical
 * breakdown only.  Default 1 for N <= 11, 2 beyond.  A negative argument keeps the current value; nothing is changed when any argument is
 * invalid.  Read at launch / capture time; every launch shape (batch, fused tick, three-kernel tick) runs the handle's mode; time-budgeted
 * real-time ticks run without the phase. */
int bmpc_set_restoration(bmpc_handle *h, int enabled, int short_steps, int cap);
int bmpc_get_restoration(const bmpc_handle *h, int *enabled, int *short_steps, int *cap);

/* Rollout of a cold start that is not a trajectory.  A STATELESS solve (bmpc_solve_batch / bmpc_solve_batch_host: no dual state buffer,
 * max_iter > 0) whose x0 violates an integrator-chain row of g (q, dq, ddq, phi, dphi, ddphi) by more than 0.5 starts from the rollout of x0's
 * own jerks from the measured state, lifted variables projected.  The reference's own starts (its cold start BoundMPC.py:316-321, a shifted
 * plan :322-375) stay below the threshold; solves that carry a dual state (bmpc_solve_batch_warm, stream ticks) are never touched.
 * enabled: 1 (default) / 0 (x0 as given: what a caller that drives a receding-horizon loop through the stateless entry points wants).
 * Read at launch / capture time. */
int bmpc_set_start_rollout(bmpc_handle *h, int enabled);
int bmpc_get_start_rollout(const bmpc_handle *h);      /* 0 / 1; -1: no handle */

int bmpc_num_vars(const bmpc_handle *h);    /* 44 N */
int bmpc_num_cons(const bmpc_handle *h);    /* 43 N */
int bmpc_num_params(const bmpc_handle *h);  /* 141 + 91 S */

/* HOST pointers, lengths 44 N, 44 N, 43 N, 43 N */
int bmpc_get_bounds(const bmpc_handle *h, double *lbx, double *ubx, double *lbg, double *ubg);

/* DEVICE pointers; g, lam_g, lam_x, f, iters, status, kkt may be NULL.  hip_stream: hipStream_t to launch on (NULL = default stream).
 * Asynchronous w.r.t. the host. */
int bmpc_solve_batch(bmpc_handle *h, int B, const double *p, const double *x0, double *x, double *g, double *lam_g, double *lam_x,
                     double *f, int *iters, int *status, double *kkt, void *hip_stream);

/* Warm-started solve for receding-horizon streams: the reference's x0 (the shifted previous solution, BoundMPC.py:322-375) plus the solver's dual
 * state carried across ticks (the reference's lam_g0 / lam_x0 hand-over is commented out, :451-452).
 *   state [B][bmpc_state_len(h)] DEVICE doubles, read and updated in place: [nu (N x 57 internal inequality rows) | mu | iterations].
 *   mu <= 0 (a zeroed row): cold start, identical to bmpc_solve_batch.  Otherwise the barrier restarts at clamp(stored mu, mu_warm, mu_init),
 *   every slack at max(-h_i(x0), min(mu / nu_i, slack_push)).  The caller shifts `state` with x0 when the horizon advances (bmpc_stream_pack does).
 * max_iter > 0 overrides options.max_iter for this call (real-time iteration: status 1 is then the normal outcome); 0 keeps the option. */
int bmpc_state_len(const bmpc_handle *h);   /* 57 N + 2 */
int bmpc_solve_batch_warm(bmpc_handle *h, int B, const double *p, const double *x0, double *state, int max_iter, double *x, double *g,
                          double *lam_g, double *lam_x, double *f, int *iters, int *status, double *kkt, void *hip_stream);

/* The same step captured once into a hipGraph and replayed per tick: buffers are fixed at capture time, the caller refreshes their contents.
 * Launches of one handle (direct or replayed) share its workspace: the library orders them against each other with an event whatever streams the
 * caller uses (exception: a launch on a stream the CALLER is capturing neither waits for nor records that event).  A graph keeps the workspace and
 * the latency buffer registered at capture time and holds a reference on the handle: bmpc_destroy of a handle with live graphs closes it (the
 * graphs then refuse to launch), the last bmpc_graph_destroy frees it.  bmpc_graph_launch(g, NULL) replays on a non-blocking stream of the handle,
 * ordered by events against the legacy null stream (tests/cabi/graph_nullstream.cpp has the ROCm 7.2 fault this avoids). */
typedef struct bmpc_graph bmpc_graph;
int bmpc_graph_create(bmpc_handle *h, int B, const double *p, const double *x0, double *state, int max_iter, double *x, double *g,
                      double *lam_g, double *lam_x, double *f, int *iters, int *status, double *kkt, bmpc_graph **out);
int bmpc_graph_launch(bmpc_graph *g, void *hip_stream);
int bmpc_graph_destroy(bmpc_graph *g);

/* ---- Receding-horizon streams: the per-tick host arithmetic of BoundMPC.step() as device kernels (one wave per stream) ----
 * bmpc_stream_pack <-> step() pre-solve, BoundMPC.py:310-443 (ReferencePath window ReferencePath.py:178-238, warm start :316-333,372-375,
 *     compute_initial_rot_errors util_functions.py:11-31, projection vectors :267-304, tube quartics :219-265): (path table, stream state,
 *     robot record) -> p, x0; also shifts the solver's dual state with the plan (dual_state may be NULL).
 * bmpc_stream_post <-> step() post-solve, BoundMPC.py:460-506 (feasibility rule, fallback to the previous plan) and compute_return_data :513-611.
 *     flags bit 0 (simulate): also advance the robot record like the node's kinematic simulation (util_functions.py:152-161, bound_mpc_node.py:292-372).
 *     flags bit 1 (real-time iteration, not in the reference): the acceptance rule of BoundMPC.py:460-465 decides with the threshold of
 *     bmpc_stream_set_rt_feasibility_tol and with the violation of lbx <= x <= ubx added; a rejected iterate is not applied (the previous plan is
 *     replayed, :468-489) and the next warm start continues from it (bmpc_stream_pack_rt with xlast = the solver's x; bmpc_stream_tick does
 *     so on every launch shape; plain bmpc_stream_pack restarts from the accepted plan as the reference does; status 3 is never continued from).
 * DEVICE doubles, one row per stream, row lengths from bmpc_stream_lengths:
 *   path   [B][path_entries][path_entry]  static via-point table (layout: csrc/bmpc_stream.inl; builder: boundmpc_amd.stream.path_table)
 *   sstate [B][state]   [header 32: phi-state, rotation reference, sector, error count, weights | previous solution 44 N | Cartesian pos, vel,
 *                        acc, jerk of the previous plan 4 x 3 x N | updated flag, pad]
 *   robot  [B][robot]   q dq ddq p_lie v x_phi_d jerk = the arguments of step() (read; written when simulate)
 *   traj   [B][traj]    q dq ddq dddq (7 x N) | p v a (6 x N) | phi dphi ddphi dddphi (N) | n_valid using_previous success g_viol
 * Re-planning (BoundMPC.update, BoundMPC.py:163-217): the host writes the new path table and the state scalars and raises the `updated` flag
 * (boundmpc_amd.stream.apply_update); bmpc_stream_pack then takes the re-projection branch of step() (:335-369). */
int bmpc_stream_lengths(const bmpc_handle *h, int *path_entry, int *state, int *robot, int *traj);
int bmpc_stream_pack(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, const double *robot, double *p, double *x0,
                     double *dual_state, void *hip_stream);
/* bmpc_stream_pack with the real-time continuation: xlast (DEVICE [B][44 N] or NULL) = the iterate the solver produced last tick */
int bmpc_stream_pack_rt(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, const double *robot, double *p, double *x0,
                        double *dual_state, const double *xlast, void *hip_stream);
int bmpc_stream_post(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, double *robot, const double *x, const double *g,
                     const int *status, double *traj, int flags, void *hip_stream);
/* Threshold on the summed violation (beyond 1e-6 per row) below which bmpc_stream_post applies an iteration-capped iterate in real-time mode.
 * Default 1e-4 = the reference's rule (BoundMPC.py:462-465).  Read when a post is launched or captured. */
int bmpc_stream_set_rt_feasibility_tol(bmpc_handle *h, double tol);
/* Real-time mode (flags bit 1): a position tube row (rows 39, 40 of any stage of g: l^2 - w^2, m^2) above `cap_m2` vetoes the iterate whatever the
 * summed violation (an excess of d metres at half width w is a row of 2 w d + d^2: 1e-5 keeps every stage of an accepted plan -- also the tail a
 * stream replays after failed ticks, BoundMPC.py:468-489 -- within 0.5 mm of a 10 mm tube).  0 (default) = off: the reference's summed rule alone. */
int bmpc_stream_set_rt_position_row_cap(bmpc_handle *h, double cap_m2);
/* Real-time iteration on a HELD barrier level (what bench.py reports for BASELINE configs[4]).  bmpc_set_barrier_hold(h, 1): a solve keeps the level it
 * starts on -- clamp(level stored in the dual state, options.mu_warm, options.mu_init); a cold state starts on mu_init -- instead of walking the barrier
 * down; `tol` then never fires, the iteration cap or time budget ends every solve.  bmpc_stream_set_level_rule(h, c, lo, hi): bmpc_stream_pack (and the
 * fused ticks) write the level of a stream's next solve into its warm dual state: clamp(c (phi_max - phi), lo, hi) -- robust level `hi` far from the
 * end of the path, lower near it, where the barrier of phi <= phi_max would otherwise stall the stream short of its goal (the reference runs Ipopt's
 * adaptive barrier strategy, BoundMPC.py:130-136).  hi = 0 (default): off.  Use lo = options.mu_warm, hi = options.mu_init.  Read at launch / capture time. */
int bmpc_set_barrier_hold(bmpc_handle *h, int enabled);
int bmpc_stream_set_level_rule(bmpc_handle *h, double c, double lo, double hi);
/* Time budget of a FUSED tick in microseconds from kernel entry; 0 (default) = none.  No further iteration starts once it is used up (status 1):
 * the tick is bounded by budget + one iteration + the post-processing; results of such ticks depend on the clock.  Read at launch / capture time. */
int bmpc_stream_set_time_budget(bmpc_handle *h, double microseconds);
/* One whole tick {pack, warm-started solve with max_iter (0 = options), post} of B streams.  For B within the resident workgroups of the device
 * (bmpc_launch_info; teams: bmpc_team_info) it is ONE kernel launch, whatever N and S; otherwise the three kernels are enqueued.  In the fused
 * launch a stream that has lost its plan (error count >= N: step() returns five Nones there, BoundMPC.py:498-506) is skipped (status 3, 0
 * iterations): re-plan it (StreamBatch.update) or restart it. */
int bmpc_stream_tick(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, double *robot, double *p, double *x0,
                     double *dual_state, int max_iter, double *x, double *g, int *iters, int *status, double *kkt, double *traj, int flags,
                     void *hip_stream);
/* the same tick captured into a hipGraph (one kernel node when it fuses); launch with bmpc_graph_launch */
int bmpc_stream_graph_create(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, double *robot, double *p, double *x0,
                             double *dual_state, int max_iter, double *x, double *g, int *iters, int *status, double *kkt, double *traj,
                             int flags, bmpc_graph **out);

/* HOST pointers; one staged copy each way on a stream of the handle, then a stream synchronisation (the single-problem solver(...) call) */
int bmpc_solve_batch_host(bmpc_handle *h, int B, const double *p, const double *x0, double *x, double *g, double *lam_g, double *lam_x,
                          double *f, int *iters, int *status, double *kkt);

/* Timing of the solver kernel with HIP events on the launch stream: the {start, stop} pairs of the last `keep` launches are kept (0 = off).
 * bmpc_kernel_ms(h, back, &ms): duration of the launch `back` launches ago (0 = the last; synchronises on its stop event). */
int bmpc_set_timing(bmpc_handle *h, int keep);
int bmpc_kernel_ms(bmpc_handle *h, int back, float *ms);
int bmpc_last_kernel_ms(bmpc_handle *h, float *ms);
/* Per-solve latency: while a DEVICE buffer [>= B] is registered, every solve stores its own in-kernel duration in microseconds; NULL = off. */
int bmpc_set_latency_buffer(bmpc_handle *h, double *latency_us);

/* Teams and pairs: for short horizons (S <= 4) the library also holds kernels that put several COOPERATING WAVES on one problem (the item-parallel
 * passes of an iteration run over all their lanes, the recursions on one wave, recursion-independent work beside them on another).  Teams: 4
 * waves, N <= 10; a team owns a CU (most of its workspace lives in the CU's LDS): 256 resident on an MI355X.  Pairs: 2 waves on the one-wave
 * budget (40 KB of LDS, workspace in the global slab), N <= 11: 512 resident.  They serve the batches that leave SIMDs idle anyway: closed-loop
 * streams (BASELINE configs[4]: 256), the single solver(...) call of the drop-in (BoundMPC.py:446-453), batches up to 512.
 * bmpc_set_team_waves(h, 0) (default): teams when the batch fits into the resident teams, else pairs when it fits into the resident pairs,
 * else one wave per problem; (h, 1): always one wave; (h, 2): pairs whatever the batch; (h, 4): teams whatever the batch (an error where the
 * instantiation does not exist).  Results agree with the one-wave kernels up to the order of a few sums (same iterates unless a filter
 * decision sits on a rounding error).  bmpc_team_info: waves per problem a batch of B would get, resident teams / pairs of that kind, LDS
 * bytes of its workgroup.  The fused closed-loop ticks run on teams or on one wave per stream. */
int bmpc_set_team_waves(bmpc_handle *h, int waves);
int bmpc_team_info(const bmpc_handle *h, int B, int *waves, int *resident_teams, int *lds_bytes);

/* Work-queue order of a batch that takes several rounds of the resident waves (stateless solves, one wave per problem, B > bmpc_launch_info's grid):
 * mode 1 = longest-expected-first -- an evaluation pass ranks the problems by decreasing objective at x0 and the queue hands them out in that
 * order (a launch lasts until its last wave is done); mode 0 = natural order.  Default 1 for N > 11, 0 for shorter horizons.  Batches beyond 65 536
 * problems keep their natural order.  Results do not depend on the order. */
int bmpc_set_queue_order(bmpc_handle *h, int mode);
int bmpc_get_queue_order(const bmpc_handle *h);      /* 0 / 1; -1: no handle */

/* Second attempt: a stateless solve (no dual state buffer) that ends with status 2 is run once more from x0 on the barrier start of the short horizons
 * (mu_init 0.1, slacks pushed to 1e-2) for at most `cap` iterations; `iters` is the sum of both attempts, a second attempt that runs into its cap keeps
 * status 2.  cap = 0: off.  Default 100 for N > 11, 0 for shorter horizons.  (What Ipopt users do by hand with another mu_init; the reference has no
 * counterpart: BoundMPC.py:465-489 reports the failure.) */
int bmpc_set_second_attempt(bmpc_handle *h, int cap);
int bmpc_get_second_attempt(const bmpc_handle *h);      /* cap; -1: no handle */

/* launch geometry actually used: resident workgroups (one wave each), LDS bytes per workgroup, scratch bytes per workgroup */
int bmpc_launch_info(const bmpc_handle *h, int *grid, int *lds_bytes, long long *scratch_bytes);

#ifdef __cplusplus
}
#endif
#endif
