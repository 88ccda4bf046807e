/*
 * boundmpc_hip.h -- C ABI of the MI355X-native batched BoundMPC OCP solver.
 *
 * Drop-in boundary.  In the reference (Thieso/BoundMPC) the per-step optimisation is one
 * CasADi `nlpsol('solver','ipopt',...)` Function object created at
 *   bound_mpc/bound_mpc/BoundMPC/casadi_ocp_formulation.py:389
 * and invoked only at
 *   bound_mpc/bound_mpc/BoundMPC/BoundMPC.py:446-453   sol = solver(x0=, lbx=, ubx=, lbg=, ubg=, p=)
 *   bound_mpc/bound_mpc/BoundMPC/BoundMPC.py:456        stats = solver.stats()
 * The reference defines no C ABI for it (it is a Python object backed by CasADi/Ipopt/MUMPS);
 * the entry points below carry exactly the information that call carries, batched:
 *
 *   bmpc_create        <-> setup_optimization_problem(N, 7, nr_segs, dt, ...)  (casadi_ocp_formulation.py:9-11)
 *                          + the Ipopt option dict (BoundMPC.py:120-148: tol, max_iter)
 *   bmpc_get_bounds    <-> the lbu, ubu, lbg, ubg lists returned there (casadi_ocp_formulation.py:384-391)
 *   bmpc_solve_batch   <-> solver(x0=, lbx=, ubx=, lbg=, ubg=, p=) -> {'x','f','g','lam_x','lam_g'}  (BoundMPC.py:446-453)
 *                          and solver.stats() -> iter_count / success (BoundMPC.py:456-457), one record per problem
 *   bmpc_destroy       <-> garbage collection of the Function object
 *
 * All arithmetic is fp64.  Layouts (row-major, one problem per row):
 *   p      [B][n_p]   n_p = 141 + 91*S  parameter vector, layout of casadi_ocp_formulation.py:361-376
 *   x0, x  [B][44*N]  stage variables z_k = [u(7) u_phi | q dq ddq | p(6) | v(6) | phi dphi ddphi] (:90-153)
 *   g      [B][43*N]  constraints in the reference's order/form (:272-349)
 *   lam_g  [B][43*N]  multipliers of g (sign: L = f + lam_g.g), lam_x [B][44*N] multipliers of lbx <= x <= ubx
 *   f, kkt [B], iters [B], status [B]  (0 converged, 1 max_iter reached, 2 locally infeasible -- N <= 11: the restoration phase
 *                                        (bmpc_set_restoration) converged to a point whose constraint violation is not zero, or did not reach a
 *                                        feasible point within its budget; longer horizons: the primal infeasibility has not halved over
 *                                        `stall_window` iterations after up to three barrier restarts from the stalled iterate --, 3 numerical failure)
 * lbx/ubx/lbg/ubg are structural constants of the formulation (robot limits, 36 equalities
 * + 7 inequalities per stage) and do not cross the ABI per call; bmpc_get_bounds returns them.
 *
 * Ownership: the caller owns every buffer; the library owns the handle and its device
 * scratch.  Errors are integer return codes (no exceptions cross the ABI); per-problem
 * failure is data (status[]), as in the reference (BoundMPC.py:465-489).
 * Thread-safety: one in-flight bmpc_solve_batch per handle.
 * Memory: the handle's device workspace (one 173 KB slab per resident wave at N=10, 519 KB at N=30, 693 KB at N=40) is allocated by the first solve
 * or graph capture, for min(B, resident waves) waves, and grows when a later call brings a larger batch (after a host wait for
 * the handle's own last launch; other streams and handles of the process are not stalled); while captured graphs of the handle exist it cannot grow -- capture for the largest batch first.
 */
#ifndef BOUNDMPC_HIP_H
#define BOUNDMPC_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct bmpc_handle bmpc_handle;

typedef struct {
    double tol;         /* KKT tolerance, Ipopt-style scaled error (reference: 'tol': 10e-6, BoundMPC.py:121); default 1e-8 */
    int max_iter;       /* reference: 500 (BoundMPC.py:122) */
    double mu_init;     /* initial barrier parameter: 0.1 (Ipopt's default) for N <= 11, 3.0 for longer horizons, whose cold start is far
                           from the solution and violates the tube rows (N=30 tight: 35 instead of 46 iterations on average); see bmpc_default_options_for */
    double mu_min_fac;  /* final barrier = tol * mu_min_fac */
    double slack_push;  /* minimum initial slack of an inequality row: 1e-2 (Ipopt bound_push) for N <= 11, 0.1 for longer horizons */
    int exact_hessian;  /* 1: exact Lagrangian Hessian (reference uses CasADi's exact Hessian); 0: Gauss-Newton */
    int verbose;
    double mu_warm;     /* warm start (bmpc_solve_batch_warm): the barrier restarts at clamp(stored mu, mu_warm, mu_init); default 1e-2 (round 2: closed loops converge in 9.5 instead of 11.2 iterations with it; 1e-4 jams the iterate against moved constraints) */
    int stall_window;   /* the main phase counts as stalled when the primal infeasibility has not halved over this many iterations (checked every
                           stall_window/2 iterations); 0 = never; default 40 for N <= 11, 20 for longer horizons.  A stall starts the restoration
                           phase (bmpc_set_restoration; N <= 11) or a barrier restart / status 2 (longer horizons, or restoration off).  Warm-started receding-horizon streams converge in ~10-12 iterations: 16 is the
                           recommended value there (a stream that is losing its plan runs every solve to this test and a batched tick lasts as
                           long as its slowest stream: 256 closed loops, tick p50 7.1 -> 3.0 ms with the same streams keeping their plan) */
    double bound_margin;/* joint position / velocity limits tightened by this much (rad, rad/s) INSIDE the solver; default 0 = the reference's limits
                           (RobotModel.py:20-39).  For real-time closed loops solved to a loose tolerance or an iteration / time budget: a plan whose
                           bound rows are met to 1e-3 only then still respects the true limits the acceptance rule checks.  (appended in round 4) */
} bmpc_options;

enum { BMPC_OK = 0, BMPC_ERR_ARG = 1, BMPC_ERR_HIP = 2, BMPC_ERR_NOGPU = 4 };

/* ABI note: bmpc_options grew by `bound_margin` in round 4 (appended).  A caller compiled against an older header passes a shorter struct:
 * compare sizeof(bmpc_options) with bmpc_options_size() before bmpc_create.  Later additions are handle setters (bmpc_set_restoration), not fields. */
int bmpc_options_size(void);
/* 16 hex digits: hash of the source text and compiler flags the library was built from (boundmpc_amd/build.py source_hash) */
const char *bmpc_build_hash(void);
int bmpc_default_options(bmpc_options *o);                 /* the N <= 11 defaults */
int bmpc_default_options_for(int N, bmpc_options *o);      /* defaults for horizon N; what bmpc_create(…, NULL, …) uses */
const char *bmpc_error_string(int code);

/* N horizon (1..40, also for the closed-loop stream entry points bmpc_stream_*), S path segments in the window (2..6; with 5 or 6 the iterate
 * lives in the workspace instead of LDS at every horizon, as it does for N > 11), dt sampling time */
int bmpc_create(int N, int S, double dt, const bmpc_options *opts, bmpc_handle **out);
int bmpc_destroy(bmpc_handle *h);

/* Restoration phase -- what stands where Ipopt's filter line search falls back to its restoration phase (BoundMPC.py:135 'line_search_method':
 * 'filter'; Waechter & Biegler 2006, 3.3).  When the main interior-point phase is jammed (`short_steps` consecutive steps shorter than 10 % with the
 * primal infeasibility still open; default 6) or its stall test fires, the solve switches to the feasibility problem
 *     min  rho * sum_i e_i   s.t.  dynamics,  h_i(x) - e_i <= 0,  e_i >= 0      (rho = 1000: the l1 norm of the violation, objective weights zero),
 * solved by the same interior-point iteration on the same stage-wise Riccati recursion (the elastic variables are eliminated row by row), and
 *   - returns to the main phase from the first strictly feasible iterate (slacks and multipliers re-centred there), or
 *   - ends the solve with status 2 when it converges to a point with non-zero violation (a local minimiser of the violation: Ipopt's
 *     "converged to a point of local infeasibility"), when `cap` iterations (default 40) did not produce a feasible point (Ipopt: "restoration
 *     failed"), or at the fourth call within one solve.
 * enabled: 0 never (status 2 / 3 as in round 4); 1 full -- on a jam, a stall or a NUMERICAL BREAKDOWN of the main phase (dual residual beyond 1e12,
 * which without the phase ends the solve as status 3); 2 after a numerical breakdown only.  Default 1 for N <= 11, 2 for longer horizons.  Long
 * horizons (N > 11) keep their three barrier restarts from a stalled iterate; in mode 1 the phase follows them as the LAST RESORT (never on a jam: a
 * tight 30-stage solve takes short steps for its first 15 iterations anyway): it rescues 14 of the 26 problems of BASELINE configs[3] that end as
 * status 2 (99.68 -> 99.84 % converged), but the slowest problem of that launch then takes 314 instead of 170 iterations and the launch 37 % longer;
 * instead of the restarts it would be worse (oracle/bmpc_oracle.c).  Mode 2 costs such a batch nothing and rescues far-off starts: 64 loose N = 20
 * problems started with noise 0.3 on every variable: 64 converge (mode 0: 50, 14 numerical breakdowns).
 * A negative argument keeps the current value.  Read at launch / capture time.  Fixture g13b (every first failing tick of 256 closed loops): the 28
 * locally infeasible problems end as status 2 after 22-54 iterations, 8 of the 10 feasible ones converge in 69-121 (tests/test_gpu_parity.py).
 * An iterate that is far off its own dynamics when the phase starts (an equality residual above 1e-2: a bad warm start) is first rolled out from
 * the measured state with its own jerks: of 256 feasible problems started with noise 0.3 on every variable 254 converge (without the phase: 69).
 * Batch solves: the batch kernels are compiled without the phase (carrying it costs their hot path 6 %); a jammed problem is continued by a second
 * kernel started right behind (it returns at once when nothing jammed), with identical numbers.  Fused closed-loop ticks carry it in the kernel;
 * time-budgeted real-time ticks (bmpc_stream_set_time_budget) run without it.  Warm-started closed loops: cap = 24 keeps the same streams alive as 40
 * and a stream that is losing its plan then costs a tick about what the stall test did (bench_stream.py --resto-cap). */
int bmpc_set_restoration(bmpc_handle *h, int enabled, int short_steps, int cap);
int bmpc_get_restoration(const bmpc_handle *h, int *enabled, int *short_steps, int *cap);

/* Rollout of a cold start that is not a trajectory (round 5; oracle/bmpc_oracle.c solve_one).  The reference hands Ipopt its own cold start (the
 * measured state repeated, BoundMPC.py:316-321) or the shifted previous plan (:322-375); a caller of this library may hand over anything.  A STATELESS
 * solve (bmpc_solve_batch / bmpc_solve_batch_host: no dual state buffer) whose x0 violates the integrator chains (the q, dq, ddq, phi, dphi, ddphi rows of g) by more than 0.5 is started from the rollout
 * of x0's own jerks from the measured state, the lifted variables (pos, i-omega, v) projected -- a dynamically consistent trajectory with the same
 * controls.  128 feasible N = 10 problems from the reference's cold start + noise 0.1 ... 2.0 on every variable, from all zeros, from uniform(-1, 1)
 * noise: all converge, in 13-21 iterations on average (through the restoration phase alone: 94-100 % in 30-54; with neither: none).  The reference's
 * own starts are not touched (a cold start is a trajectory; a shifted plan is off by h dq at its last node -- 99 % of 12 267 closed-loop ticks below
 * 0.51 -- and comes with a dual state), nor is any
 * solve that carries a dual state buffer (bmpc_solve_batch_warm, the stream ticks -- warm or not): on closed loops the same step costs plans.  A call with max_iter = 0 (the evaluation of f and g AT x0) is never rolled out.
 * enabled: 1 (default) / 0 (x0 as given).  Read at launch / capture time. */
int bmpc_set_start_rollout(bmpc_handle *h, int enabled);
int bmpc_get_start_rollout(const bmpc_handle *h);      /* 0 / 1; -1: no handle */

int bmpc_num_vars(const bmpc_handle *h);    /* 44 N */
int bmpc_num_cons(const bmpc_handle *h);    /* 43 N */
int bmpc_num_params(const bmpc_handle *h);  /* 141 + 91 S */

/* HOST pointers, lengths 44N, 44N, 43N, 43N */
int bmpc_get_bounds(const bmpc_handle *h, double *lbx, double *ubx, double *lbg, double *ubg);

/* DEVICE pointers (e.g. torch-ROCm tensor.data_ptr()); g, lam_g, lam_x, f, iters, status, kkt may be NULL.
 * hip_stream: hipStream_t to launch on (NULL = default stream).  Asynchronous w.r.t. the host. */
int bmpc_solve_batch(bmpc_handle *h, int B, const double *p, const double *x0, double *x, double *g, double *lam_g, double *lam_x,
                     double *f, int *iters, int *status, double *kkt, void *hip_stream);

/* Warm-started solve for receding-horizon streams.  The reference warm-starts Ipopt from the previous tick's shifted solution
 * (x0: BoundMPC.py:322-375, 'warm_start_init_point': 'yes' :134; its lam_g0/lam_x0 hand-over is commented out, :451-452).  This
 * call takes the same x0 and additionally carries the solver's dual state across ticks:
 *   state [B][bmpc_state_len(h)] DEVICE doubles, read and updated in place: [nu (N x 57 internal inequality rows) | mu | iterations].
 *   A row with mu <= 0 (e.g. a zeroed buffer) is a cold start, identical to bmpc_solve_batch.  Otherwise the barrier restarts at
 *   clamp(stored mu, options.mu_warm, options.mu_init), every slack at max(-h_i(x0), min(mu/nu_i, slack_push)), nu_i = mu/t_i.
 *   The caller shifts `state` the way it shifts x0 (rows of node k+1 -> node k) when the horizon advances by one stage.
 * max_iter > 0 overrides options.max_iter for this call (real-time iteration: a fixed number of Newton steps per tick; status 1
 * is then the normal outcome and the state carries the unfinished iterate's multipliers to the next tick); 0 keeps the option. */
int bmpc_state_len(const bmpc_handle *h);   /* 57 N + 2 */
int bmpc_solve_batch_warm(bmpc_handle *h, int B, const double *p, const double *x0, double *state, int max_iter, double *x, double *g,
                          double *lam_g, double *lam_x, double *f, int *iters, int *status, double *kkt, void *hip_stream);

/* The same step captured once into a hipGraph (work-queue reset + solver kernel) and replayed per tick with hipGraphLaunch: the
 * buffers are fixed at capture time, the caller refreshes their contents (p, x0, state) between launches.  state may be NULL
 * (cold starts).  Launches of one handle (direct or replayed) share its workspace and work queue: the library orders them against each
 * other with an event whatever streams the caller uses, so they never overlap (exception: a launch on a stream the CALLER is capturing
 * -- a graph of the caller's own around these entry points -- neither waits for nor records that event; ordering such a graph against the
 * handle's other work is the caller's).  A graph keeps the handle's workspace and the latency
 * buffer registered at capture time (re-capture after bmpc_set_latency_buffer) and holds a reference on the handle: bmpc_destroy of
 * a handle with live graphs closes it (those graphs then refuse to launch, BMPC_ERR_ARG) and the last bmpc_graph_destroy frees it.
 * bmpc_graph_launch(g, NULL): the replay runs on a non-blocking stream of the handle, ordered by events after the work the legacy
 * null stream holds so far and before its later work (a graph replayed ON the null stream and followed by unsynchronised null-stream
 * launches ended in a GPU memory fault on ROCm 7.2; tests/cabi/graph_nullstream.cpp). */
typedef struct bmpc_graph bmpc_graph;
int bmpc_graph_create(bmpc_handle *h, int B, const double *p, const double *x0, double *state, int max_iter, double *x, double *g,
                      double *lam_g, double *lam_x, double *f, int *iters, int *status, double *kkt, bmpc_graph **out);
int bmpc_graph_launch(bmpc_graph *g, void *hip_stream);
int bmpc_graph_destroy(bmpc_graph *g);

/* ---- Receding-horizon streams: the per-tick host arithmetic of BoundMPC.step() as device kernels (one stream per thread) ----
 * bmpc_stream_pack  <-> BoundMPC.step() pre-solve, BoundMPC.py:310-443 (ReferencePath window ReferencePath.py:178-238, warm start
 *                       :316-333,372-375, compute_initial_rot_errors util_functions.py:11-31, projection vectors :267-304, tube
 *                       quartics :219-265): (path table, stream state, robot record) -> p [B][n_p], x0 [B][44N]; also shifts the
 *                       solver's dual state with the plan (dual_state may be NULL).
 * bmpc_stream_post  <-> BoundMPC.step() post-solve, BoundMPC.py:460-506 (feasibility rule, fallback to the previous plan) and
 *                       compute_return_data :513-611 (re-integration, Cartesian trajectory, advance of phi / rotation reference);
 *                       flags bit 0 (simulate): additionally advance the robot record like the node's kinematic simulation
 *                       (util_functions.py:152-161, bound_mpc_node.py:292-372); bit 1 (real-time iteration, not in the
 *                       reference: a fixed small number of solver iterations per tick, status 1 is the normal outcome): the
 *                       reference's acceptance rule BoundMPC.py:460-465 decides with the threshold of
 *                       bmpc_stream_set_rt_feasibility_tol in place of 1e-4, and with the violation of the variable bounds lbx <= x <= ubx
 *                       of the plan (jerk, joint position and velocity limits; an iteration-capped iterate need not satisfy them)
 *                       added to the violation of g; an iterate that fails it is not applied, the
 *                       previous plan is replayed (BoundMPC.py:468-489).  The next tick's warm start then CONTINUES FROM THE REJECTED
 *                       ITERATE (shifted like an accepted plan; its multipliers are in the dual state anyway) instead of the last
 *                       accepted plan -- the iterations spent on it are kept; the reference would restart from the accepted plan.
 *                       bmpc_stream_tick / bmpc_stream_graph_create do this on every launch shape (fused or three kernels); callers of
 *                       the separate entry points get it by packing with bmpc_stream_pack_rt(..., xlast = the solver's x buffer, ...);
 *                       plain bmpc_stream_pack (xlast = NULL) restarts from the accepted plan as the reference does.  A numerical
 *                       failure (status 3) is never continued from.
 * All buffers are DEVICE doubles, one row per stream, row lengths from bmpc_stream_lengths:
 *   path   [B][path_entries][path_entry]  static via-point table (built on the host once per path; layout in
 *                                          boundmpc_amd/csrc/bmpc_stream.inl, builder boundmpc_amd.stream.path_table)
 *   sstate [B][state]   phi-state, rotation reference, sector, error count, weights, previous solution
 *   robot  [B][robot]   q dq ddq p_lie v x_phi_d jerk  = the arguments of step()            (read; written when simulate)
 *   traj   [B][traj]    q dq ddq dddq (7 x N) | p v a (6 x N) | phi dphi ddphi dddphi (N) | n_valid using_previous success g_viol
 *   sstate row = [header 32 | previous solution 44 N | Cartesian pos, vel, acc, jerk of the previous plan 4 x 3 x N | updated flag, pad].
 * Re-planning (BoundMPC.update, BoundMPC.py:163-217): the host writes the new path table and the state scalars update() sets and
 * raises the `updated` flag (boundmpc_amd.stream.apply_update); bmpc_stream_pack then takes the re-projection branch of step()
 * (BoundMPC.py:335-369) for good, as the reference does, from the arrays bmpc_stream_post keeps. */
int bmpc_stream_lengths(const bmpc_handle *h, int *path_entry, int *state, int *robot, int *traj);
int bmpc_stream_pack(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, const double *robot, double *p, double *x0,
                     double *dual_state, void *hip_stream);
/* bmpc_stream_pack with the real-time continuation: xlast (DEVICE [B][44 N] or NULL) = the iterate the solver produced last tick */
int bmpc_stream_pack_rt(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, const double *robot, double *p, double *x0,
                        double *dual_state, const double *xlast, void *hip_stream);
int bmpc_stream_post(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, double *robot, const double *x, const double *g,
                     const int *status, double *traj, int flags, void *hip_stream);
/* Threshold on the summed violation of g (beyond 1e-6 per row) below which bmpc_stream_post applies an iteration-capped iterate in
 * real-time mode (flags bit 1).  Default 1e-4 = the reference's rule (BoundMPC.py:462-465).  Read when a post is launched or captured. */
int bmpc_stream_set_rt_feasibility_tol(bmpc_handle *h, double tol);
/* Time budget of a FUSED tick (bmpc_stream_tick / bmpc_stream_graph_create when they fuse), in microseconds from kernel entry; 0 (default) = none.
 * With a budget the solver of a tick starts no further iteration once the budget is used up (status 1, like the iteration cap): every stream
 * gets the iterations that fit -- a stream whose Riccati sweep repeats or whose line search takes extra trials gets fewer -- and the tick is
 * bounded by budget + one iteration + the post-processing.  The iteration count of a stream then depends on the clock: results of such ticks
 * are not reproducible bit for bit.  Use with flags bit 1 (the acceptance rule decides what is applied).  Read at launch / capture time. */
int bmpc_stream_set_time_budget(bmpc_handle *h, double microseconds);
/* One whole tick {pack, warm-started solve with max_iter (0 = options), post} of B streams.  For B within the resident workgroups of the device
 * (bmpc_launch_info: grid; teams: bmpc_team_info) it is ONE kernel launch, whatever N and S: the wave that owns a stream packs its problem, solves it and
 * post-processes the result (no work queue, no launch boundary between the steps); otherwise the three kernels are enqueued.
 * Arguments as bmpc_stream_pack / bmpc_solve_batch_warm / bmpc_stream_post.  In the fused launch a stream that has lost its plan (error count
 * >= N: BoundMPC.step() returns five Nones there, BoundMPC.py:498-506) is skipped (status 3, 0 iterations): re-plan it (StreamBatch.update)
 * or restart it. */
int bmpc_stream_tick(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, double *robot, double *p, double *x0,
                     double *dual_state, int max_iter, double *x, double *g, int *iters, int *status, double *kkt, double *traj, int flags,
                     void *hip_stream);
/* the same tick captured into a hipGraph (one kernel node when it fuses); launch with bmpc_graph_launch */
int bmpc_stream_graph_create(bmpc_handle *h, int B, const double *path, int path_entries, double *sstate, double *robot, double *p, double *x0,
                             double *dual_state, int max_iter, double *x, double *g, int *iters, int *status, double *kkt, double *traj,
                             int flags, bmpc_graph **out);

/* HOST pointers; copies in/out and synchronises (convenience for the single-problem solver(...) call) */
int bmpc_solve_batch_host(bmpc_handle *h, int B, const double *p, const double *x0, double *x, double *g, double *lam_g, double *lam_x,
                          double *f, int *iters, int *status, double *kkt);

/* Timing of the solver kernel with HIP events on the launch stream.  bmpc_set_timing(h, keep): the {start, stop} event pairs of the
 * last `keep` launches are kept (0 = off; 1 = the last launch only), so a caller can enqueue a run of launches back to back and read
 * the durations afterwards.  bmpc_kernel_ms(h, back, &ms): duration of the launch `back` launches ago (0 = the last; synchronises on
 * its stop event); bmpc_last_kernel_ms = bmpc_kernel_ms(h, 0, .). */
int bmpc_set_timing(bmpc_handle *h, int keep);
int bmpc_kernel_ms(bmpc_handle *h, int back, float *ms);
int bmpc_last_kernel_ms(bmpc_handle *h, float *ms);
/* Per-solve latency: while a DEVICE buffer [>= B] is registered, every solve stores its own in-kernel duration in microseconds
 * (wall-clock counter read when a wavefront takes the problem from the queue and when it has written the outputs); NULL = off. */
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

/* launch geometry actually used: resident workgroups (one wave each), LDS bytes per workgroup, scratch bytes per workgroup */
int bmpc_launch_info(const bmpc_handle *h, int *grid, int *lds_bytes, long long *scratch_bytes);

#ifdef __cplusplus
}
#endif
#endif
