#!/usr/bin/env python3
"""BASELINE.json configs[4] (SURVEY 8d "Config 5"): warm-started receding-horizon stream, hipGraph-captured step,
batch=256 parallel closed-loop trajectories.

Not the driver's bench line (that is bench.py / configs[1]).  256 independent closed loops (generator of configs[1], seed 3:
random q0, own experiment1-pattern path each), same OCP (N=10, h=0.1 s, S=4).  Like the reference node, simulated time
advances by h per tick while the wall-clock tick rate is what is measured.  Each tick, entirely on the device and replayed
from ONE captured hipGraph: pack (path window, warm-start shift, parameter vector) -> solver step -> post (feasibility rule,
re-integration, phi / rotation-reference advance, kinematic plant step).  Timed with HIP events around the graph launch.

Modes: converged (solve to tol every tick, cold duals = what the reference does with Ipopt), warm (dual state carried),
rt-tolX-capK (cold duals, loose tolerance X, at most K iterations per tick, a capped iterate is applied as it is), rtw-* (the same with
the dual state carried; rtgn-*: and the Gauss-Newton Hessian),
rti-K (K Newton steps per tick from the carried primal-dual state).  For every mode the closed-loop result is compared
with the converged loop: RMS joint deviation over all ticks, and path progress phi after the last tick.  Independently of how far
the loops have drifted apart, every tick's applied plan is also compared with the converged solution OF THE SAME PROBLEM (same p,
same warm start; solved outside the timed region): joint RMS distance over the horizon and relative objective excess -- the
real-time-iteration error proper."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--ticks", type=int, default=131)
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--mu-warm", type=float, default=1e-2)
    ap.add_argument("--max-iter", type=int, default=100, help="iteration cap of the converged modes (reference: 500)")
    ap.add_argument("--rt-mu-warm", type=float, default=3e-2, help="barrier restart level of the warm real-time modes (rtw-*)")
    ap.add_argument("--rt-tol", type=float, default=1e-3, help="KKT tolerance of the real-time modes (rt-*: cold duals, hard iteration cap)")
    ap.add_argument("--rt-feas-tol", type=float, default=1e-2,
                    help="threshold of the reference's acceptance rule (summed violation of g, BoundMPC.py:462-465: 1e-4) that decides whether an "
                         "iteration-capped iterate is applied in the real-time modes; an iterate that fails it is not applied, the previous plan is replayed")
    ap.add_argument("--rt-bound-margin", type=float, default=2e-3, help="joint limits tightened inside the solver of the time-budgeted modes (rad, rad/s)")
    ap.add_argument("--rt-row-cap", type=float, default=1e-5, help="fixed-level modes: a position tube row (l^2 - w^2, any stage, m^2) above this vetoes the iterate (bmpc_stream_set_rt_position_row_cap; 0 = the summed rule alone)")
    ap.add_argument("--rtfix-mu", default="0.1,0.05", help="barrier levels of the fixed-barrier time-budgeted modes (rtfix-*)")
    ap.add_argument("--rtfix-budgets", default="625", help="their time budgets in microseconds")
    ap.add_argument("--cfb", default="24,14,1.0", help="converged-fallback mode: iteration cap of the solve to tolerance, Newton steps and barrier level of the fallback")
    ap.add_argument("--only", default="", help="comma-separated substrings: run only the modes whose name contains one of them (the converged loop always runs: it is the reference of the deviations)")
    ap.add_argument("--stall-window", type=int, default=16,
                    help="stall window of the converged / warm loops (0: the handle's default, 40 for N <= 11).  A stream that is losing its plan (locally infeasible "
                         "ticks, DESIGN.md 3) runs every solve to this test, and a tick lasts as long as its slowest stream: with 40 the converged loops take "
                         "p50 7.1 / p99 11.8 ms per tick, with 16 3.0 / 7.1 ms -- the same 92.6 %% of the streams keep their plan, the same ticks are applied "
                         "(12: 2.7 / 5.4 ms, but two more streams lose their plan); healthy warm-started ticks converge in ~10-12 iterations")
    ap.add_argument("--resto-cap", type=int, default=24, help="iterations one restoration phase may take in the converged / warm loops (0: the handle default, 40; closed loops: 24 keeps the same streams alive and a dying stream then costs a tick about what the stall test did, DESIGN.md 5b)")
    ap.add_argument("--no-restoration", action="store_true", help="converged / warm loops without the restoration phase (round 4's behaviour: stall test only)")
    ap.add_argument("--unsafe-too", action="store_true", help="also run the real-time modes with every capped iterate applied (the round-2 behaviour), for comparison")
    args = ap.parse_args()
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload, stream as bstream
    B, T = args.batch, args.ticks
    q0s = workload.random_q0(B, seed=3)
    mpcs, recs = [], []
    for q0 in q0s:
        m, p0fk = workload.make_mpc(q0)
        mpcs.append(m)
        recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
    recs = np.stack(recs)
    solver = BatchedOCPSolver(10, 4, 0.1, tol=args.tol, mu_warm=args.mu_warm, max_iter=args.max_iter, stall_window=args.stall_window or None)
    solver.set_timing(True)
    if args.no_restoration:
        solver.set_restoration(False)
    elif args.resto_cap:
        solver.set_restoration(cap=args.resto_cap)
    # real-time modes: loose tolerance + hard iteration cap per tick, COLD duals (the barrier restarts centred every tick: carrying a
    # small mu jams the iterate against the constraints that change with the shifted horizon)
    rt = {}
    for cap in (8, 7, 6, 5):
        rt[cap] = BatchedOCPSolver(10, 4, 0.1, tol=args.rt_tol, max_iter=cap)
        rt[cap].set_timing(True)
    # warm real-time modes: the same, but the dual state is carried (shifted with the plan on the device) and the barrier restarts at
    # a MODERATE level (mu_warm 3e-2): active rows keep their multipliers, nothing is jammed; mean 4.4 iterations per tick
    rtw = {}
    for cap in (7, 6, 5, 4, 3):
        rtw[cap] = BatchedOCPSolver(10, 4, 0.1, tol=args.rt_tol, max_iter=cap, mu_warm=args.rt_mu_warm)
        rtw[cap].set_timing(True)
    # warm real-time modes with the Gauss-Newton Hessian (the classical choice of real-time iteration schemes): positive semidefinite by
    # construction, so no tick ever repeats a Riccati sweep after a failed factorisation
    rtgn = {}
    for cap in (6, 5, 4, 3):
        rtgn[cap] = BatchedOCPSolver(10, 4, 0.1, tol=args.rt_tol, max_iter=cap, mu_warm=args.rt_mu_warm, exact_hessian=False)
        rtgn[cap].set_timing(True)
    # time-budgeted real-time modes (round 4): no fixed iteration count -- the fused tick starts no further iteration once the budget (from
    # kernel entry) is used up; loose tolerance, dual state carried
    rtb = {}
    for us in (600, 700, 800):
        for gn in (False, True):
            rtb[(us, gn)] = BatchedOCPSolver(10, 4, 0.1, tol=args.rt_tol, max_iter=30, mu_warm=args.rt_mu_warm, exact_hessian=not gn, bound_margin=args.rt_bound_margin)
            rtb[(us, gn)].set_timing(True)
    # time-budgeted modes on a FIXED barrier level (round 5; the classical real-time iteration of an interior-point method): mu_init = mu_warm = final
    # level = MU, so a tick spends its few iterations as Newton steps on ONE barrier problem whose solution the previous tick left nearby, instead of
    # restarting the barrier at mu_warm and re-converging through its levels; tol never fires (the complementarity stays at MU): every tick uses its
    # budget and the reference's acceptance rule (threshold rt_feas_tol) decides.  Larger MU: plans further from the tube / limit rows, more robust.
    rtf = {}
    for us in [int(v) for v in args.rtfix_budgets.split(",") if v]:
        for MU in [float(v) for v in args.rtfix_mu.split(",") if v]:
            rtf[(us, MU)] = BatchedOCPSolver(10, 4, 0.1, tol=args.rt_tol, max_iter=30, fixed_barrier=MU, bound_margin=args.rt_bound_margin)
            rtf[(us, MU)].set_timing(True); rtf[(us, MU)].set_rt_position_row_cap(args.rt_row_cap)
    # converged loops with a barrier-level fallback (round 5; StreamBatch.tick_with_fallback): every tick is solved to tolerance with at most 24 iterations (no
    # restoration phase); the streams that did not converge are solved again from the same warm start on the fixed barrier level 1 (14 Newton steps) and
    # the reference's acceptance rule at the reference's threshold 1e-4 decides
    cfb_main = BatchedOCPSolver(10, 4, 0.1, tol=args.tol, mu_warm=args.mu_warm, max_iter=args.max_iter, stall_window=args.stall_window or None)
    cfb_main.set_restoration(False); cfb_main.set_timing(True)
    cfb_cap, cfb_k, cfb_mu = int(args.cfb.split(",")[0]), int(args.cfb.split(",")[1]), float(args.cfb.split(",")[2])
    cfb_level = BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=cfb_k, fixed_barrier=cfb_mu)
    cfb_level.set_restoration(False)
    rtfc = {}      # the fixed-level modes with an ITERATION cap instead of a time budget: every stream takes exactly that many Newton steps -- no clock in the result
    for MU in [float(v) for v in args.rtfix_mu.split(",") if v][:1]:
        rtfc[MU] = BatchedOCPSolver(10, 4, 0.1, tol=args.rt_tol, max_iter=30, fixed_barrier=MU, bound_margin=args.rt_bound_margin)
        rtfc[MU].set_timing(True); rtfc[MU].set_rt_position_row_cap(args.rt_row_cap)
    rtfa = BatchedOCPSolver(10, 4, 0.1, tol=args.rt_tol, max_iter=30, fixed_barrier="auto", bound_margin=args.rt_bound_margin)      # the level that sets itself per stream (round 6)
    rtfa.set_timing(True); rtfa.set_rt_position_row_cap(args.rt_row_cap)
    evaluate = BatchedOCPSolver(10, 4, 0.1, max_iter=0)       # f, g at a given point (no iteration)
    reference = BatchedOCPSolver(10, 4, 0.1, tol=args.tol, max_iter=args.max_iter, start_rollout=False)    # every tick's problem solved to tolerance, untimed; x0 taken as given, like every stream solve
    res, ref_q = [], None
    FT = args.rt_feas_tol
    modes = [("converged", solver, 0, False, None), ("warm", solver, 0, True, None)] \
        + [(f"rtw-tol{args.rt_tol:g}-cap{c}-feas{FT:g}", rtw[c], 0, True, FT) for c in (7, 5, 4)] \
        + [(f"rtgn-tol{args.rt_tol:g}-cap{c}-feas{FT:g}", rtgn[c], 0, True, FT) for c in (6, 5, 4, 3)] \
        + [(f"rtgn-tol{args.rt_tol:g}-cap4-feas1e-4(reference rule)", rtgn[4], 0, True, 1e-4)] \
        + [(f"rt-tol{args.rt_tol:g}-cap{c}-feas{FT:g}", rt[c], 0, False, FT) for c in (8, 6)] \
        + [("rti-3-feas%g" % FT, solver, 3, True, FT)] \
        + [("warm-continue-feas1e-4 (converged solves; a stalled tick's iterate is the next warm start; reference acceptance rule + variable bounds)", solver, 0, True, 1e-4)] \
        + [(f"rtb{'gn' if gn else 'w'}-tol{args.rt_tol:g}-budget{us}us-feas{FT:g}", rtb[(us, gn)], 0, True, FT) for (us, gn) in sorted(rtb)]
    modes += [(f"rtfix-mu{MU:g}-tol{args.rt_tol:g}-budget{us}us-feas{FT:g}", rtf[(us, MU)], 0, True, FT) for (us, MU) in sorted(rtf)]
    modes += [(f"rtfixcap-mu{MU:g}-cap{c}-feas{FT:g} (fixed barrier level, exactly {c} Newton steps per tick, no clock: reproducible bit for bit)", rtfc[MU], c, True, FT) for MU in sorted(rtfc) for c in (5, 6, 7)]
    modes += [(f"rtfixcap-muauto-cap{c}-feas{FT:g} (barrier level held inside the tick, set per stream: clamp(0.02 (phi_max - phi), 0.01, 0.1); exactly {c} Newton steps per tick; the mode bench.py reports for configs[4])", rtfa, c, True, FT) for c in (5, 6)]
    modes += [(f"converged-fallback-cap{cfb_cap}-level{cfb_mu:g}-k{cfb_k}-feas1e-4 (solved to tolerance within {cfb_cap} iterations, else {cfb_k} steps on the barrier level {cfb_mu:g}; the reference's rule at 1e-4)", cfb_main, cfb_cap, True, 1e-4)]
    if args.unsafe_too:
        modes += [(f"UNSAFE rtgn-tol{args.rt_tol:g}-cap{c} (every capped iterate applied)", rtgn[c], 0, True, 1e30) for c in (4, 3)]
    from boundmpc_amd.robot_model import RobotModel
    qlim = np.array(RobotModel().q_lim_upper)
    # Everything below runs on an explicit HIP stream.  On the legacy null stream, a hipGraph replay followed by further kernel launches
    # without a host synchronisation in between ended in a GPU memory fault on ROCm 7.2 (80 ticks into the warm mode, reproducibly); the
    # same sequence with direct launches instead of the graph, or on any explicit stream, is clean.
    torch.cuda.set_stream(torch.cuda.Stream())
    only = [k for k in args.only.split(",") if k]
    budget_of = {id(rtb[k]): k[0] for k in rtb}
    budget_of.update({id(rtf[k]): k[0] for k in rtf})
    for mode, slv, cap, warm, feas in modes:
        if only and mode != "converged" and not any(k in mode for k in only):
            continue
        budget = budget_of.get(id(slv), 0)
        capped = feas is not None
        if capped:
            slv.set_rt_feasibility_tol(feas)          # read when the tick graph is captured
        sb = bstream.StreamBatch(slv, mpcs)
        sb.set_robot(recs)
        ms, its, Q, ok, wall, tick_dq, tick_df, alive, gviol, tube_p, tube_r, skipped, maxit, row_p, row_r = [], [], [], [], [], [], [], [], [], [], [], [], [], [], []
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for t in range(T):
            if os.environ.get("BENCH_STREAM_TRACE"):
                print("trace", mode, t, file=sys.stderr, flush=True)
            # the first tick of every stream is its cold start from rest: solved to tolerance in all modes (not timed)
            if t == 0:
                if budget:
                    slv.set_time_budget_us(0)
                sb.tick(max_iter=100, warm_dual=True, simulate=True)
                if not warm:
                    sb.dual.zero_()
                if budget:
                    slv.set_time_budget_us(budget)      # read when the tick graph is captured (next tick)
            else:
                ev0.record()
                if mode.startswith("converged-fallback"):
                    sb.tick_with_fallback(cfb_level, max_iter=cap, simulate=True)      # (three launches + a host read of the failed count: not a graph)
                else:
                    sb.tick_graph(max_iter=cap, warm_dual=warm, simulate=True, accept_capped=capped)
                ev1.record(); ev1.synchronize()
                wall.append(ev0.elapsed_time(ev1))         # whole tick {pack, queue reset, solve, post, plant}: HIP events around the graph launch
            ms.append(slv.last_kernel_ms())
            if t > 0 and mode != "converged":
                # this tick's problem solved to 1e-8 from the same warm start (untimed): how far is the applied plan from its minimiser?
                star = reference.solve_batch(sb.p, sb.x0, want=("f", "status"))
                at_x = evaluate.solve_batch(sb.p, sb.x, want=("f",))
                torch.cuda.synchronize()
                f_rt = at_x["f"]
                good = star["status"] == 0
                d = (sb.x - star["x"]).reshape(B, 10, 44)[:, :, 8:15]
                tick_dq.append(torch.sqrt((d * d).mean(dim=(1, 2)))[good].cpu().numpy())
                tick_df.append(((f_rt - star["f"]) / star["f"].abs().clamp_min(1.0))[good].cpu().numpy())
            its.append(float(sb.iters.double().mean().item()))
            maxit.append(int(sb.iters.max().item()))
            if t > 0:
                # BoundMPC's contract: is the MEASURED state of this tick inside its tubes?  (the packed p holds the measured pose, phi and the tube
                # quartics: stream.tube_excess_of_state evaluates the five tube rows of casadi_ocp_formulation.py:316-349 at node 0; streams that
                # have lost their plan -- skipped by the fused tick, their p is stale -- are left out and counted)
                has_plan = (sb.state[:, bstream.SS["ERRCNT"]] < 10).cpu().numpy()
                ex_p, ex_r = bstream.tube_excess_of_state(sb.p.cpu().numpy())
                tube_p.append(np.where(has_plan[:, None], ex_p, -np.inf).max(axis=1)); tube_r.append(np.where(has_plan[:, None], ex_r, -np.inf).max(axis=1))
                skipped.append(int((~has_plan).sum()))
                # ... and as the reference's own logging shows it (BoundMPC.py:614-752: err_data against the bounds, per stage of the APPLIED plan, in
                # the tick's own linearisation of the orientation error): the tube rows of the plan's first stage -- the state the plant reaches next --
                # in the reference's form l^2 - w^2 (rows 38..42 of g), over the ticks whose plan was applied
                g0 = sb.g.reshape(B, 10, 43)[:, 0, 38:43].cpu().numpy(); app = (sb.traj[:, -2] > 0.5).cpu().numpy()
                row_p.append(np.where(app, g0[:, 1:3].max(axis=1), -np.inf)); row_r.append(np.where(app, g0[:, [0, 3, 4]].max(axis=1), -np.inf))
            Q.append(sb.robot[:, :7].clone())
            ok.append(float((sb.traj[:, -2] > 0.5).double().mean().item()))
            alive.append(float((sb.state[:, bstream.SS["VALID"]] > 0.5).double().mean().item()))
            if t > 0:
                gviol.append(sb.traj[:, -1].cpu().numpy())
        Q = torch.stack(Q).cpu().numpy()
        phi = sb.state[:, bstream.SS["PHI"]].cpu().numpy()
        if ref_q is None:
            ref_q = Q
        ms = np.array(ms[1:]); its = np.array(its[1:]); wall = np.array(wall)
        dev = Q - ref_q
        per_stream = np.sqrt(np.mean(dev ** 2, axis=(0, 2)))
        pt = {}
        if tick_dq:
            dq_all, df_all = np.concatenate(tick_dq), np.concatenate(tick_df)
            pt = {"per_tick_joint_rms_vs_own_minimiser_rad": {"median": float(np.median(dq_all)), "p90": float(np.percentile(dq_all, 90)), "p99": float(np.percentile(dq_all, 99))},
                  "per_tick_relative_objective_excess": {"median": float(np.median(df_all)), "p90": float(np.percentile(df_all, 90)), "p99": float(np.percentile(df_all, 99))}}
        tube_p, tube_r = np.array(tube_p), np.array(tube_r)      # [ticks][streams]: largest excess over the rows, -inf where the stream had no plan
        n_samples = int(np.isfinite(tube_p).sum())
        tube = {"plant_samples": n_samples, "tolerance": 1e-6,
                "fraction_outside_the_position_tube": float((tube_p > 1e-6).sum() / max(n_samples, 1)), "largest_position_excess_m": float(max(tube_p.max(), 0.0)),
                "fraction_outside_the_orientation_tube": float((tube_r > 1e-6).sum() / max(n_samples, 1)), "largest_orientation_excess_rad": float(max(tube_r.max(), 0.0)),
                "streams_ever_outside_a_tube": int(((tube_p > 1e-6) | (tube_r > 1e-6)).any(axis=0).sum()),
                "note": "rows of casadi_ocp_formulation.py:316-349 at the MEASURED state (node 0 of the next packed problem), linear form |l| - |w|.  Position rows: exact.  "
                        "Orientation rows: the exact zyx split of the measured orientation error, which the NLP only constrains through its per-tick linearisation -- "
                        "a loop that solves every tick to 1e-8 shows the same excess (the `converged` row), so it measures the reference's formulation, not the solver"}
        row_p, row_r = np.array(row_p), np.array(row_r); n_app = int(np.isfinite(row_p).sum())
        tube["applied_plans_first_stage_rows_reference_form"] = {
            "applied_plans": n_app, "fraction_with_a_position_row_above_1e-6": float((row_p > 1e-6).sum() / max(n_app, 1)), "largest_position_row_m2": float(max(row_p.max(), 0.0)),
            "fraction_with_an_orientation_row_above_1e-6": float((row_r > 1e-6).sum() / max(n_app, 1)), "largest_orientation_row_rad2": float(max(row_r.max(), 0.0)),
            "note": "l^2 - w^2 <= 0 (g rows 38..42 of stage 0) of every applied plan: what the reference's err_data / bounds logging shows for the state the plant reaches next"}
        res.append({**pt, "mode": mode, "tick_ms_p50": float(np.percentile(wall, 50)), "tick_ms_p99": float(np.percentile(wall, 99)),
                    "tube_compliance_of_the_measured_states": tube, "streams_skipped_per_tick_mean": float(np.mean(skipped)), "streams_skipped_at_the_end": int(skipped[-1]),
                    "slowest_stream_iterations_per_tick": {"p50": float(np.percentile(maxit[1:], 50)), "p99": float(np.percentile(maxit[1:], 99)), "max": int(max(maxit[1:]))},
                    "solver_kernel_ms_p50": float(np.percentile(ms, 50)), "solver_kernel_ms_p99": float(np.percentile(ms, 99)),
                    "ticks_per_s": float(1e3 / wall.mean()), "solves_per_s": float(B * 1e3 / wall.mean()), "mean_iters": float(its.mean()),
                    "applied_tick_fraction": float(np.mean(ok[1:])), "streams_with_a_plan_at_the_end": float(alive[-1]), "streams_with_a_plan_min_over_ticks": float(np.min(alive)),
                    "acceptance_threshold_g_viol": feas, "g_viol_of_the_solver_iterates": {"median": float(np.median(np.concatenate(gviol))), "p90": float(np.percentile(np.concatenate(gviol), 90)), "p99": float(np.percentile(np.concatenate(gviol), 99))},
                    "joint_limit_violations_of_the_plant_state": int((np.abs(Q) > qlim + 1e-9).sum()), "largest_joint_limit_excess_rad": float(max((np.abs(Q) - qlim).max(), 0.0)),
                    "rms_joint_dev_vs_converged_loop_rad": float(np.sqrt(np.mean(dev ** 2))),
                    "median_stream_rms_dev_rad": float(np.median(per_stream)), "streams_within_1e-2_rad_rms": float((per_stream <= 1e-2).mean()),
                    "max_joint_dev_rad": float(np.abs(dev).max()), "mean_phi_after_last_tick": float(phi.mean())})
        sb.close()
    # the 1 kHz target, as the round-3 verdict states it: a safeguarded mode whose tick is p50 <= 1.0 ms / p99 <= 1.3 ms, >= 75 % of the streams hold a
    # plan after the last tick, no plant sample outside the joint limits -- computed from this run's modes, not asserted
    met = [r["mode"] for r in res if r["tick_ms_p50"] <= 1.0 and r["tick_ms_p99"] <= 1.3 and r["streams_with_a_plan_at_the_end"] >= 0.75
           and r["joint_limit_violations_of_the_plant_state"] == 0]
    verdict = ("met by " + ", ".join(met)) if met else "not met by any mode of this run"
    # the STRICT reading (round-4 verdict, item 2b): tick p99 <= 1.0 ms with >= 85 % of the streams holding a plan after the last tick, no plant sample outside
    # the joint limits; the fixed-barrier modes (rtfix-*) are the ones built for it
    strict = [r["mode"] for r in res if r["tick_ms_p99"] <= 1.0 and r["streams_with_a_plan_at_the_end"] >= 0.85 and r["joint_limit_violations_of_the_plant_state"] == 0]
    verdict_strict = ("met by " + ", ".join(strict)) if strict else "not met by any mode of this run"
    # ONE JSON line a driver can parse like bench.py's: value = ticks/s of the mode reported: of those that meet the strict criteria (else the round-3 ones) the one that keeps most plans (0 when none does)
    best = max((r for r in res if r["mode"] in (strict or met)), key=lambda r: (r["streams_with_a_plan_at_the_end"], r["ticks_per_s"]), default=None)      # strict modes first; among them the one that keeps most plans
    print(json.dumps({"metric": "closed-loop ticks/s of 256 parallel receding-horizon streams (BASELINE configs[4]: 1 kHz target)", "value": best["ticks_per_s"] if best else 0.0,
                      "unit": "ticks/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f64", "data": "synthetic", "vs_baseline": None,
                      "mode_reported": best["mode"] if best else None, "tick_ms_p50": best["tick_ms_p50"] if best else None, "tick_ms_p99": best["tick_ms_p99"] if best else None,
                      "config": {"workload": "BASELINE.json configs[4]: warm-started receding-horizon stream at 1 kHz, one captured launch per tick, batch=256 parallel trajectories "
                                             "(random q0 seed 3, own experiment1-pattern path each, 7-DOF, N=10, S=4, dt=0.1; pack + solve + post + plant on the device)",
                                 "batch": B, "ticks": T - 1},
                      "batch": B, "ticks": T - 1,
                      "tol": args.tol, "mu_warm": args.mu_warm, "budget_ms": 1.0,
                      "verdict_on_the_strict_1_kHz_target": verdict_strict + " (criteria: tick p99 <= 1.0 ms, >= 85 % of the streams with a plan after the last tick, no plant sample outside the joint limits)",
                      "verdict_on_the_1_kHz_target": verdict + " (criteria: tick p50 <= 1.0 ms, p99 <= 1.3 ms, >= 75 % of the streams with a plan after the last tick, no plant sample outside the joint limits)",
                      "workload": "256 closed loops, random q0 (seed 3), own experiment1-pattern path, N=10, h=0.1 s; pack+solve+post+plant on device",
                      "results": res}))


if __name__ == "__main__":
    main()
