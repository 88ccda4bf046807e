"""GPU: the whole tick {pack, solve, post} of B closed-loop streams on the device (bmpc_stream_* of the C ABI), direct
launches and the captured hipGraph, against the committed closed-loop fixtures of the reference's host code."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _mpcs():
    from boundmpc_amd import workload
    from boundmpc_amd.bound_mpc import BoundMPC
    from tests.test_stream import _Oracle
    out = []
    for which in (1, 2):
        d6 = np.load(os.path.join(G, f"g6_pack_exp{which}_tick0.npz"))
        mk = lambda k: [np.array(v) for v in d6[k]]
        out.append((BoundMPC(mk("p_via"), mk("r_via"), [mk("p_lower"), mk("p_upper")], [mk("r_lower"), mk("r_upper")], mk("bp1_in"), mk("br1_in"),
                             list(d6["s_in"]), list(d6["e_p_min_in"]), list(d6["e_r_min_in"]), list(d6["e_p_max_in"]), list(d6["e_r_max_in"]),
                             p0=d6["p0fk"].copy(), params=workload.Params(weights=d6["weights_f64"]), solver=_Oracle()), d6))
    return out


def _robot0(mpc, d6):
    from boundmpc_amd import stream as bstream
    from boundmpc_amd.robot_model import RobotModel
    q = d6["q0"].copy()
    return bstream.robot_record(q, np.zeros(7), np.zeros(7), RobotModel().forward_kinematics(q, np.zeros(7))[0], np.zeros(6),
                                np.array([mpc.phi_max[0], 0, 0]), np.zeros(7))


@pytest.mark.parametrize("graph", [False, True])
def test_device_closed_loop_retraces_fixtures(graph):
    import torch
    from boundmpc_amd import BatchedOCPSolver, stream as bstream
    ms = _mpcs()
    solver = BatchedOCPSolver(10, 4, 0.1)
    sb = bstream.StreamBatch(solver, [m for m, _ in ms] * 2)          # 4 streams: exp1, exp2, exp1, exp2
    sb.set_robot(np.stack([_robot0(m, d) for m, d in ms] * 2))
    d7 = [np.load(os.path.join(G, f"g7_closedloop_exp{w}.npz")) for w in (1, 2)]
    for t in range(59):
        (sb.tick_graph if graph else sb.tick)(simulate=True)
        torch.cuda.synchronize()
        st = sb.state.cpu().numpy(); tr = sb.traj.cpu().numpy(); rb = sb.robot.cpu().numpy()
        for b in range(4):
            f = d7[b % 2]
            td, fl = bstream.unpack_traj(tr[b], 10)
            assert fl["success"] and fl["n_valid"] == 10
            np.testing.assert_allclose(td["q"], f["traj_q"][t], atol=2e-5, err_msg=f"tick {t} stream {b}")
            np.testing.assert_allclose(td["p"], f["traj_p"][t], atol=2e-5)
            assert abs(st[b, bstream.SS["PHI"]] - f["phi_current"][t]) < 1e-6 and int(st[b, 0]) == int(f["sector"][t])
            if t + 1 < f["q"].shape[0]:
                np.testing.assert_allclose(rb[b, :7], f["q"][t + 1], atol=2e-6)
        assert np.abs(sb.iters.cpu().numpy()[:2] - np.array([d7[0]["iters"][t], d7[1]["iters"][t]])).max() <= 1
    assert np.array_equal(st[0], st[2]) and np.array_equal(st[1], st[3])      # identical streams stay bit-identical
    sb.close(); solver.close()


def test_device_pack_and_post_equal_cpu_build_of_the_same_text():
    """bmpc_stream_pack / bmpc_stream_post on the GPU against tests/emu's g++ build of csrc/bmpc_stream.inl, open loop over
    recorded ticks (device libm vs host libm: round-off only)."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, stream as bstream
    from oracle import c_oracle
    from tests.emu import emu
    (mpc, d6), _ = _mpcs()
    d7 = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))
    solver = BatchedOCPSolver(10, 4, 0.1)
    sb = bstream.StreamBatch(solver, [mpc])
    T, M = bstream.path_table(mpc.ref_path)
    ss = bstream.initial_state(mpc, 10); ss[bstream.SS["NENT"]] = M
    xphid = np.array([mpc.phi_max[0], 0, 0])
    for t in range(0, 155, 7):
        rb = bstream.robot_record(d7["q"][t], d7["dq"][t], d7["ddq"][t], d7["p_lie"][t], d7["v"][t], xphid, d7["jerk"][t])
        if t:                                                    # state of the stream just before tick t
            ss[bstream.SS["HASPREV"]] = 1; ss[bstream.SS["PREV"]:bstream.SS["PREV"] + 440] = d7["x"][t - 1]
            ss[bstream.SS["PHI"]], ss[bstream.SS["DPHI"]], ss[bstream.SS["DDPHI"]], ss[bstream.SS["DDDPHI"]] = (
                d7["phi_current"][t - 1], d7["dphi_current"][t - 1], d7["ddphi_current"][t - 1], d7["dddphi_current"][t - 1])
            ss[7:10], ss[10:13] = d7["pr_ref"][t - 1], d7["iw_ref"][t - 1]
            ss[0] = d7["sector"][t - 1]
        sb.state.copy_(torch.tensor(ss[None])); sb.set_robot(rb[None])
        sb.pack(); torch.cuda.synchronize()
        ss_c = ss.copy(); p_c, x0_c = emu.stream_pack(10, 4, T, ss_c, rb)
        np.testing.assert_allclose(sb.p.cpu().numpy()[0], p_c, atol=1e-12, rtol=1e-12)
        np.testing.assert_allclose(sb.x0.cpu().numpy()[0], x0_c, atol=1e-14)
        g = c_oracle.eval_fg(d7["p"][t], d7["x"][t], 10, 4, 0.1)[1]
        sb.x.copy_(torch.tensor(d7["x"][t][None])); sb.g.copy_(torch.tensor(g[None])); sb.status.zero_()
        sb.post(simulate=True); torch.cuda.synchronize()
        rb_c = rb.copy(); tr_c = emu.stream_post(10, 4, 0.1, T, ss_c, rb_c, d7["x"][t], g, 0, simulate=True)
        np.testing.assert_allclose(sb.traj.cpu().numpy()[0], tr_c, atol=1e-11, rtol=1e-11)
        np.testing.assert_allclose(sb.state.cpu().numpy()[0], ss_c, atol=1e-11, rtol=1e-11)
        np.testing.assert_allclose(sb.robot.cpu().numpy()[0], rb_c, atol=1e-12)
    sb.close(); solver.close()


def test_fallback_replay_on_the_gpu_matches_host_mirror():
    """SURVEY 8 row f3 on the device: solver failures forced on the GPU (a handle capped at 2 iterations ends with status 1 and a
    grossly infeasible iterate) at ticks 3, 4 and 9 -- the device-side state machine (error count, replay of the previous plan from
    index error_count, shortened trajectories; BoundMPC.py:465-506,514-524) against the host mirror BoundMPC.step() fed by the same
    GPU solves through the nlpsol shim."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, NlpSolverShim, stream as bstream, workload
    from boundmpc_amd.bound_mpc import integrate_joint
    from boundmpc_amd.robot_model import RobotModel
    fails = (3, 4, 9)
    good, bad = BatchedOCPSolver(10, 4, 0.1), BatchedOCPSolver(10, 4, 0.1, max_iter=2)

    class Switch:          # nlpsol-shaped: the capped handle at the failing ticks
        def __init__(self):
            self.calls, self.s = 0, [NlpSolverShim(good), NlpSolverShim(bad)]

        def generate_dependencies(self, *a, **k):
            pass

        def __call__(self, **kw):
            self.cur = self.s[1 if self.calls in fails else 0]
            self.calls += 1
            return self.cur(**kw)

        def stats(self):
            return self.cur.stats()
    q0 = workload.random_q0(3, seed=5)[2]
    mpc, p0fk = workload.make_mpc(q0, solver=Switch())
    ref, _ = workload.make_mpc(q0)
    sb = bstream.StreamBatch(good, [ref])
    rm = RobotModel()
    q, dq, ddq, jerk, v = q0.copy(), np.zeros(7), np.zeros(7), np.zeros(7), np.zeros(6)
    x_phi_d = np.array([mpc.phi_max[0], 0, 0])
    sb.set_robot(bstream.robot_record(q, dq, ddq, p0fk, v, x_phi_d, jerk)[None])
    out = dict(x=sb.x, g=sb.g, iters=sb.iters, status=sb.status, kkt=sb.kkt)
    seen = []
    for t in range(14):
        p_lie = rm.forward_kinematics(q, dq)[0]
        traj, _, _, _, _ = mpc.step(q, dq, ddq, p_lie, v, x_phi_d, jerk)
        sb.pack()
        (bad if t in fails else good).solve_batch(sb.p, sb.x0, out=out, want=("g", "iters", "status", "kkt"))
        sb.post(simulate=True)
        torch.cuda.synchronize()
        st = sb.state.cpu().numpy()[0]
        td, fl = bstream.unpack_traj(sb.traj.cpu().numpy()[0], 10)
        assert int(sb.status.cpu().numpy()[0]) == (1 if t in fails else 0)
        assert int(st[bstream.SS["ERRCNT"]]) == mpc.error_count
        assert fl["using_previous"] == (t in fails) and fl["n_valid"] == 10 - mpc.error_count
        seen.append(mpc.error_count)
        for k in ("q", "dq", "ddq", "dddq", "p", "v", "a", "phi", "dphi", "ddphi", "dddphi"):
            np.testing.assert_allclose(td[k], traj[k], atol=2e-5 if k in ("dddq", "dddphi") else 2e-6, err_msg=f"tick {t} {k}")
        jm = np.concatenate((jerk[:, None], traj["dddq"][:, :2]), axis=1)
        q, dq, ddq, p_lie, v = integrate_joint(rm, jm, q, dq, ddq, mpc.dt)[:5]
        jerk = traj["dddq"][:, 0].copy()
        np.testing.assert_allclose(sb.robot.cpu().numpy()[0, :7], q, atol=1e-7)
    assert seen[3] == 1 and seen[4] == 2 and seen[5] == 0 and seen[9] == 1
    sb.close(); good.close(); bad.close()


def test_256_streams_real_time_mode_is_safeguarded_and_tracks_the_converged_loops():
    """BASELINE configs[4] at batch 256: closed loops ticked from ONE captured launch per tick (fused pack + solve + post).  Real-time mode
    (Gauss-Newton Hessian, 4 iterations per tick, dual state carried, barrier restart at 3e-2): an iteration-capped iterate is applied
    only if it passes the reference's acceptance rule (summed violation of g, BoundMPC.py:462-465) with the threshold 1e-2; otherwise the
    previous plan is replayed and the next tick continues from the rejected iterate.  Over the first 40 ticks of the paths (the later
    segments are not met: DESIGN.md 5b): every applied iterate really passed the rule, almost all streams keep a plan, the loops stay
    close to the loops solved to 1e-8 every tick, the capped tick is bounded; the unsafeguarded round-2 behaviour is gone by default."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, stream as bstream, workload
    from boundmpc_amd.robot_model import RobotModel
    B, T, FEAS = 256, 40, 1e-2
    q0s = workload.random_q0(B, seed=3)
    mpcs, recs = [], []
    for q0 in q0s:
        m, p0fk = workload.make_mpc(q0)
        mpcs.append(m)
        recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
    recs = np.stack(recs)
    st = torch.cuda.Stream()
    runs = {}
    with torch.cuda.stream(st):
        for name, slv, capped in (("converged", BatchedOCPSolver(10, 4, 0.1, max_iter=100), False),
                                  ("rtgn", BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=4, mu_warm=3e-2, exact_hessian=False), True)):
            if capped:
                slv.set_rt_feasibility_tol(FEAS)
            sb = bstream.StreamBatch(slv, mpcs)
            sb.set_robot(recs)
            Q, ms, applied, viol_applied, alive = [], [], [], [], []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for t in range(T):
                if t == 0:      # cold start from rest: to tolerance in both runs
                    sb.tick(max_iter=100, warm_dual=True, simulate=True)
                    if not capped:
                        sb.dual.zero_()
                else:
                    e0.record(); sb.tick_graph(simulate=True, warm_dual=capped, accept_capped=capped); e1.record(); e1.synchronize()
                    ms.append(e0.elapsed_time(e1))
                    ok = (sb.traj[:, -2] > 0.5)
                    applied.append(float(ok.double().mean().item()))
                    if capped:
                        bad = ok & (sb.status != 0) & (sb.traj[:, -1] >= FEAS)
                        viol_applied.append(int(bad.sum().item()))
                    alive.append(float((sb.state[:, bstream.SS["VALID"]] > 0.5).double().mean().item()))
                Q.append(sb.robot[:, :7].clone())
            runs[name] = (torch.stack(Q).cpu().numpy(), np.array(ms), np.array(applied), np.array(viol_applied), np.array(alive),
                          sb.state[:, bstream.SS["PHI"]].cpu().numpy())
            sb.close(); slv.close()
    Qc, msc, apc, _, alc, phic = runs["converged"]
    Qr, msr, apr, bad, alr, phir = runs["rtgn"]
    per_stream = np.sqrt(np.mean((Qr - Qc) ** 2, axis=(0, 2)))
    qlim = np.array(RobotModel().q_lim_upper)
    print(f"\n256 streams x {T - 1} ticks: converged tick p50 {np.percentile(msc, 50):.2f} / p99 {np.percentile(msc, 99):.2f} ms; real-time (GN, 4 iterations, rule at "
          f"{FEAS:g}) p50 {np.percentile(msr, 50):.2f} / p99 {np.percentile(msr, 99):.2f} ms; applied {apr.mean():.3f}, streams with a plan {alr.min():.3f}; "
          f"per-stream RMS deviation median {np.median(per_stream):.2e}, p90 {np.percentile(per_stream, 90):.2e}, max {per_stream.max():.2e} rad")
    assert apc.mean() >= 0.98 and alc.min() == 1.0      # a few of the 256 x 39 converged ticks run into the 100-iteration cap (fallback plan)
    assert bad.sum() == 0                                # no applied iterate violates the acceptance rule
    assert apr.mean() >= 0.75 and alr.min() >= 0.9       # measured 0.85 / 0.95
    assert np.median(per_stream) <= 2.5e-2               # measured 1.1e-2
    assert np.percentile(msr, 99) < np.percentile(msc, 50)      # the capped tick is bounded: its p99 is below the converged p50
    assert (np.abs(Qc) <= qlim + 1e-9).all()             # the converged loops never leave the joint limits
    # the real-time loops: the acceptance rule counts the plan's variable bounds (round 4), so the plant -- which follows accepted plans only --
    # stays inside the joint limits
    over = np.abs(Qr) - qlim
    print(f"real-time loops: plant samples beyond the joint limits {int((over > 1e-9).sum())} (largest excess {max(float(over.max()), 0.0):.2e} rad); "
          f"largest joint deviation from the converged loops {np.abs(Qr - Qc).max():.3f} rad; path progress phi (mean over streams) {phir.mean():.3f} vs {phic.mean():.3f}")
    assert over.max() <= 1e-9                            # (round 4: a plan with any variable outside its bounds is never applied)
    # no runaway loop (round 2's unsafeguarded mode: joint deviations of 43-131 rad): every loop stays inside the joint limits (above), nine of ten
    # within 0.1 rad RMS of their converged loop; single streams follow another branch of the redundant arm (measured: largest deviation 4.3 rad)
    assert np.percentile(per_stream, 90) <= 0.1          # measured 4.1e-2
    assert phir.mean() >= 0.8 * phic.mean()              # the loops make progress along their paths


def test_replanning_on_the_device_matches_reference_update_g11():
    """BoundMPC.update() for a stream on the GPU (StreamBatch.update: new table + state scalars from the host, the re-projected warm
    starts and the Cartesian derivatives of the previous plan on the device) against the REFERENCE's own update()/step() (fixture G11)."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, stream as bstream
    from oracle import c_oracle
    (mpc, d6), _ = _mpcs()
    d7 = np.load(os.path.join(G, "g7_closedloop_exp1.npz"))
    d = np.load(os.path.join(G, "g11_update.npz"))
    T_UPD = int(d["t_update"])
    solver = BatchedOCPSolver(10, 4, 0.1)
    sb = bstream.StreamBatch(solver, [mpc])
    xphid = np.array([mpc.phi_max[0], 0, 0])

    def feed(q, dq, ddq, p_lie, v, jerk, x, p_fix, status):
        sb.set_robot(bstream.robot_record(q, dq, ddq, p_lie, v, xphid, jerk)[None])
        sb.pack(); torch.cuda.synchronize()
        p, x0 = sb.p.cpu().numpy()[0].copy(), sb.x0.cpu().numpy()[0].copy()
        g = c_oracle.eval_fg(p_fix, x, 10, 4, 0.1)[1]
        sb.x.copy_(torch.tensor(x[None])); sb.g.copy_(torch.tensor(g[None])); sb.status.fill_(int(status))
        sb.post(simulate=False); torch.cuda.synchronize()
        return p, x0
    for t in range(T_UPD):
        feed(d7["q"][t], d7["dq"][t], d7["ddq"][t], d7["p_lie"][t], d7["v"][t], d7["jerk"][t], d7["x"][t], d7["p"][t], 0)
    L = lambda k: [np.array(v) for v in d["upd_" + k]]
    phi_max = sb.update(0, L("p_via"), L("r_via"), [L("p_lower"), L("p_upper")], [L("r_lower"), L("r_upper")], L("bp1"), L("br1"), list(d["upd_s"]),
                        list(d["upd_e_p_min"]), list(d["upd_e_r_min"]), list(d["upd_e_p_max"]), list(d["upd_e_r_max"]),
                        d["upd_p"], d["upd_v"], d["upd_a"], d["upd_jerk"], d["upd_p"], d["weights"])
    assert abs(phi_max - float(d["after_update_phi_max"])) < 1e-13
    xphid = np.array([phi_max, 0, 0])
    mask = d6["p_defined_mask"]
    for i in range(len(d["x"])):
        p, x0 = feed(d["q"][i], d["dq"][i], d["ddq"][i], d["p_lie"][i], d["v"][i], d["jerk"][i], d["x"][i], d["p"][i], int(d["status"][i]))
        np.testing.assert_allclose(p[mask], d["p"][i][mask], atol=5e-11, rtol=1e-11, err_msg=f"p, tick {i} after update")
        np.testing.assert_allclose(x0, d["x0"][i], atol=1e-11, err_msg=f"x0, tick {i} after update")
        st = sb.state.cpu().numpy()[0]
        td, _ = bstream.unpack_traj(sb.traj.cpu().numpy()[0], 10)
        np.testing.assert_allclose(td["q"], d["traj_q"][i], atol=1e-11)
        assert abs(st[bstream.SS["PHI"]] - d["phi_current"][i]) < 1e-11 and int(st[0]) == int(d["sector"][i])
        np.testing.assert_allclose(st[10:13], d["iw_ref"][i], atol=1e-11)
    sb.close(); solver.close()


@pytest.mark.parametrize("N,S", [(5, 2), (8, 3), (20, 4), (6, 5), (12, 6)])
def test_device_tick_other_horizons_and_windows_g12(N, S):
    """The whole device tick {pack, solve, post} for other (n, nr_segs) than the experiments': closed loop against the reference's own
    step() driven with the CPU oracle (fixture G12) -- parameter layout 141 + 91 S, warm start 44 N, N <= 11 and N > 11 kernels."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, stream as bstream, workload
    from boundmpc_amd.bound_mpc import BoundMPC
    from tests.test_stream import _Oracle
    d6 = np.load(os.path.join(G, "g6_pack_exp2_tick0.npz"))
    d = np.load(os.path.join(G, "g12_pack_other_sizes.npz"))
    k = f"n{N}s{S}_"
    dt, mask = float(d[k + "dt"]), d[k + "mask"]
    mk = lambda key: [np.array(v) for v in d6[key]]
    mpc = BoundMPC(mk("p_via"), mk("r_via"), [mk("p_lower"), mk("p_upper")], [mk("r_lower"), mk("r_upper")], mk("bp1_in"), mk("br1_in"),
                   list(d6["s_in"]), list(d6["e_p_min_in"]), list(d6["e_r_min_in"]), list(d6["e_p_max_in"]), list(d6["e_r_max_in"]),
                   p0=d6["p0fk"].copy(), params=workload.Params(n=N, dt=dt, nr_segs=S, weights=d["weights"]), solver=_Oracle())
    solver = BatchedOCPSolver(N, S, dt)
    sb = bstream.StreamBatch(solver, [mpc])
    sb.set_robot(bstream.robot_record(d[k + "q"][0], d[k + "dq"][0], d[k + "ddq"][0], d[k + "p_lie"][0], d[k + "v"][0],
                                      np.array([mpc.phi_max[0], 0, 0]), d[k + "jerk"][0])[None])
    for i in range(len(d[k + "x"])):
        sb.tick(simulate=True); torch.cuda.synchronize()
        # closed loops of two solvers: round-off of each solve (tol 1e-8) feeds the next tick, the jerk entries feel it most
        np.testing.assert_allclose(sb.p.cpu().numpy()[0][mask], d[k + "p"][i][mask], atol=2e-4, err_msg=f"tick {i}")
        np.testing.assert_allclose(sb.x0.cpu().numpy()[0], d[k + "x0"][i], atol=2e-4)
        td, fl = bstream.unpack_traj(sb.traj.cpu().numpy()[0], N)
        assert fl["success"] and fl["n_valid"] == N
        np.testing.assert_allclose(td["q"], d[k + "traj_q"][i], atol=2e-6)
        assert abs(sb.state.cpu().numpy()[0][bstream.SS["PHI"]] - d[k + "phi_current"][i]) < 1e-6
    sb.close(); solver.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [2, 1])
def test_fused_tick_runs_the_restoration_mode_of_the_handle(mode):
    """One handle, one setting, whatever the launch shape: the fused tick (bmpc_stream_tick: one kernel) and the three-kernel tick (pack, batch solve
    with the restoration kernel behind it, post) of the SAME stream states give the same statuses and iteration counts, tick by tick, through the
    stretch of the path where ticks jam (phi ~ 5).  Mode 2 (numerical breakdowns only): until round 6 the fused tick ran mode 1 there -- a jammed
    tick then enters the restoration phase (status 2 after 20-50 iterations) where the three-kernel tick stalls (status 2 after 60-80)."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, stream as bstream, workload
    B, T = 48, 130
    q0s = workload.random_q0(256, seed=3)[:B]
    mpcs, recs = [], []
    for q0 in q0s:
        m, p0fk = workload.make_mpc(q0)
        mpcs.append(m)
        recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
    sa, sb_ = BatchedOCPSolver(10, 4, 0.1, max_iter=100), BatchedOCPSolver(10, 4, 0.1, max_iter=100)
    sa.set_restoration(mode); sb_.set_restoration(mode)
    a, b = bstream.StreamBatch(sa, mpcs), bstream.StreamBatch(sb_, mpcs)
    a.set_robot(np.stack(recs)); b.set_robot(np.stack(recs))
    nd, ns, worst, n2 = 0, 0, 0, 0
    for t in range(T):
        for k in ("state", "robot", "dual", "x", "traj"):      # the unfused loop starts every tick from the fused loop's state
            getattr(b, k).copy_(getattr(a, k))
        a.tick(max_iter=100, warm_dual=True, simulate=True, fused=True)
        b.tick(max_iter=100, warm_dual=True, simulate=True, fused=False)
        torch.cuda.synchronize()
        live = (a.status != 3) | (a.iters > 0)      # (streams that lost their plan are skipped by the fused tick)
        sta, stb, ia, ib = a.status[live].cpu().numpy(), b.status[live].cpu().numpy(), a.iters[live].cpu().numpy(), b.iters[live].cpu().numpy()
        ns += int((sta != stb).sum()); d = np.abs(ia - ib); nd += int((d > 4).sum()); worst = max(worst, int(d.max()) if d.size else 0); n2 += int((sta == 2).sum())
    a.close(); b.close(); sa.close(); sb_.close()
    assert n2 > 0, "no tick of the run ended as status 2: the test does not reach the jams"
    assert ns == 0 and nd == 0, (ns, nd, worst, n2)


@pytest.mark.gpu
def test_long_closed_loops_through_the_hard_part_of_the_path():
    """64 random streams over 130 ticks (the benchmark stops at 60): later segments are harder, some streams stall and run their error
    count past N (the reference's `step()` returns None from there on, BoundMPC.py:498-506).  Everything stays finite, the stalled
    streams stop where they are, the others keep their progress; graph replays on an explicit stream."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, stream as bstream, workload
    B, T = 64, 130
    q0s = workload.random_q0(256, seed=3)[:B]
    mpcs, recs = [], []
    for q0 in q0s:
        m, p0fk = workload.make_mpc(q0)
        mpcs.append(m)
        recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
    slv = BatchedOCPSolver(10, 4, 0.1, max_iter=100)
    sb = bstream.StreamBatch(slv, mpcs)
    sb.set_robot(np.stack(recs))
    side = torch.cuda.Stream()
    phis, ecs, oks = [], [], []
    with torch.cuda.stream(side):
        for t in range(T):
            if t == 0:
                sb.tick(max_iter=100, warm_dual=True, simulate=True)
            else:
                sb.tick_graph(warm_dual=True, simulate=True)
            side.synchronize()
            phis.append(sb.state[:, bstream.SS["PHI"]].cpu().numpy().copy())
            ecs.append(sb.state[:, bstream.SS["ERRCNT"]].cpu().numpy().copy())
            oks.append((sb.traj[:, -2] > 0.5).cpu().numpy().copy())
            assert bool(torch.isfinite(sb.state).all()) and bool(torch.isfinite(sb.robot).all()) and bool(torch.isfinite(sb.x).all()), t
    phis, ecs, oks = np.array(phis), np.array(ecs), np.array(oks)
    # no jump backwards on healthy ticks (a stream may creep back: dphi has no lower bound).  A stream that is replaying a plan it accepted on the reference's
    # rule (summed violation < 1e-4) from an unconverged iterate may retreat faster: round 5, a feasible point of the restoration phase, which ignores the
    # objective, applied at the iteration limit -- measured -0.14 per tick on one stream that then loses its plan
    dphi = np.diff(phis, axis=0)
    assert dphi[ecs[1:] == 0].min() > -0.05 and dphi.min() > -0.3, (dphi[ecs[1:] == 0].min(), dphi.min())
    assert phis[-1].max() > 5.0 and np.median(phis[-1]) > 3.0       # the loops got through the later segments
    assert oks.mean() > 0.9
    stuck = ecs[-1] >= 10                                           # error count past N: no plan is returned any more
    assert stuck.any(), ecs[-1].max()
    assert np.abs(phis[-1][stuck] - phis[-5][stuck]).max() == 0.0   # ... and the stream stays where it is
    sb.close(); slv.close()


def test_graph_replays_mixed_with_direct_launches_on_the_null_stream(tmp_path):
    """Torch-free reproducer of the round-2 fault, built with hipcc against the in-tree library: 256 problems x 200 rounds of
    {bmpc_graph_launch(g, NULL), bmpc_solve_batch(..., NULL)} without a host synchronisation, outputs bit-identical to a synchronised solve;
    part A checks the runtime's own ordering of a null-stream graph replay against null-stream launches with trivial kernels; the graph
    outlives bmpc_destroy of its handle and then refuses to launch."""
    import subprocess
    from boundmpc_amd import LIB_PATH, workload
    B, ticks = 256, 200
    P, X, _ = workload.make_batch(B, seed=3)
    prob = tmp_path / "problems.bin"
    with open(prob, "wb") as fh:
        P.astype(np.float64).tofile(fh); X.astype(np.float64).tofile(fh)
    exe = tmp_path / "graph_nullstream"
    libdir = os.path.dirname(os.path.realpath(LIB_PATH))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "--offload-arch=gfx950", "-o", str(exe), os.path.join(os.path.dirname(__file__), "cabi", "graph_nullstream.cpp"),
                           "-L" + libdir, "-lboundmpc_hip", "-Wl,-rpath," + libdir], stderr=subprocess.DEVNULL)
    r = subprocess.run(["timeout", "-k", "10", "120", str(exe), str(prob), str(B), str(ticks)], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert r.stdout.strip() == "A broken=0 B mismatch=0 status=ok"


def test_solver_close_destroys_its_graphs_first():
    """BatchedOCPSolver.close() closes the StepGraph / StreamBatch objects made from it before the handle goes (graph destroy
    touched a freed handle); closing them again afterwards is a no-op."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, workload
    P, X, _ = workload.make_batch(4, seed=5)
    s = BatchedOCPSolver(10, 4, 0.1)
    p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
    st = torch.cuda.Stream()
    g = s.capture_step(p, x0)
    with torch.cuda.stream(st):
        a = g.launch(stream=st)["x"].clone()
        b = s.solve_batch(p, x0, stream=st)["x"].clone()      # direct launch on the same handle right behind the replay
    st.synchronize()
    assert torch.equal(a, b)
    s.close()
    assert g._g is None
    g.close()


@pytest.mark.gpu
def test_streams_at_36_stages_fused_tick_matches_the_three_kernel_tick():
    """Closed-loop streams beyond 32 stages (round 4): the fused tick of the long-horizon instantiation (pack + solve + post in one launch,
    iterate in the workspace) against the three separate launches, 6 streams x 5 ticks at N = 36 -- same plants, plans and statuses; and the
    CPU build of the stream functions around the oracle retraces stream 0 (tests/test_stream.py pins that build against the host mirror)."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, stream as bstream, workload
    N = 36
    q0s = workload.random_q0(6, seed=11)
    runs = []
    for fused in (True, False):
        s = BatchedOCPSolver(N, 4, 0.1)
        mpcs, recs = [], []
        for q0 in q0s:
            m, p0fk = workload.make_mpc(q0, N=N)
            mpcs.append(m)
            recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0, 0]), np.zeros(7)))
        sb = bstream.StreamBatch(s, mpcs)
        sb.set_robot(np.stack(recs))
        for t in range(5):
            sb.tick(simulate=True, fused=fused)
        torch.cuda.synchronize()
        runs.append((sb.robot.cpu().numpy().copy(), sb.traj.cpu().numpy().copy(), sb.status.cpu().numpy().copy(), sb.x.cpu().numpy().copy()))
        sb.close(); s.close()
    (r1, t1, s1, x1), (r2, t2, s2, x2) = runs
    assert (s1 == 0).all() and np.array_equal(s1, s2)
    np.testing.assert_allclose(r1, r2, atol=1e-9); np.testing.assert_allclose(t1, t2, atol=1e-8); np.testing.assert_allclose(x1, x2, atol=1e-8)
    td, fl = bstream.unpack_traj(t1[0], N)
    assert fl["success"] and fl["n_valid"] == N and td["q"].shape == (7, N)


def _budgeted_closed_loops(slv, budget_us, cap=0, row_cap=0.0):
    """256 closed loops x 130 ticks under a time budget per fused tick (budget_us; 0 = none) and / or an iteration cap (cap; 0 = the handle's): tick times, plans kept / applied, plant joint positions, tube excess of the measured
    states (stream.tube_excess_of_state: the tube rows of casadi_ocp_formulation.py:316-349 at node 0 of the packed problem) and the first-stage
    position rows of the applied plans."""
    import torch
    from boundmpc_amd import stream as bstream, workload
    B, T = 256, 131
    q0s = workload.random_q0(B, seed=3)
    mpcs, recs = [], []
    for q0 in q0s:
        m, p0fk = workload.make_mpc(q0)
        mpcs.append(m)
        recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
    slv.set_rt_feasibility_tol(1e-2)
    slv.set_rt_position_row_cap(row_cap)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sb = bstream.StreamBatch(slv, mpcs)
        sb.set_robot(np.stack(recs))
        assert slv.team_info(B)["waves"] == 4      # 256 streams = the resident teams of an MI355X
        ms, Q, applied, tube_p, tube_r, row_p = [], [], [], [], [], []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for t in range(T):
            if t == 0:      # cold start from rest: to tolerance, without a budget
                sb.tick(max_iter=100, warm_dual=True, simulate=True)
                slv.set_time_budget_us(budget_us)      # read when the tick graph is captured
            else:
                e0.record(); sb.tick_graph(max_iter=cap, simulate=True, warm_dual=True, accept_capped=True); e1.record(); e1.synchronize()
                ms.append(e0.elapsed_time(e1))
                applied.append(float((sb.traj[:, -2] > 0.5).double().mean().item()))
                has_plan = (sb.state[:, bstream.SS["ERRCNT"]] < 10).cpu().numpy()
                ex_p, ex_r = bstream.tube_excess_of_state(sb.p.cpu().numpy())
                tube_p.append(np.where(has_plan, ex_p.max(axis=1), -np.inf)); tube_r.append(np.where(has_plan, ex_r.max(axis=1), -np.inf))
                app = (sb.traj[:, -2] > 0.5).cpu().numpy()
                row_p.append(np.where(app, sb.g.reshape(B, 10, 43)[:, 0, 39:41].cpu().numpy().max(axis=1), -np.inf))
            Q.append(sb.robot[:, :7].clone())
        alive = float((sb.state[:, bstream.SS["VALID"]] > 0.5).double().mean().item())
        Q = torch.stack(Q).cpu().numpy()
        sb.close(); slv.close()
    return np.array(ms), alive, float(np.mean(applied)), Q, np.array(tube_p), np.array(tube_r), np.array(row_p)


@pytest.mark.gpu
def test_256_streams_time_budgeted_tick_meets_the_1_khz_budget_over_130_ticks():
    """BASELINE configs[4] (256 closed-loop streams, one captured launch per tick, 1 ms budget), the round-4 mode that meets it: exact Hessian, dual
    state carried, KKT tolerance 1e-3, NO fixed iteration count but a time budget of 800 us per fused tick (bmpc_stream_set_time_budget), joint
    limits tightened by 2e-3 inside the solver, acceptance rule at 1e-2 with the variable bounds of the plan and of the re-integrated trajectory.
    Over the whole 130 ticks (through the hard third segment of the paths): tick p50 <= 1.0 ms and p99 <= 1.3 ms (HIP events around the graph
    launch), at least 75 % of the streams still hold a plan at the end (the converged loops: 93 %; the rest are the locally infeasible ticks of
    fixture g13), and no plant sample outside the joint limits.  Measured: 0.96 / 1.15 ms, 82 %, 0."""
    from boundmpc_amd import BatchedOCPSolver
    from boundmpc_amd.robot_model import RobotModel
    slv = BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=30, mu_warm=3e-2, bound_margin=2e-3)
    ms, alive, applied, Q, tube_p, tube_r, row_p = _budgeted_closed_loops(slv, 800)
    qlim = np.array(RobotModel().q_lim_upper)
    print(f"\n256 streams x {len(ms)} ticks, 800 us budget: tick p50 {np.percentile(ms, 50):.3f} / p99 {np.percentile(ms, 99):.3f} ms, applied {applied:.3f}, "
          f"streams with a plan at the end {alive:.3f}, plant samples beyond the joint limits {int((np.abs(Q) > qlim + 1e-9).sum())}")
    # the 1 kHz criteria (p50 <= 1.0 ms, p99 <= 1.3 ms) are REPORTED; asserted is a regression bound with room for the clock state of a shared box
    # (measured over rounds 4-5: 0.94-0.96 / 1.11-1.15 ms) -- a wall-clock assertion with 4 % of margin would make the suite flaky
    print("1 kHz criteria (p50 <= 1.0 ms, p99 <= 1.3 ms):", "MET" if np.percentile(ms, 50) <= 1.0 and np.percentile(ms, 99) <= 1.3 else "NOT MET in this run")
    assert np.percentile(ms, 50) <= 1.15 and np.percentile(ms, 99) <= 1.5
    assert alive >= 0.75 and applied >= 0.8
    assert (np.abs(Q) <= qlim + 1e-9).all()
    # tube compliance of the executed trajectories (round 5).  Position tube (exact): measured 0.15 % of the plant samples outside, by at most 1.8e-4 m
    # (tube half widths 0.01 ... 0.5 m); the position rows of the applied plans' first stage are within the acceptance threshold.  Orientation (the
    # exact split of the measured orientation error, which the NLP constrains only through its per-tick linearisation): 2.8 % of the samples, <= 0.15 rad
    # -- the loops that solve every tick to 1e-8 show 6.0 % / 0.18 rad on the same measure: it is a property of the reference's formulation.
    n = np.isfinite(tube_p).sum()
    print(f"tube compliance of {n} plant samples: position outside {(tube_p > 1e-6).sum() / n:.2e} (max {max(tube_p.max(), 0):.1e} m), orientation outside "
          f"{(tube_r > 1e-6).sum() / n:.2e} (max {max(tube_r.max(), 0):.2e} rad); applied plans: largest first-stage position row {max(row_p.max(), 0):.1e} m^2")
    assert (tube_p > 1e-6).sum() / n <= 5e-3 and tube_p.max() <= 1e-3
    assert (tube_r > 1e-6).sum() / n <= 0.08 and tube_r.max() <= 0.3
    assert row_p.max() <= 1e-2


@pytest.mark.gpu
def test_256_streams_on_a_fixed_barrier_level_meet_the_strict_1_khz_target():
    """configs[4], the STRICT reading (tick p99 <= 1.0 ms with >= 85 % of the streams keeping a plan): the real-time iteration of an interior-point method
    -- the barrier is not restarted and walked down in every tick, it stays on ONE level (BatchedOCPSolver(fixed_barrier=0.1): mu_init = mu_warm = final
    level), so the ~6.5 iterations that fit the budget of 625 us are Newton steps on a barrier problem whose solution the previous tick left nearby.
    Measured: tick p50 0.94 / p99 0.96 ms, 95.7 % of the streams hold a plan after 130 ticks (restarted barrier, 700 us: 76.6 %; loops solved to 1e-8: 93.0 %
    at 12.5 ms), 96.7 % of the ticks apply their plan, no plant sample outside the joint limits or the position tube, 0.4 % outside the orientation tube
    (loops solved to 1e-8: 5.9 %: the barrier keeps the plans off the tube walls).  The price: the loops are not the converged loops' (0.5 rad RMS apart in
    joint space: another resolution of the arm's redundancy) and progress along the path is 2 % slower."""
    from boundmpc_amd import BatchedOCPSolver
    from boundmpc_amd.robot_model import RobotModel
    slv = BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=30, fixed_barrier=0.1, bound_margin=2e-3)
    ms, alive, applied, Q, tube_p, tube_r, row_p = _budgeted_closed_loops(slv, 625, row_cap=1e-5)
    qlim = np.array(RobotModel().q_lim_upper)
    n = np.isfinite(tube_p).sum()
    print(f"\n256 streams x {len(ms)} ticks on the barrier level 0.1, 625 us budget: tick p50 {np.percentile(ms, 50):.3f} / p99 {np.percentile(ms, 99):.3f} ms, applied {applied:.3f}, "
          f"streams with a plan at the end {alive:.3f}; of {n} plant samples outside the position tube {(tube_p > 1e-6).sum() / n:.2e}, the orientation tube {(tube_r > 1e-6).sum() / n:.2e}")
    print("strict 1 kHz criteria (p99 <= 1.0 ms, >= 85 % of the streams with a plan):", "MET" if np.percentile(ms, 99) <= 1.0 and alive >= 0.85 else "NOT MET in this run")
    assert np.percentile(ms, 50) <= 1.1 and np.percentile(ms, 99) <= 1.3      # (wall clock on a shared box: the strict bound is reported above, a regression bound asserted)
    assert alive >= 0.90 and applied >= 0.93
    assert (np.abs(Q) <= qlim + 1e-9).all()
    assert (tube_p > 1e-6).sum() / n <= 1e-3 and (tube_r > 1e-6).sum() / n <= 0.02
    assert tube_p.max() <= 1e-3      # metres: the EXCESS, not only the fraction (position rows held per row: no accepted plan, replayed tail included, leaves the tube)


@pytest.mark.gpu
def test_tick_with_a_barrier_level_fallback():
    """StreamBatch.tick_with_fallback: converged ticks (at most 24 iterations) whose failures are solved again from the same warm start on a fixed barrier
    level and judged by the reference's rule at 1e-4.  32 closed loops x 60 ticks: healthy ticks are the plain converged ticks bit for bit (same x as
    a loop without the fallback while nothing fails); a tick forced to fail (iteration cap 2) is rescued by the fallback -- its plan is applied --
    where the plain tick replays the previous plan."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, stream as bstream, workload
    B = 32
    q0s = workload.random_q0(256, seed=3)[:B]
    mpcs, recs = [], []
    for q0 in q0s:
        m, p0fk = workload.make_mpc(q0)
        mpcs.append(m)
        recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
    def loop(with_fallback, ticks=60, fail_at=30):
        slv = BatchedOCPSolver(10, 4, 0.1, max_iter=100, stall_window=16); slv.set_restoration(False); slv.set_rt_feasibility_tol(1e-4)
        level = BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=14, fixed_barrier=1.0); level.set_restoration(False)
        sb = bstream.StreamBatch(slv, mpcs); sb.set_robot(np.stack(recs)); X, applied, nfb = [], [], []
        for t in range(ticks):
            cap = 2 if t == fail_at else 24      # a tick on which no solve can converge
            if with_fallback:
                nfb.append(sb.tick_with_fallback(level, max_iter=cap))
            else:
                sb.tick(max_iter=cap, warm_dual=True, simulate=True, fused=False, accept_capped=True); nfb.append(0)
            X.append(sb.x.clone()); applied.append((sb.traj[:, -2] > 0.5).cpu().numpy())
        alive = (sb.state[:, bstream.SS["VALID"]] > 0.5).cpu().numpy()
        sb.close(); slv.close(); level.close()
        return torch.stack(X), np.array(applied), np.array(nfb), alive
    Xa, app_a, nfb_a, alive_a = loop(True)
    Xb, app_b, nfb_b, alive_b = loop(False)
    first = int(np.argmax(nfb_a > 0))
    assert first > 5 and torch.equal(Xa[:first], Xb[:first])                      # until the first failure the fallback changes nothing
    assert nfb_a[30] == B and app_a[30].mean() >= 0.9 and app_b[30].mean() <= 0.1      # the forced failure: rescued by the level plans / previous plans replayed
    assert alive_a.mean() >= alive_b.mean() - 0.04 and alive_a.mean() >= 0.85


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0.1, "auto"])
def test_fixed_barrier_level_with_an_iteration_cap_is_reproducible_and_meets_the_strict_target(level):
    """The fixed-level real-time iteration WITHOUT a clock: exactly five Newton steps per stream and tick on the barrier level 0.1 (`tol` never fires on a
    fixed level, so the cap ends every solve).  Two runs of the 256 loops x 130 ticks are identical bit for bit (the time-budgeted modes are not: how many
    iterations fit depends on the clock), >= 90 % of the streams keep their plan, no plant sample leaves the joint limits, and the tick is p50 0.82 / p99
    0.98 ms (reported; asserted with room for a shared box)."""
    from boundmpc_amd import BatchedOCPSolver
    from boundmpc_amd.robot_model import RobotModel
    runs = []
    for rep in range(2):
        slv = BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=30, fixed_barrier=level, bound_margin=2e-3)
        runs.append(_budgeted_closed_loops(slv, 0, cap=5, row_cap=1e-5))
    (ms, alive, applied, Q, tube_p, tube_r, row_p), (ms2, alive2, applied2, Q2, *_rest) = runs
    qlim = np.array(RobotModel().q_lim_upper)
    n = np.isfinite(tube_p).sum()
    print(f"\n256 streams x {len(ms)} ticks, level {level}, five Newton steps per tick: tick p50 {np.percentile(ms, 50):.3f} / p99 {np.percentile(ms, 99):.3f} ms, applied {applied:.3f}, "
          f"streams with a plan at the end {alive:.3f}; outside the position tube {(tube_p > 1e-6).sum() / n:.2e} of {n} plant samples")
    print("strict 1 kHz criteria (p99 <= 1.0 ms, >= 85 % of the streams with a plan):", "MET" if np.percentile(ms, 99) <= 1.0 and alive >= 0.85 else "NOT MET in this run")
    assert np.array_equal(Q, Q2) and alive == alive2 and applied == applied2      # no clock in the result
    assert alive >= 0.90 and applied >= 0.93 and (np.abs(Q) <= qlim + 1e-9).all()
    assert np.percentile(ms, 50) <= 1.0 and np.percentile(ms, 99) <= 1.3
    assert (tube_p > 1e-6).sum() / n <= 1e-3
    # the EXCESS: until round 6 one plant sample of 32 636 was 5.1 mm outside an 18 mm tube -- a stream that had lost its plan replaying the tail of a plan
    # the summed rule (1e-2) had let through; with the position rows held per row (1e-5 m^2) the largest excess is negative: nothing leaves the tube
    assert tube_p.max() <= 1e-3, float(tube_p.max())


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0.1, "auto"])
def test_fixed_barrier_level_loops_on_the_gpu_retrace_the_cpu_mirror(level):
    """The capped fixed-level ticks have no clock in them, so they can be checked like everything else: 6 closed loops x 40 ticks (five Newton steps per
    tick on the level 0.1, duals and rejected iterates carried, acceptance at 1e-2 incl. the variable bounds) on the GPU -- fused team ticks from a captured
    graph -- against the CPU mirror: the g++ build of the stream functions (tests/emu) around the CPU oracle with the same options.  Plant joint positions
    to 1e-6 rad on every tick, the same ticks applied.  level "auto": the level sets itself per stream (clamp(0.02 (phi_max - phi), 0.01, 0.1), written into the
    dual state by the device-side pack, held inside the tick): mirrored by the CPU build of stream_pack with the same rule and the oracle's hold_mu."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, stream as bstream, workload
    from oracle import c_oracle
    from tests.emu import emu
    B, T, N, S, H = 6, 40, 10, 4, 0.1
    q0s = workload.random_q0(256, seed=3)[:B]
    mpcs, recs = [], []
    for q0 in q0s:
        m, p0fk = workload.make_mpc(q0)
        mpcs.append(m)
        recs.append(bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([m.phi_max[0], 0.0, 0.0]), np.zeros(7)))
    slv = BatchedOCPSolver(N, S, H, tol=1e-3, max_iter=30, fixed_barrier=level); slv.set_rt_feasibility_tol(1e-2)
    sb = bstream.StreamBatch(slv, mpcs); sb.set_robot(np.stack(recs))
    Qg, Ag = [], []
    for t in range(T):
        if t == 0:
            sb.tick(max_iter=100, warm_dual=True, simulate=True)
        else:
            sb.tick_graph(max_iter=5, warm_dual=True, simulate=True, accept_capped=True)
        torch.cuda.synchronize()
        Qg.append(sb.robot[:, :7].cpu().numpy().copy()); Ag.append((sb.traj[:, -2] > 0.5).cpu().numpy().copy())
    sb.close(); slv.close()
    auto = level == "auto"
    kw = dict(tol=1e-3, mu_init=0.1, mu_warm=0.01, mu_min_fac=10.0, hold_mu=1) if auto else dict(tol=1e-3, mu_init=0.1, mu_warm=0.1, mu_min_fac=100.0)
    rule = (0.02, 0.01, 0.1) if auto else (0.0, 0.0, 0.0)
    for b in range(B):
        Tb, M = bstream.path_table(mpcs[b].ref_path)
        ss = bstream.initial_state(mpcs[b], N); ss[bstream.SS["NENT"]] = M
        rb = recs[b].copy(); state = np.zeros((1, c_oracle.state_len(N))); xlast = None
        for t in range(T):
            p, x0 = emu.stream_pack(N, S, Tb, ss, rb, dual=state[0], xlast=xlast, level_rule=rule)
            r = c_oracle.solve(p, x0, N, S, H, opts=c_oracle.default_opts(max_iter=100 if t == 0 else 5, **kw), nthreads=1, state=state)
            tr = emu.stream_post(N, S, H, Tb, ss, rb, r["x"][0], r["g"][0], int(r["status"][0]), simulate=True, flags=0 if t == 0 else 2, rt_tol=1e-2)
            _, fl = bstream.unpack_traj(tr, N)
            xlast = r["x"][0]
            assert bool(fl["success"]) == bool(Ag[t][b]), (b, t)
            np.testing.assert_allclose(rb[bstream.RB["Q"]:bstream.RB["Q"] + 7], Qg[t][b], atol=1e-6, err_msg=f"stream {b} tick {t}")


@pytest.mark.gpu
def test_fixed_level_loops_reach_the_goals_of_the_reference_experiments():
    """The reference's own two experiments (run to the END of their paths, tubes down to +-0.01) as fixed-level closed loops through the stream API: on
    the barrier level 0.01 with eight Newton steps per tick both reach the goal (phi_max - phi <= 0.01) after 160 / 62 ticks -- the loops solved to 1e-8:
    155 / 59 -- with >= 97 % of the ticks applied and no failed stream; on the level 0.1 (the robust choice for the 130-tick benchmark loops) they stay
    alive but stall short of the end point.  The level that sets itself (fixed_barrier="auto": clamp(0.02 (phi_max - phi), 0.01, 0.1) per stream) reaches both
    goals after 161 / 64 ticks -- and keeps 94.9 % of the 256 benchmark plans (the next test): one default for both tasks."""
    import torch
    from boundmpc_amd import BatchedOCPSolver, stream as bstream
    ms = _mpcs()

    def run(level, K=8, ticks=200):
        slv = BatchedOCPSolver(10, 4, 0.1, tol=1e-3, max_iter=30, fixed_barrier=level); slv.set_rt_feasibility_tol(1e-2)
        sb = bstream.StreamBatch(slv, [m for m, _ in ms]); sb.set_robot(np.stack([_robot0(m, d) for m, d in ms]))
        done, app = [None, None], []
        for t in range(ticks):
            if t == 0:
                sb.tick(max_iter=100, warm_dual=True, simulate=True)
            else:
                sb.tick_graph(max_iter=K, warm_dual=True, simulate=True, accept_capped=True)
            torch.cuda.synchronize()
            st = sb.state.cpu().numpy(); app.append((sb.traj[:, -2] > 0.5).cpu().numpy().copy())
            for b in range(2):
                if done[b] is None and ms[b][0].phi_max[0] - st[b, bstream.SS["PHI"]] <= 0.01:
                    done[b] = t + 1
            if all(d is not None for d in done):
                break
        valid = st[:, bstream.SS["VALID"]].copy(); phi = st[:, bstream.SS["PHI"]].copy()
        sb.close(); slv.close()
        return done, np.mean(app, axis=0), valid, phi
    done, app, valid, phi = run(0.01)
    assert done[0] is not None and done[1] is not None and done[0] <= 170 and done[1] <= 70 and app.min() >= 0.97 and valid.all(), (done, app)
    done, app, valid, phi = run("auto")      # ONE default for both tasks (round 6): the level follows the distance to the end of the path
    assert done[0] is not None and done[1] is not None and done[0] <= 172 and done[1] <= 72 and app.min() >= 0.97 and valid.all(), (done, app)
    done, app, valid, phi = run(0.1)
    assert done == [None, None] and valid.all() and app.min() >= 0.97 and phi[0] > 6.0 and phi[1] > 1.0      # alive, applied, short of the end point

