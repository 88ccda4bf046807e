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
            ss[bstream.SS["HASPREV"]] = 1; ss[bstream.SS["PREV"]:] = d7["x"][t - 1]
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
