"""Diagnostic (CPU, build container): the closed loops of BASELINE configs[4] replayed without a GPU -- the stream functions of
csrc/bmpc_stream.inl (g++ build, tests/emu) around the CPU oracle as the solver -- to find out WHAT the failing / iteration-capped ticks of the
converged mode are (7 % of 256 random closed loops lose their plan, "in almost every tick some stream runs to the
100-iteration cap").  Every tick whose solve does not converge, hits the cap, or needs many iterations is recorded with its (p, x0), and the
hard ones are handed to scipy's SLSQP from the same start (oracle/solve_scipy_batch.py's solver): feasible KKT point or not?

  python tests/cpu_closed_loop.py [--streams 64] [--ticks 130] [--procs 6] [--slsqp 24]   ->  tests/golden/g13_hard_ticks.npz + a report on stdout
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boundmpc_amd import stream as bstream, workload  # noqa: E402
from oracle import c_oracle, nlp  # noqa: E402
from tests.emu import emu  # noqa: E402

N, S, H = 10, 4, 0.1


def run_stream(args):
    b, q0, ticks, max_iter, opts_kw = args
    mpc, p0fk = workload.make_mpc(q0)
    T, M = bstream.path_table(mpc.ref_path)
    ss = bstream.initial_state(mpc, N); ss[bstream.SS["NENT"]] = M
    rb = bstream.robot_record(q0, np.zeros(7), np.zeros(7), p0fk, np.zeros(6), np.array([mpc.phi_max[0], 0.0, 0.0]), np.zeros(7))
    o = c_oracle.default_opts(max_iter=max_iter, **opts_kw)
    rows = []
    for t in range(ticks):
        p, x0 = emu.stream_pack(N, S, T, ss, rb)
        r = c_oracle.solve(p, x0, N, S, H, opts=o, nthreads=1)
        x, g, st, it = r["x"][0], r["g"][0], int(r["status"][0]), int(r["iters"][0])
        tr = emu.stream_post(N, S, H, T, ss, rb, x, g, st, simulate=True)
        _, fl = bstream.unpack_traj(tr, N)
        rows.append(dict(b=b, t=t, status=st, iters=it, kkt=float(r["kkt"][0]), success=fl["success"], using_previous=fl["using_previous"],
                         n_valid=fl["n_valid"], g_viol=fl["g_viol"], phi=float(ss[bstream.SS["PHI"]]), phi_max=float(ss[bstream.SS["PHIMAX"]]),
                         errcnt=int(ss[bstream.SS["ERRCNT"]]), valid=bool(ss[bstream.SS["VALID"]] > 0.5), p=p, x0=x0, x=x))
        if not rows[-1]["valid"]:
            break
    return rows


def slsqp_job(job):
    i, p, x0 = job
    from oracle.solve_scipy_batch import solve
    t0 = time.time()
    res = solve(p, x0, N=N, maxiter=1500)
    f, g = nlp.nlp_eval(res.x, p, N, S, H)
    g2 = g.reshape(N, 43)
    lbx, ubx, _, _ = nlp.bounds(N)
    bx = float(max((lbx - res.x).max(), (res.x - ubx).max(), 0.0))
    return i, res.x, float(f), int(res.nit), int(res.status), float(np.abs(g2[:, :36]).max()), float(g2[:, 36:].max()), bx, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--ticks", type=int, default=130)
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--max-iter", type=int, default=100)
    ap.add_argument("--slsqp", type=int, default=24, help="hard ticks handed to SLSQP (0 = none)")
    ap.add_argument("--first-only", action="store_true", help="hand only the FIRST failing tick of every stream to SLSQP (what follows is a warm start from an outdated plan)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "g13_hard_ticks.npz"))
    a = ap.parse_args()
    q0s = workload.random_q0(256, seed=3)[:a.streams]
    with mp.get_context("fork").Pool(a.procs) as pool:
        allrows = pool.map(run_stream, [(b, q0s[b], a.ticks, a.max_iter, {}) for b in range(a.streams)], chunksize=1)
    rows = [r for rr in allrows for r in rr]
    its = np.array([r["iters"] for r in rows]); st = np.array([r["status"] for r in rows])
    lost = [rr[-1] for rr in allrows if not rr[-1]["valid"]]
    print(f"{a.streams} streams x {a.ticks} ticks: {len(rows)} ticks solved, mean iterations {its.mean():.1f}, p50 {np.median(its):.0f}, p99 {np.percentile(its, 99):.0f}, "
          f"at the cap {int((its >= a.max_iter).sum())}; status 0/1/2/3: {[int((st == k).sum()) for k in range(4)]}; not applied {sum(1 for r in rows if r['using_previous'])}; "
          f"streams that lost their plan {len(lost)} ({[(r['b'], r['t']) for r in lost]})")
    # hard ticks: everything that did not converge, plus the slowest converged ones
    hard = [r for r in rows if r["status"] != 0]
    slow = sorted([r for r in rows if r["status"] == 0 and r["iters"] >= 40], key=lambda r: -r["iters"])
    print(f"non-converged ticks {len(hard)}; converged with >= 40 iterations {len(slow)}")
    for r in hard[:40]:
        print(f"  stream {r['b']:3d} tick {r['t']:3d}: status {r['status']} iters {r['iters']:3d} kkt {r['kkt']:.1e} g_viol {r['g_viol']:.1e} phi {r['phi']:.3f}/{r['phi_max']:.3f} errcnt {r['errcnt']}")
    # SLSQP gets the FIRST failing tick of every stream (what follows a failure is a warm start from an outdated plan), then the slow ones
    first = [r for r in hard if r["errcnt"] == 1]
    pick = ((first if a.first_only else first + slow + [r for r in hard if r["errcnt"] == 2])[:a.slsqp]) if a.slsqp else []
    out = {}
    if pick:
        with mp.get_context("fork").Pool(a.procs) as pool:
            res = pool.map(slsqp_job, [(i, r["p"], r["x0"]) for i, r in enumerate(pick)], chunksize=1)
        for i, x, f, nit, code, eq, iq, bx, secs in res:
            r = pick[i]
            fo, go = nlp.nlp_eval(r["x"], r["p"], N, S, H)
            g2 = go.reshape(N, 43)
            d = (x - r["x"]).reshape(N, 44)[:, 8:15]
            print(f"  SLSQP stream {r['b']:3d} tick {r['t']:3d} (ours: status {r['status']}, {r['iters']} its, f {fo:.8g}, eq {np.abs(g2[:, :36]).max():.1e}, ineq {g2[:, 36:].max():.1e}): "
                  f"exit {code} nit {nit} f {f:.8g} eq {eq:.1e} ineq {iq:.1e} bounds {bx:.1e}; joint RMS vs ours {np.sqrt((d ** 2).mean()):.2e} rad  [{secs:.0f} s]")
        out = dict(stream=np.array([r["b"] for r in pick]), tick=np.array([r["t"] for r in pick]), p=np.array([r["p"] for r in pick]), x0=np.array([r["x0"] for r in pick]),
                   oracle_status=np.array([r["status"] for r in pick]), oracle_iters=np.array([r["iters"] for r in pick]), oracle_x=np.array([r["x"] for r in pick]),
                   slsqp_x=np.array([x for _, x, *_ in res]), slsqp_f=np.array([v[2] for v in res]), slsqp_nit=np.array([v[3] for v in res]),
                   slsqp_exit=np.array([v[4] for v in res]), slsqp_eq=np.array([v[5] for v in res]), slsqp_ineq=np.array([v[6] for v in res]),
                   slsqp_bounds=np.array([v[7] for v in res]))
        np.savez_compressed(a.out, **out)
        print("wrote", a.out)


if __name__ == "__main__":
    main()
