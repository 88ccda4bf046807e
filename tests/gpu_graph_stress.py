"""Diagnostic (GPU box): captured steps and direct launches of ONE handle mixed at random over three streams (the null stream and two explicit ones), every
kernel choice (one wave, pairs, teams), without host synchronisation between the launches of a round; every result is compared with a reference handle
that runs alone on its own stream.  Written after the memset-node finding of round 6 (profiles/r06_e_graph_memset_node.txt).
Usage: python tests/gpu_graph_stress.py [rounds]"""
import os, sys, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from boundmpc_amd import BatchedOCPSolver, workload
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(7)
B = 200
batches = [workload.make_batch(B, seed=90 + i)[:2] for i in range(4)]
ref = BatchedOCPSolver(10, 4, 0.1); ref.set_team_waves(1)
refstream = torch.cuda.Stream()
expect = {}
for i, (P, X) in enumerate(batches):      # cold solves capped at four iterations: what every launch below must return (to round-off across kernels)
    with torch.cuda.stream(refstream):
        o = ref.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda"), out={}, state=ref.new_state(B), max_iter=4, stream=refstream)
    refstream.synchronize(); expect[i] = o["x"].clone()
s = BatchedOCPSolver(10, 4, 0.1)
streams = [None, torch.cuda.Stream(), torch.cuda.Stream()]
graphs = []
for waves in (1, 2, 4):
    s.set_team_waves(waves)
    p = torch.empty((B, 505), dtype=torch.float64, device="cuda"); x0 = torch.empty((B, 440), dtype=torch.float64, device="cuda")
    st = s.new_state(B)
    graphs.append((waves, p, x0, st, s.capture_step(p, x0, state=st, max_iter=4)))
bad = 0; launches = 0
for r in range(rounds):
    pending = []; used = set()
    for _ in range(rng.randint(2, 5)):
        i = rng.randrange(4); P, X = batches[i]; stream = rng.choice(streams)
        cur = stream if stream is not None else torch.cuda.default_stream()
        if rng.random() < 0.5:
            free = [g_ for g_ in graphs if g_[0] not in used]      # a graph's buffers are the caller's: refreshed once per round (a second refresh from another
            if not free: continue                                 # stream would race with the first replay -- the caller's ordering, not the library's)
            waves, p, x0, st, g = rng.choice(free); used.add(waves)
            with torch.cuda.stream(cur):
                p.copy_(torch.tensor(P), non_blocking=False); x0.copy_(torch.tensor(X)); st.zero_()      # cold state: the capped result depends on nothing else
                out = g.launch(cur)["x"]
                keep = out.clone()
            pending.append((i, f"graph waves {waves} stream {streams.index(stream)}", keep, cur))
        else:
            waves = rng.choice((0, 1, 2, 4)); s.set_team_waves(waves)
            with torch.cuda.stream(cur):
                o = s.solve_batch(torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda"), out={}, state=s.new_state(B), max_iter=4, stream=cur)
            pending.append((i, f"direct waves {waves} stream {streams.index(stream)}", o["x"], cur))
        launches += 1
    torch.cuda.synchronize()
    for i, what, x, _ in pending:
        d = float((x - expect[i]).abs().max())
        if not d < 1e-7:
            bad += 1
            if bad <= 10: print(f"round {r}: {what}: max |x - reference| {d:.3e}", flush=True)
    if r % 25 == 0: print(f"round {r}: {launches} launches so far, {bad} wrong", flush=True)
print(f"{launches} launches in {rounds} rounds over three streams, graphs and direct launches of one handle mixed: {bad} wrong")
for g in graphs: g[4].close()
s.close(); ref.close()
sys.exit(1 if bad else 0)
