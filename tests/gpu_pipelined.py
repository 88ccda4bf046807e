"""Diagnostic (GPU box): throughput with SEVERAL independent batches of 1024 in flight (one solver handle and one HIP stream
each), next to the one-batch-at-a-time figure of bench.py.  With a single batch of 1024 on 1024 wave slots the kernel time is
the latency of the slowest problem (36 iterations against a mean of 13); a second batch in flight fills the slots that drain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boundmpc_amd import BatchedOCPSolver, workload
B, steps = 1024, 24
dev = torch.device("cuda", 0)
for nfl in (1, 2, 3, 4):
    solvers = [BatchedOCPSolver(10, 4, 0.1) for _ in range(nfl)]
    streams = [torch.cuda.Stream(dev) for _ in range(nfl)]
    data = []
    for j in range(nfl):
        P, X, _ = workload.make_batch(B, seed=j)
        data.append((torch.tensor(P, device=dev), torch.tensor(X, device=dev), {}))
    torch.cuda.synchronize()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s_ in range(steps):
            j = s_ % nfl
            with torch.cuda.stream(streams[j]):
                solvers[j].solve_batch(data[j][0], data[j][1], out=data[j][2], want=("status",))
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    ok = all(int((d[2]["status"] == 0).sum()) == B for d in data)
    print(f"{nfl} batch(es) of {B} in flight: {B * steps / el:,.0f} solves/s ({el / steps * 1e3:.2f} ms per batch), all converged: {ok}", flush=True)
    for s in solvers: s.close()
