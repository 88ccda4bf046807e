"""Multi-process path of the sharded solve (world_size 2, gloo, CPU): contiguous batch slices per rank, no
exchange during the solve, one all-gather of the solutions.  The per-rank solve is stood in for by the lane
emulator here (the HIP path needs a GPU); what is tested is the sharding / gather plumbing bench.py uses."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as tmp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from boundmpc_amd import workload
    from boundmpc_amd.distributed import shard_range, solve_sharded
    from tests.emu import emu
    P, X, _ = workload.make_batch(B, seed=11, workers=1)

    def solve_fn(p, x0):
        return torch.from_numpy(emu.solve(p.numpy(), x0.numpy(), 10, 4, 0.1)["x"])
    x = solve_sharded(solve_fn, torch.from_numpy(P), torch.from_numpy(X))
    lo, hi = shard_range(B, rank, world)
    q.put((rank, lo, hi, x.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(B):
    from tests.emu import emu
    from boundmpc_amd import workload
    emu.build()
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P, X, _ = workload.make_batch(B, seed=11, workers=1)
    full = emu.solve(P, X, 10, 4, 0.1)["x"]
    ranges = sorted((r[1], r[2]) for r in res)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == B
    for _, _, _, x in res:
        np.testing.assert_array_equal(x, full)        # every rank holds the full, correctly ordered solution


def test_sharded_solve_even():
    _run(8)


def test_sharded_solve_ragged():
    _run(7)


def test_shard_range_covers_batch():
    from boundmpc_amd.distributed import shard_range
    for B in (0, 1, 7, 8, 1024, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_rank_shards_of_the_global_batch_tile_it():
    """bench.py --gpus N: rank r packs only rows shard_range(G, r, shards) of the global batch (workload.make_batch(rows=...));
    the shards must tile the batch a single process would generate, in order, with no problem repeated."""
    from boundmpc_amd import workload
    from boundmpc_amd.distributed import shard_range
    G_, shards = 24, 4
    P, X, Q = workload.make_batch(G_, seed=1, workers=1)
    parts = [workload.make_batch(G_, seed=1, workers=1, rows=shard_range(G_, r, shards)) for r in range(shards)]
    np.testing.assert_array_equal(np.concatenate([a[0] for a in parts]), P)
    np.testing.assert_array_equal(np.concatenate([a[1] for a in parts]), X)
    assert len({tuple(q) for q in Q}) == G_
    # the first rows of a larger draw are the smaller batch (so rank 0's shard of the 1024 x N batch is configs[1] itself)
    np.testing.assert_array_equal(workload.random_q0(6, 0), workload.random_q0(24, 0)[:6])
