"""Multi-GPU sharding of a batch of independent OCPs: contiguous slices of the batch
dimension per rank, no exchange during the solve, ONE all-gather of the solution tensor at the end
(RCCL over xGMI on the GPU box; gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """Contiguous [lo, hi) slice of a batch of B problems for `rank` of `world` (sizes differ by at most one)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_solutions(x_local, B, group=None):
    """All-gather the per-rank solution blocks [b_r][n_w] into the full [B][n_w] tensor on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return x_local
    if x_local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal path only (bench.py --rehearse-one-gpu): gloo has no device all-gather, stage through the host
        return gather_solutions(x_local.cpu(), B, group).to(x_local.device)
    sizes = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
    nw = x_local.shape[1]
    if len(set(sizes)) == 1:
        out = torch.empty((B, nw), dtype=x_local.dtype, device=x_local.device)
        dist.all_gather_into_tensor(out, x_local.contiguous(), group=group)
        return out
    mx = max(sizes)
    pad = torch.zeros((mx, nw), dtype=x_local.dtype, device=x_local.device)
    pad[:x_local.shape[0]] = x_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][:sizes[r]] for r in range(world)], dim=0)


def solve_sharded(solve_fn, p, x0, group=None):
    """Each rank solves its slice with solve_fn(p_slice, x0_slice) -> x_slice, then all ranks gather x."""
    B = p.shape[0]
    if dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_range(B, rank, world)
    x_local = solve_fn(p[lo:hi], x0[lo:hi])
    return gather_solutions(x_local, B, group)
