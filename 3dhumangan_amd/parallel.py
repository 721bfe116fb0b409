"""Batch sharding of generator inference across the GPUs of one node (one process per GPU, torch.distributed).

Inference needs no data-path collective: every sample is independent in eval mode, so ranks take a
contiguous slice of the batch and run the same kernels on replicated weights.  The only optional exchange is an
all-gather of the finished images (RCCL over xGMI on GPUs, gloo in the CPU tests) when one rank wants them all.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) slice of n items for `rank` of `world` (earlier ranks take the remainder)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(latent, conditions, rank=None, world=None):
    """Slice the batch dimension of the latents and of every tensor in `conditions`."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(latent.shape[0], rank, world)
    return latent[lo:hi], {k: v[lo:hi] for k, v in conditions.items()}


def gather_images(local, total, group=None):
    """All-gather variable-size batch shards of images [b_r, C, H, W] into [total, C, H, W] on every rank.
    Shards are padded to the largest one so a single fixed-size all_gather moves the data."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    biggest = (total + world - 1) // world
    pad = local.new_zeros((biggest,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    parts = []
    for r, t in enumerate(out):
        lo, hi = shard_bounds(total, r, world)
        parts.append(t[: hi - lo])
    return torch.cat(parts, dim=0)


def max_over_ranks(seconds, device=None):
    """The bench's timing reduction: the job is as slow as its slowest rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
