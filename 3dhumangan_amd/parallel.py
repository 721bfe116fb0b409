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


def r1_allgather(local_stat, group=None):
    """The one data-path collective of the training step that the north_star names ("RCCL all-gather over xGMI for the
    discriminator R1 step only"): all-gather of the R1 statistics of every rank's batch shard -- the per-sample
    ||grad_x D(x_i)||^2 [b_r], or the reference's per-channel norms of the shard's first sample [C] -> one vector with every
    rank's values, identical on every rank, so that every rank applies the SAME penalty 0.5 * r1_lambda * mean(...) (the
    reference gets a mean over ranks implicitly, through DDP's gradient averaging of per-rank penalties:
    lib/trainers/phase_trainer.py:259-294, 392).  A few floats per rank: latency-bound.

    Shards may be uneven (shard_bounds gives earlier ranks the remainder): the lengths are exchanged first and the values
    travel padded to the longest shard, as gather_images does.

    Autograd: this rank's slice of the result keeps its graph (the double-backward through D), the other ranks' slices are
    constants -- backward of mean(result) therefore yields exactly this rank's share of the global gradient."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_stat
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if local_stat.dim() != 1:
        raise ValueError(f"r1_allgather takes a vector of statistics, got shape {tuple(local_stat.shape)}")
    n = torch.tensor([local_stat.shape[0]], dtype=torch.int64, device=local_stat.device)
    lens = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(lens, n, group=group)
    lens = [int(t.item()) for t in lens]
    longest = max(lens)
    pad = local_stat.new_zeros(longest)
    pad[: lens[rank]] = local_stat.detach()
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    parts = [t[:k] for t, k in zip(parts, lens)]
    parts[rank] = local_stat
    return torch.cat(parts, dim=0)


def allreduce_gradients(parameters, average=True, group=None, bucket_bytes=64 << 20):
    """Sum (or average) the .grad of `parameters` over the ranks with a few large flat all-reduces (RCCL ring all-reduce is
    per-link bound on xGMI: fewer, larger messages -- 64 MB buckets -- instead of one collective per tensor).

    The set of tensors that travel is the same on every rank by construction: every ``requires_grad`` parameter that has a
    gradient on ANY rank (one small MAX all-reduce of the has-gradient flags first); a rank that lacks one of them
    contributes zeros.  Parameters without a gradient anywhere (a head the loss does not touch, an unused latent pool) stay
    ``None`` everywhere, as under DDP.  The reduction writes straight into views of the flat bucket: one pack, one collective,
    one unpack per bucket."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return
    dev = params[0].device
    flags = torch.tensor([0 if p.grad is None else 1 for p in params], dtype=torch.int32, device=dev)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
    live = [p for p, f in zip(params, flags.tolist()) if f]
    for p in live:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    bucket, size = [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        torch._foreach_copy_(bucket, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in bucket]), bucket)])
        bucket.clear()

    for p in live:
        g = p.grad
        if bucket and (g.dtype != bucket[0].dtype):
            flush()
            size = 0
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
            size = 0
    flush()


def max_over_ranks(seconds, device=None):
    """The bench's timing reduction: the job is as slow as its slowest rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
