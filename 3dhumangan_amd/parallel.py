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


def r1_allgather(local_stat, group=None, equal=False):
    """The one data-path collective of the training step that the north_star names ("RCCL all-gather over xGMI for the
    discriminator R1 step only"): all-gather of the R1 statistics of every rank's batch shard -- the per-sample
    ||grad_x D(x_i)||^2 [b_r], or the reference's per-channel norms of the shard's first sample [C] -> one vector with every
    rank's values, identical on every rank, so that every rank applies the SAME penalty 0.5 * r1_lambda * mean(...) (the
    reference gets a mean over ranks implicitly, through DDP's gradient averaging of per-rank penalties:
    lib/trainers/phase_trainer.py:259-294, 392).  A few floats per rank: latency-bound.

    ``equal=True``: every rank contributes the same number of values (always true for the default "reference" statistic --
    C channel norms -- and for per-sample statistics of an evenly sharded batch): ONE collective, no host synchronisation.
    Otherwise shards may be uneven (shard_bounds gives earlier ranks the remainder): the lengths are exchanged first (one
    more small all-gather and a host read) and the values travel padded to the longest shard, as gather_images does.

    Autograd: this rank's slice of the result keeps its graph (the double-backward through D), the other ranks' slices are
    constants -- backward of mean(result) therefore yields exactly this rank's share of the global gradient."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_stat
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if local_stat.dim() != 1:
        raise ValueError(f"r1_allgather takes a vector of statistics, got shape {tuple(local_stat.shape)}")
    if equal:
        parts = [torch.empty_like(local_stat) for _ in range(world)]
        dist.all_gather(parts, local_stat.detach().contiguous(), group=group)
        parts[rank] = local_stat
        return torch.cat(parts, dim=0)
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


DDP_BUCKET_BYTES = 25 << 20        # the reference's DDP default (bucket_cap_mb = 25: lib/trainers/base_trainer.py:102-104)


class GradReducer:
    """Bucketed gradient all-reduce that runs WHILE backward is still running (what DDP's reducer does for the reference:
    lib/trainers/base_trainer.py:102-104).  The parameters are cut into flat buckets of ~`bucket_bytes` (25 MB, DDP's default: a
    95-122 MB discriminator gives 4-5 buckets, the 39 MB generator 2, so the first all-reduce starts a quarter into backward);
    a post-accumulate hook on every parameter marks it ready and, once a bucket is complete, packs the bucket with ONE
    multi-tensor copy and launches its all-reduce asynchronously -- RCCL works on its own stream while autograd keeps computing
    the earlier layers' gradients.  Buckets are always launched in bucket order, on every rank, whatever order the hooks fire
    in, so the collective sequence is the same everywhere.

        reducer = GradReducer(D.parameters())          # once
        reducer.prepare(); loss.backward(); reducer.finish()      # every step

    Bucket order: the first step uses reverse registration order (about the order in which backward produces gradients); at
    the end of that step the buckets are REBUILT in the order in which the gradients actually arrived on rank 0 (broadcast, so
    every rank cuts the same buckets; DDP does the same after its first iteration), with the parameters that produced no
    gradient in a last bucket of their own -- a head the loss does not touch then no longer holds bucket 0 back until finish().

    ``finish`` launches what is left (a bucket with parameters that received no gradient on this rank travels with zeros in
    their place), waits, divides by the world size (``average``) and writes the reduced gradients back to ``p.grad``.  Which
    parameters have a gradient on ANY rank rides along in the bucket itself (one flag per parameter, summed): a parameter
    with no gradient anywhere keeps ``grad = None``, as under DDP; the flags are only read (a host synchronisation) on a rank
    that is missing a gradient -- never in the steady state where every rank produces every gradient.
    A gradient that arrives after its bucket was reduced (a second backward between prepare() and finish()) raises instead of
    leaving a stale reduction behind.  World size 1 / no process group: prepare / finish do nothing."""

    def __init__(self, parameters, average=True, group=None, bucket_bytes=DDP_BUCKET_BYTES, rebuild=True):
        self.params = [p for p in parameters if p.requires_grad]
        self.average, self.group, self.bucket_bytes = average, group, bucket_bytes
        self.active = dist.is_initialized() and dist.get_world_size(group) > 1 and bool(self.params)
        self.buckets, self._armed, self._handles, self._hooks = [], False, [], []
        self._rebuild_pending, self._arrival = bool(rebuild), []
        if not self.active:
            return
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._cut(list(reversed(self.params)))
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._hook))

    def signature(self):
        """What a cached reducer was built for: the identity of the parameters that require gradients."""
        return tuple(id(p) for p in self.params)

    def close(self):
        """Remove the hooks (a reducer that is being replaced must not keep firing)."""
        for h in self._hooks:
            h.remove()
        self._hooks, self.active, self._armed = [], False, False

    def _cut(self, ordered, tail=()):
        """Buckets over `ordered` (then `tail`, never merged into a bucket of `ordered`)."""
        self.buckets = []
        for group_ in (ordered, list(tail)):
            cur, size = [], 0
            for p in group_:
                if cur and (p.dtype != cur[0].dtype or p.device != cur[0].device or size >= self.bucket_bytes):
                    self._close(cur)
                    cur, size = [], 0
                cur.append(p)
                size += p.numel() * p.element_size()
            if cur:
                self._close(cur)
        self._slot = {}
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b["params"]):
                self._slot[id(p)] = (bi, pi)

    def _close(self, params):
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n + len(params), dtype=params[0].dtype, device=params[0].device)     # gradients, then one flag each
        views = [v.view_as(p) for v, p in zip(flat[:n].split([p.numel() for p in params]), params)]
        self.buckets.append(dict(params=list(params), flat=flat, views=views, flags=flat[n:], ready=0, launched=False,
                                 have=[False] * len(params)))

    def prepare(self):
        """Arm the hooks for the backward that follows."""
        if not self.active:
            return
        for b in self.buckets:
            b["ready"], b["launched"], b["have"] = 0, False, [False] * len(b["params"])
        self._handles, self._next, self._armed, self._arrival = [], 0, True, []

    def _hook(self, p):
        if not self._armed:
            return
        bi, pi = self._slot[id(p)]
        b = self.buckets[bi]
        if b["launched"]:
            raise RuntimeError("GradReducer: a gradient arrived after its bucket was reduced (two backward passes between "
                               "prepare() and finish()?)")
        if not b["have"][pi]:                   # the gradient itself stays in p.grad until the bucket is packed (_launch)
            b["have"][pi] = True
            b["ready"] += 1
            if self._rebuild_pending:
                self._arrival.append(self._index[id(p)])
            self._launch_ready()

    def _launch(self, b):
        have = [pi for pi, h in enumerate(b["have"]) if h]
        if have:                                # ONE multi-tensor copy per bucket, not one launch per parameter
            torch._foreach_copy_([b["views"][pi] for pi in have], [b["params"][pi].grad for pi in have])
        if len(have) == len(b["have"]):
            b["flags"].fill_(1.0)
        else:
            for pi, h in enumerate(b["have"]):
                if not h:
                    b["views"][pi].zero_()
            b["flags"].copy_(torch.tensor([1.0 if h else 0.0 for h in b["have"]], dtype=b["flat"].dtype))
        self._handles.append(dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        b["launched"] = True

    def _launch_ready(self):
        while self._next < len(self.buckets) and self.buckets[self._next]["ready"] == len(self.buckets[self._next]["params"]):
            self._launch(self.buckets[self._next])
            self._next += 1

    def finish(self):
        """After backward: reduce what has not been launched yet, wait, write the reduced gradients to ``p.grad``."""
        if not self.active:
            return
        if not self._armed:
            raise RuntimeError("GradReducer.finish() without prepare()")
        self._armed = False
        for b in self.buckets[self._next:]:
            for pi, p in enumerate(b["params"]):            # gradients produced outside the hooks' reach (set by hand)
                if not b["have"][pi] and p.grad is not None:
                    b["have"][pi] = True
            self._launch(b)
        self._next = len(self.buckets)
        for h in self._handles:
            h.wait()
        world = dist.get_world_size(self.group)
        for b in self.buckets:
            if self.average:
                b["flat"][: b["flat"].numel() - len(b["params"])].div_(world)
            missing = [pi for pi, h in enumerate(b["have"]) if not h]
            anywhere = None
            if missing:                                       # rare: read the summed flags to see who else had a gradient
                anywhere = (b["flags"] > 0).tolist()
            dst, src = [], []
            for pi, p in enumerate(b["params"]):
                if b["have"][pi]:
                    dst.append(p.grad)
                    src.append(b["views"][pi])
                elif anywhere[pi]:
                    p.grad = b["views"][pi].clone()
            if dst:
                torch._foreach_copy_(dst, src)
        if self._rebuild_pending:
            self._rebuild()

    def _rebuild(self):
        """Once, after the first step: buckets in rank 0's gradient-arrival order, the parameters without a gradient last."""
        self._rebuild_pending = False
        n = len(self.params)
        seen = set(self._arrival)
        order = self._arrival + [i for i in range(n) if i not in seen]
        t = torch.tensor(order + [len(self._arrival)], dtype=torch.int64, device=self.params[0].device)
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        got = t.tolist()
        order, n_arrived = got[:-1], got[-1]
        if sorted(order) != list(range(n)):
            raise RuntimeError("GradReducer: rank 0 broadcast an inconsistent bucket order")
        self._cut([self.params[i] for i in order[:n_arrived]], [self.params[i] for i in order[n_arrived:]])


def allreduce_gradients(parameters, average=True, group=None, bucket_bytes=DDP_BUCKET_BYTES):
    """Sum (or average) the .grad of `parameters` over the ranks with a few large flat all-reduces (RCCL ring all-reduce is
    per-link bound on xGMI: fewer, larger messages -- 25 MB buckets, DDP's default -- instead of one collective per tensor).

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


def reducer_of(module, **kw):
    """The module's GradReducer, created at first use (its hooks stay on the parameters: one reducer per module).  Rebuilt --
    the old one's hooks removed first -- when the process group appears / disappears or when the set of parameters that
    require gradients changes (a frozen sub-network, a new training phase)."""
    red = getattr(module, "_h3d_grad_reducer", None)
    sig = tuple(id(p) for p in module.parameters() if p.requires_grad)
    # (a reducer over NO trainable parameter is inactive whatever the process group says: compare like with like, or a module
    # without trainable parameters gets a new reducer at every call)
    want_active = dist.is_initialized() and dist.get_world_size(kw.get("group")) > 1 and bool(sig)
    if red is None or want_active != red.active or sig != red.signature():
        if red is not None:
            red.close()
        red = GradReducer(module.parameters(), **kw)
        object.__setattr__(module, "_h3d_grad_reducer", red)
    return red
