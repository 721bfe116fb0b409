"""World-size-2 (and 3) gloo tests of the batch-sharding helpers used by bench.py / the app (CPU only)."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

par = importlib.import_module("3dhumangan_amd.parallel")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z = torch.arange(total * 4, dtype=torch.float32).reshape(total, 4)
        cond = {"a": torch.arange(total * 6, dtype=torch.float32).reshape(total, 2, 3), "s": torch.arange(total).float()}
        zl, cl = par.shard_batch(z, cond)
        lo, hi = par.shard_bounds(total, rank, world)
        assert torch.equal(zl, z[lo:hi]) and torch.equal(cl["a"], cond["a"][lo:hi]) and torch.equal(cl["s"], cond["s"][lo:hi])
        # "generate": an image that encodes the global sample index
        imgs = zl[:, :1, None, None].expand(-1, 3, 2, 2).contiguous() + 0.5
        full = par.gather_images(imgs, total)
        want = z[:, :1, None, None].expand(-1, 3, 2, 2) + 0.5
        assert torch.equal(full, want)
        slow = par.max_over_ranks(1.0 + rank)
        assert slow == float(world)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 8), (2, 5), (3, 7)])
def test_shard_and_gather(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 16, 33):
        for w in (1, 2, 3, 8):
            spans = [par.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
