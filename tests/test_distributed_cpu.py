"""CPU, world_size 2 over gloo: the data-parallel driver shards the batch contiguously and the single
all-gather reassembles outputs in batch order (the N>1 path of bench.py / SURVEY.md section 8e)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mastermetastyletransfer_amd.distributed import AsyncImageGather, ShardedStylizer, all_gather_images, shard_range


def test_shard_ranges_cover_batch():
    for B in (1, 2, 7, 32, 33, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_forward(c, s, k):
    # any per-sample function stands in for the model here: the test is about sharding + gather order
    return (c * 2 + s.flip(-1)) * k


def _worker(rank, world, port, batch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        c, s = torch.randn(batch, 3, 8, 8, generator=g), torch.randn(batch, 3, 8, 8, generator=g)
        out = ShardedStylizer(_fake_forward, rank, world)(c, s, 3)
        ok = torch.equal(out, _fake_forward(c, s, 3))
        if batch % world == 0:
            # the asynchronous gather used by bench.py: several steps in flight, same result as the blocking one
            from mastermetastyletransfer_amd.distributed import shard_range as sr
            lo, hi = sr(batch, rank, world)
            ag = AsyncImageGather(batch, rank, world, depth=2)
            outs = [ag.submit(_fake_forward(c[lo:hi], s[lo:hi], kk)) for kk in (1, 2)]
            last = ag.wait()
            ok = ok and torch.equal(last, _fake_forward(c, s, 2)) and torch.equal(outs[0], _fake_forward(c, s, 1))
        q.put((rank, ok, tuple(out.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [8, 5])
def test_world_size_2_gloo(batch):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(ok and shape == (batch, 3, 8, 8) for _, ok, shape in res)


def test_world_size_1_is_identity():
    x = torch.randn(3, 3, 4, 4)
    assert all_gather_images(x, 3, 0, 1) is x
