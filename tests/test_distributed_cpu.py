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
            # the in-place form bench.py uses at N > 1: a producer with TWO alternating output buffers (two captured graphs); the
            # gather reads them in place, release_source() fences a buffer before it is overwritten, results land in caller-owned
            # destinations
            ag2 = AsyncImageGather(batch, rank, world, depth=2)
            bufs = [torch.empty_like(c[lo:hi]) for _ in range(2)]
            dests = [torch.empty_like(c) for _ in range(5)]
            for i in range(5):
                ag2.release_source(bufs[i % 2])
                bufs[i % 2].copy_(_fake_forward(c[lo:hi], s[lo:hi], i + 1))
                got = ag2.submit(bufs[i % 2], out=dests[i])
                ok = ok and got is dests[i]
            ag2.wait()
            ok = ok and all(torch.equal(dests[i], _fake_forward(c, s, i + 1)) for i in range(5)) and not ag2.pending
        q.put((rank, ok, tuple(out.shape)))
    finally:
        dist.destroy_process_group()


def _oracle_forward(c, s, k):
    """a REAL forward per rank: the CPU oracle of the whole path (Swin-B[:2 stages] -> StyleTransformer -> decoder) with seeded
    weights - on the GPU box the same driver code shards MasterStyleTransferModel.forward"""
    from oracle import master_oracle as O
    from oracle.seeded import seeded_state_dict
    sds = (seeded_state_dict(O.swin_shapes(), 1), seeded_state_dict(O.style_transformer_shapes(), 2), seeded_state_dict(O.decoder_shapes(), 3))
    with torch.no_grad():
        return O.full_forward(c, s, k, *sds)


def _worker_real(rank, world, port, batch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.seeded import seeded_input
        c = seeded_input("dist_c", (batch, 3, 32, 32), 5, kind="image"); s = seeded_input("dist_s", (batch, 3, 32, 32), 6, kind="image")
        out = ShardedStylizer(_oracle_forward, rank, world)(c, s, 1)
        # rank ids through the same collective the bench uses to prove every rank is connected
        ids = torch.empty((world,), dtype=torch.int64)
        dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int64))
        lo, hi = shard_range(batch, rank, world)
        mine = _oracle_forward(c[lo:hi], s[lo:hi], 1)
        whole = _oracle_forward(c, s, 1) if rank == 0 else None
        q.put((rank, tuple(out.shape), ids.tolist(), bool(torch.equal(out[lo:hi], mine)),
               float((out - whole).abs().max()) if whole is not None else 0.0, out.numpy().tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [4, 3])
def test_world_size_2_gloo_real_forward(batch):
    """every rank ends with the SAME gathered batch, bit for bit; each shard inside it is bit-identical to that rank's own forward; and
    the gathered batch equals the single-process forward of the whole batch (per-sample ops: only the BLAS blocking of a different batch size differs: <= 1e-4 on outputs of magnitude 3)"""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    for rank, shape, ids, shard_equal, err, _ in res:
        assert shape == (batch, 3, 32, 32) and ids == [0, 1] and shard_equal and err <= 1e-4, (rank, shape, ids, shard_equal, err)
    assert res[0][5] == res[1][5], "ranks disagree on the gathered batch"


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


@pytest.mark.parametrize("batch", [32, 13])
def test_world_size_8_gloo(batch):
    """the node size the driver's scaling run uses (8 ranks): sharding, the asynchronous gather with two steps in flight and the in-place
    two-buffer form, even (32 = 4 per rank) and ragged (13) batches - eight CPU processes over gloo"""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, batch, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(8))
    assert all(ok and shape == (batch, 3, 8, 8) for _, ok, shape in res)


def test_world_size_1_is_identity():
    x = torch.randn(3, 3, 4, 4)
    assert all_gather_images(x, 3, 0, 1) is x


def _worker_grads(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mastermetastyletransfer_amd.meta import allreduce_gradients
        g = torch.Generator().manual_seed(rank)
        ps = [torch.nn.Parameter(torch.zeros(s)) for s in ((5, 3), (7,), (2, 2, 2))] + [torch.nn.Parameter(torch.zeros(4)),
                                                                                       torch.nn.Parameter(torch.zeros(3))]
        for p in ps[:3]:
            p.grad = torch.randn(p.shape, generator=g)
        if rank == 1:
            ps[4].grad = torch.full((3,), 2.0)        # a gradient on ONE rank only (a parameter the other rank's step did not use)
        allreduce_gradients(ps)                       # ps[3] has no gradient anywhere: stays None on every rank
        q.put((rank, [p.grad.clone() for p in ps[:3]], ps[3].grad is None, ps[4].grad.clone()))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world_2():
    """the data-parallel train step's one collective: a flat all-reduce that averages every gradient over the ranks"""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_grads, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = []
    for shape in ((5, 3), (7,), (2, 2, 2)):
        want.append(sum(torch.randn(shape, generator=torch.Generator().manual_seed(r)) for r in range(2)) / 2)
    # generators advance tensor by tensor: rebuild in the same order
    gens = [torch.Generator().manual_seed(r) for r in range(2)]
    want = [sum(torch.randn(shape, generator=gens[r]) for r in range(2)) / 2 for shape in ((5, 3), (7,), (2, 2, 2))]
    for rank, grads, none_kept, lonely in res:
        assert none_kept
        assert torch.equal(lonely, torch.full((3,), 1.0))         # (0 + 2) / 2 on both ranks: buffers aligned although the ranks disagreed
        for a, b in zip(grads, want):
            assert torch.allclose(a, b, atol=1e-7)
