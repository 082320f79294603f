"""Data-parallel batched inference: one process per GPU, batch sharded contiguously by rank, ONE
collective - an all-gather of the stylized images (RCCL over xGMI; backend "nccl" is RCCL on ROCm).

The path shards trivially: every op is per sample (LayerNorm per token, InstanceNorm per (image,
channel), windows per image), weights (25.7 MB fp32) are replicated, so there is
no data-path collective besides the output gather.  On CPU (tests) the same code runs over gloo.
"""
from __future__ import annotations

import os
from typing import Callable, Tuple

import torch
import torch.distributed as dist


def init_from_env(prefer_gpu: bool = True) -> Tuple[int, int, torch.device]:
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as set by torch.distributed.run."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    use_gpu = prefer_gpu and torch.cuda.device_count() > 0
    # MASTER_DIST_REHEARSAL=1: several ranks share ONE GPU over gloo (collectives staged through the host) - only to rehearse the
    # N > 1 driver code on a one-GPU box; RCCL refuses two ranks on one device.  Never set on a real multi-GPU run.
    rehearsal = os.environ.get("MASTER_DIST_REHEARSAL") == "1"
    if use_gpu and rehearsal:
        local = local % torch.cuda.device_count()
    device = torch.device(f"cuda:{local}") if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if use_gpu and not rehearsal:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, device


def _host_staged() -> bool:
    """gloo cannot all-gather device tensors: in rehearsal mode the collectives run on host copies"""
    return dist.is_initialized() and dist.get_backend() == "gloo"


def all_gather_tensor(out: torch.Tensor, local: torch.Tensor, async_op: bool = False):
    """dist.all_gather_into_tensor; device tensors over gloo (rehearsal) go through the host, synchronously"""
    if _host_staged() and local.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, local.cpu())
        out.copy_(o)

        class _Done:
            def wait(self):
                return True
        return _Done() if async_op else None
    return dist.all_gather_into_tensor(out, local, async_op=async_op)


def shard_range(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous shard [lo, hi) of a batch; the first batch % world ranks take one extra sample"""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_images(local: torch.Tensor, batch: int, rank: int, world: int) -> torch.Tensor:
    """[b_local, 3, H, W] on every rank -> [batch, 3, H, W] on every rank (rank order = batch order)."""
    if world == 1:
        return local
    local = local.contiguous()
    if batch % world == 0:
        out = local.new_empty((batch,) + tuple(local.shape[1:]))
        all_gather_tensor(out, local)
        return out
    # ragged tail: pad every shard to the largest one, gather, then cut the padding out
    cap = -(-batch // world)
    padded = local.new_zeros((cap,) + tuple(local.shape[1:]))
    padded[: local.shape[0]] = local
    out = local.new_empty((world * cap,) + tuple(local.shape[1:]))
    all_gather_tensor(out, padded)
    parts = []
    for r in range(world):
        lo, hi = shard_range(batch, r, world)
        parts.append(out[r * cap: r * cap + (hi - lo)])
    return torch.cat(parts, dim=0)


class AsyncImageGather:
    """All-gather of the stylized images that overlaps with the next forward: the collective is enqueued
    asynchronously (RCCL runs it on its own stream) and at most `depth` gathers are in flight; wait() drains them.
    Results are identical to all_gather_images; only the point where the host/stream waits moves.

    The collective reads `local` IN PLACE (no copy is made of a contiguous shard).  A producer that reuses its output buffers - two
    captured graphs writing alternating static buffers, graph.GraphedStylizer(copies=2) - calls release_source(buffer) before it
    overwrites one: the stream then waits for the gather that still reads it.  Destination buffers come from a pool of depth + 1 and
    are recycled after `depth` further submits; pass `out=` to own the destination instead."""

    def __init__(self, batch: int, rank: int, world: int, depth: int = 2, force_collective: bool = False):
        """force_collective: run the collective even in a one-rank group (where the gather is the identity and is normally skipped) - the
        one-GPU hardware test of the RCCL path: library loading, the collective's own stream, release_source ordering against graph replays"""
        if batch % max(world, 1) != 0:
            raise ValueError("AsyncImageGather needs equal shards (batch % world == 0)")
        self.batch, self.rank, self.world, self.depth = batch, rank, world, depth
        self.force = bool(force_collective)
        self.pending = []          # (work, output buffer, local shard kept alive)
        self.buffers = []

    def submit(self, local: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        if self.world == 1 and not self.force:
            return local
        while len(self.pending) >= self.depth:
            self._retire()
        local = local.contiguous()
        pooled = out is None
        if pooled:
            out = self.buffers.pop() if self.buffers else local.new_empty((self.batch,) + tuple(local.shape[1:]))
        work = all_gather_tensor(out, local, async_op=True)
        self.pending.append((work, out, local, pooled))
        return out                 # valid after wait() (or after `depth` further submits)

    def _retire(self, i: int = 0):
        work, out, _, pooled = self.pending.pop(i)
        work.wait()
        if pooled:
            self.buffers.append(out)   # recycled: callers that keep pooled results must copy them before `depth` more submits

    def release_source(self, buf: torch.Tensor):
        """make the current stream wait for every pending gather that still reads `buf` (call before overwriting it)"""
        i = 0
        while i < len(self.pending):
            if self.pending[i][2].data_ptr() == buf.data_ptr():
                self._retire(i)
            else:
                i += 1

    def wait(self):
        last, last_pooled = None, False
        while self.pending:
            work, out, _, pooled = self.pending.pop(0)
            work.wait()
            if last is not None and last_pooled:
                self.buffers.append(last)
            last, last_pooled = out, pooled
        return last


class ShardedStylizer:
    """stylize(content[B], style[B], k) with the batch split over the ranks of the default group."""

    def __init__(self, forward_fn: Callable[[torch.Tensor, torch.Tensor, int], torch.Tensor], rank: int, world: int):
        self.forward_fn, self.rank, self.world = forward_fn, rank, world

    def __call__(self, content: torch.Tensor, style: torch.Tensor, k: int = 1) -> torch.Tensor:
        B = content.shape[0]
        lo, hi = shard_range(B, self.rank, self.world)
        local = self.forward_fn(content[lo:hi], style[lo:hi], k)
        return all_gather_images(local, B, self.rank, self.world)
