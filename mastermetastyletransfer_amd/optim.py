"""Optimizer steps on the device: multi-tensor Adam (torch.optim.Adam semantics - the reference builds
`optim.Adam([{'params': style_transformer}, {'params': decoder}], lr=inner_lr)`, train_only_inner_loop.py:468-478, and calls
zero_grad / backward / step at :573-575) and the Reptile interpolation of the meta-training outer loop (train.py:524-534).

One kernel launch per step for all parameter tensors: a device table of (param, grad, exp_avg, exp_avg_sq, numel, first_block)
records is rebuilt only when a tensor's storage changes.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Iterable, List, Optional, Sequence

import torch
from torch import Tensor

from . import ops
from .ops import _check, _p, _stream, load_library


def _rows(recs: Sequence[Sequence[int]]) -> tuple:
    """recs: (p_ptr, g_ptr, m_ptr, v_ptr, numel) per tensor -> (host rows [n][6], total blocks)"""
    chunk = load_library().master_adam_chunk()
    rows, first = [], 0
    for p, g, m, v, n in recs:
        rows.append([p, g, m, v, n, first])
        first += (n + chunk - 1) // chunk
    return rows, first


def _table(recs: Sequence[Sequence[int]], device) -> tuple:
    """-> (device int64 table [n,6], total blocks); a one-off, synchronous upload (reptile_update: once per outer iteration)"""
    rows, first = _rows(recs)
    return torch.tensor(rows, dtype=torch.int64).to(device), first


class _PointerTable:
    """The device table of one param group, refreshed WITHOUT a host-device synchronisation: `zero_grad(set_to_none=True)` (every
    caller's default) makes backward allocate fresh gradient tensors, so the gradient pointers may change every step.  The rows are
    written into one of two pinned host buffers and copied with a stream-ordered non-blocking copy; the event recorded after the copy
    guards the buffer's reuse two refreshes later (it has long completed by then)."""

    def __init__(self, n: int, device):
        self.dev = torch.empty((n, 6), dtype=torch.int64, device=device)
        self.host = [torch.empty((n, 6), dtype=torch.int64).pin_memory() for _ in range(2)]
        self.done = [None, None]
        self.turn = 0
        self.key, self.nblocks = None, 0

    def refresh(self, key, recs):
        if key == self.key:
            return
        rows, self.nblocks = _rows(recs)
        i = self.turn
        self.turn ^= 1
        if self.done[i] is not None:
            self.done[i].synchronize()
        self.host[i].copy_(torch.tensor(rows, dtype=torch.int64))
        self.dev.copy_(self.host[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.done[i] = ev
        self.key = key


def _mark_written(tensors: Iterable[Tensor]) -> None:
    """The kernels above write parameters through raw pointers, which torch cannot see: bump every tensor's version counter so that
    caches keyed on it (the modules' derived-weight cache: fused qkv concatenations, packed convolution weights, bf16 copies) notice
    the update and an in-process eval forward after training uses the new weights."""
    for t in tensors:
        torch.autograd.graph.increment_version(t)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) with the update of every tensor of a param group in one HIP launch
    (master_adam_step).  Parameters without a gradient are skipped, like torch does (frozen encoder / fast adaptation,
    train_only_inner_loop.py:254-257, :306-318)."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables: Dict[int, _PointerTable] = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = load_library()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise ops.MasterHipError("FusedAdam runs on the GPU only (there is no CPU fallback)")
            recs = []
            for p in ps:
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous():
                    raise ops.MasterHipError("FusedAdam: parameters and gradients must be contiguous fp32")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                recs.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), g))
            steps = {self.state[p]["step"] for p in ps}
            if len(steps) != 1:
                raise ops.MasterHipError("FusedAdam: the tensors of a param group must share their step count")
            key = tuple(r[:5] for r in recs)
            b1, b2 = group["betas"]
            with torch.cuda.device(dev):
                tab = self._tables.get(gi)
                if tab is None or tab.dev.shape[0] != len(recs) or tab.dev.device != dev:
                    tab = self._tables[gi] = _PointerTable(len(recs), dev)
                tab.refresh(key, [r[:5] for r in recs])
                _check(lib.master_adam_step(_p(tab.dev), len(recs), tab.nblocks, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                            float(group["weight_decay"]), int(steps.pop()), _stream()), "master_adam_step")
            _mark_written(ps)
        return loss


@torch.no_grad()
def reptile_update(theta: Iterable[Tensor], omega: Iterable[Tensor], eta: float) -> None:
    """theta += eta * (omega - theta) for every tensor pair, one launch (the meta-training outer update, train.py:524-534)"""
    th, om = [t for t in theta], [o for o in omega]
    if len(th) != len(om) or not th:
        raise ValueError("reptile_update: theta and omega must be non-empty and of equal length")
    dev = th[0].device
    recs = []
    for t, o in zip(th, om):
        if t.shape != o.shape or t.dtype != torch.float32 or o.dtype != torch.float32 or not t.is_contiguous() or not o.is_contiguous():
            raise ops.MasterHipError("reptile_update: tensors must be matching contiguous fp32")
        recs.append((t.data_ptr(), o.data_ptr(), 0, 0, t.numel()))
    table, nblocks = _table(recs, dev)
    with torch.cuda.device(dev):
        _check(load_library().master_lerp(_p(table), len(recs), nblocks, float(eta), _stream()), "master_lerp")
    _mark_written(th)
