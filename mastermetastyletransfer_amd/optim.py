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
    train_only_inner_loop.py:254-257, :306-318).

    capturable=True keeps the step count and the hyper-parameters of every group in device memory (master_adam_step_dev) so that
    `step()` can be captured in a hipGraph with the rest of the train step (graph.GraphedTrainStep): a replay then advances the
    counter itself.  `state[p]["step"]` is brought up to date from the device by state_dict() / sync_steps()."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, capturable: bool = False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = bool(capturable)
        self._tables: Dict[int, _PointerTable] = {}
        self._dev: Dict[int, dict] = {}                  # capturable: group index -> {"step": int32[1], "hyper": float32[5], "host": tuple}
        self._keep: List[Tensor] = []                    # pointer tables a captured graph reads at every replay
        self._spare: Dict[int, Tensor] = {}              # one pinned table per group, ready for the next capture

    # ---- capturable mode ------------------------------------------------------------------------------------------------------------
    def _group_dev(self, gi: int, group, dev, steps_done: int) -> dict:
        d = self._dev.get(gi)
        if d is None:
            d = self._dev[gi] = {"step": torch.tensor([steps_done], dtype=torch.int32, device=dev),
                                 "hyper": torch.zeros(5, dtype=torch.float32, device=dev), "host": None}
        return d

    def push_hypers(self) -> None:
        """copy lr / betas / eps / weight_decay of every group to the device if they changed on the host (call it outside a capture;
        GraphedTrainStep does before every replay, so a learning-rate schedule keeps working)"""
        for gi, group in enumerate(self.param_groups):
            d = self._dev.get(gi)
            if d is None:
                continue
            b1, b2 = group["betas"]
            host = (float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]))
            if host != d["host"]:
                d["hyper"].copy_(torch.tensor(host, dtype=torch.float32))
                d["host"] = host

    def sync_steps(self) -> None:
        """state[p]["step"] <- the device counters (one host read per group)"""
        for gi, group in enumerate(self.param_groups):
            d = self._dev.get(gi)
            if d is None:
                continue
            n = int(d["step"].item())
            for p in group["params"]:
                if self.state.get(p):
                    self.state[p]["step"] = n

    def state_dict(self):
        if self.capturable:
            self.sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        # a captured graph (GraphedTrainStep) holds the ADDRESSES of the moment buffers: torch's load_state_dict replaces the tensors, and
        # the graph would go on updating the old ones.  With capturable=True the loaded values are therefore copied INTO the existing
        # buffers (same shape and device), which stay the optimizer's state.
        old = {p: (st.get("exp_avg"), st.get("exp_avg_sq")) for p, st in self.state.items()} if self.capturable else {}
        super().load_state_dict(state_dict)
        for p_, (m_old, v_old) in old.items():
            st = self.state.get(p_)
            if not st:
                continue
            for name, keep in (("exp_avg", m_old), ("exp_avg_sq", v_old)):
                new = st.get(name)
                if keep is not None and new is not None and keep is not new and keep.shape == new.shape and keep.device == new.device:
                    keep.copy_(new)
                    st[name] = keep
        for gi, group in enumerate(self.param_groups):
            d = self._dev.get(gi)
            steps = {int(self.state[p]["step"]) for p in group["params"] if self.state.get(p)}
            if self.capturable and steps:
                # a resumed group HAS stepped: the late-joiner guard of step() must see that (its counter lives outside the state dict)
                self.__dict__.setdefault("_cap_stepped", {})[gi] = max(self.__dict__.get("_cap_stepped", {}).get(gi, 0), max(steps))
            if d is not None and steps:
                d["step"].fill_(steps.pop())

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = load_library()
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if capturing and not self.capturable:
            raise ops.MasterHipError("FusedAdam(capturable=True) is needed to capture step() in a graph")
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
                    if capturing:
                        raise ops.MasterHipError("FusedAdam: run one eager step before capturing (the moment buffers are created there)")
                    if self.capturable and self.__dict__.setdefault("_cap_stepped", {}).get(gi, 0) > 0:
                        # the group's ONE device counter drives every tensor's bias correction: a tensor joining at step t > 1 would be
                        # corrected for step t, not 1 (torch.optim.Adam keeps a step per tensor; the non-capturable mode raises below)
                        raise ops.MasterHipError("FusedAdam(capturable=True): a parameter received its first gradient after its param group had "
                                                 "already stepped; the tensors of a param group must share their step count (put late "
                                                 "parameters - e.g. those enabled by set_fast_adaptation - in a param group of their own)")
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not self.capturable:
                    st["step"] += 1
                recs.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), g))
            if self.capturable and not capturing:
                self.__dict__.setdefault("_cap_stepped", {})[gi] = self._cap_stepped.get(gi, 0) + 1
            steps = {self.state[p]["step"] for p in ps}
            if len(steps) != 1:
                raise ops.MasterHipError("FusedAdam: the tensors of a param group must share their step count")
            key = tuple(r[:5] for r in recs)
            b1, b2 = group["betas"]
            with torch.cuda.device(dev):
                if capturing:
                    # the graph re-reads this table at every replay: its own pinned rows (set aside by the last eager step - no host
                    # allocation inside a capture) and device copy, kept alive with the optimizer
                    rows, nblocks = _rows([r[:5] for r in recs])
                    host = self._spare.pop(gi, None)
                    if host is None or host.shape[0] != len(recs):
                        raise ops.MasterHipError("FusedAdam: run one eager step before capturing")
                    host.copy_(torch.tensor(rows, dtype=torch.int64))
                    table = torch.empty((len(recs), 6), dtype=torch.int64, device=dev)
                    table.copy_(host, non_blocking=True)
                    self._keep += [host, table]
                else:
                    if self.capturable and (gi not in self._spare or self._spare[gi].shape[0] != len(recs)):
                        self._spare[gi] = torch.empty((len(recs), 6), dtype=torch.int64).pin_memory()
                    tab = self._tables.get(gi)
                    if tab is None or tab.dev.shape[0] != len(recs) or tab.dev.device != dev:
                        tab = self._tables[gi] = _PointerTable(len(recs), dev)
                    tab.refresh(key, [r[:5] for r in recs])
                    table, nblocks = tab.dev, tab.nblocks
                if self.capturable:
                    d = self._group_dev(gi, group, dev, int(steps.pop()))
                    if not capturing:
                        self.push_hypers()
                    elif d["host"] is None:
                        raise ops.MasterHipError("FusedAdam: run one eager step before capturing")
                    _check(lib.master_adam_step_dev(_p(table), len(recs), nblocks, _p(d["hyper"]), _p(d["step"]), _stream()), "master_adam_step_dev")
                else:
                    _check(lib.master_adam_step(_p(table), len(recs), nblocks, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
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
