"""Meta-training on the HIP train step: the Reptile outer loop of the reference's train.py:402-534 and the
"fast adaptation" stage of train_only_inner_loop.py:306-318, plus the data-parallel gradient all-reduce a multi-GPU train run needs.

Host orchestration only - every tensor operation is a kernel of libmaster_hip.so (train-mode forward/backward through autograd.py,
FusedAdam, reptile_update) or a collective.

Reptile (train.py:402-534), per outer iteration:   omega <- theta;   for num_inner_updates steps: k = randint(1, max_layers),
out = decoder(StyleTransformer_omega(enc(content), enc(style), k)), loss = custom_loss(...), Adam step on omega;
theta += outer_lr * (omega - theta).   The reference file as written has its loss / backward block de-indented out of the inner
`for` (train.py:469) and the backward/step under an `else` that the default flags never take (:482, :515-517), so with default flags
no gradient step runs; this module implements the algorithm those lines describe (loss, backward and step inside every inner update).
The inner optimizer is created once and keeps its moments across outer iterations, like the reference's (train.py:389-398).
"""
from __future__ import annotations

import copy
import random
from typing import Callable, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor, nn

from .optim import FusedAdam, reptile_update


def set_fast_adaptation(model) -> List[nn.Parameter]:
    """train_only_inner_loop.py:306-318: only the StyleEncoder stays trainable (Swin encoder, StyleDecoder and the image decoder are
    frozen).  Returns the trainable parameters (hand them to the optimizer)."""
    for p in model.swin_encoder.parameters():
        p.requires_grad = False
    for p in model.style_transformer.decoder.parameters():
        p.requires_grad = False
    for p in model.style_transformer.encoder.parameters():
        p.requires_grad = True
    for p in model.decoder.parameters():
        p.requires_grad = False
    return [p for p in model.style_transformer.encoder.parameters()]


def allreduce_gradients(params: Iterable[nn.Parameter], average: bool = True) -> None:
    """Data-parallel training: ONE all-reduce of all gradients as a flat buffer (4.3 M floats = 17 MB for StyleTransformer + decoder,
    SURVEY 8e; one large collective suits the per-link-bound xGMI ring better than 69 small ones)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    # the flat buffer covers EVERY trainable parameter, in the caller's order, whether or not this rank produced a gradient for it
    # (a parameter unused on one rank contributes zeros), followed by one "has a gradient" flag per parameter: the buffer has the same
    # size on every rank by construction, so ranks that disagree about which gradients exist can neither hang the collective nor
    # mis-align it.  A parameter without a gradient on ANY rank keeps grad = None (the optimizer skips it, like torch).
    ps = [p for p in params if p.requires_grad]
    if not ps:
        return
    dev, dt = ps[0].device, ps[0].dtype
    has_local = [p.grad is not None for p in ps]
    sizes = [p.numel() for p in ps]
    total = sum(sizes)
    flat = torch.empty((total + len(ps),), dtype=dt, device=dev)
    off = 0
    for p, h, n in zip(ps, has_local, sizes):
        if h:
            flat[off:off + n].copy_(p.grad.reshape(-1))
        else:
            flat[off:off + n].zero_()
        off += n
    flat[total:].copy_(torch.tensor([1.0 if h else 0.0 for h in has_local], dtype=dt), non_blocking=True)
    dist.all_reduce(flat)
    if average:
        flat[:total] /= dist.get_world_size()
    # the flags are read on the HOST only for parameters this rank has no gradient for (r03 read all of them after every all-reduce:
    # a blocking device-to-host copy in every inner step); a rank that produced every gradient never waits here
    missing = [i for i, h in enumerate(has_local) if not h]
    elsewhere = set()
    if missing:
        got = flat[total:][torch.tensor(missing, device=dev)].tolist()
        elsewhere = {i for i, c in zip(missing, got) if c > 0.0}
    off = 0
    for i, (p, h, n) in enumerate(zip(ps, has_local, sizes)):
        if h:
            p.grad.copy_(flat[off:off + n].view_as(p))
        elif i in elsewhere:
            p.grad = flat[off:off + n].view_as(p).clone()
        off += n


class ReptileTrainer:
    """theta = `model` (MasterStyleTransferModel); omega = private copies of its StyleTransformer and image decoder."""

    def __init__(self, model, loss_fn, inner_lr: float = 1e-4, outer_lr: float = 1e-4, num_inner_updates: int = 4, max_layers: int = 4):
        self.model, self.loss_fn = model, loss_fn
        self.outer_lr, self.num_inner_updates, self.max_layers = outer_lr, num_inner_updates, max_layers
        for p in model.swin_encoder.parameters():                     # freeze_encoder (the reference's default)
            p.requires_grad = False
        for m in list(model.style_transformer.modules()) + list(model.decoder.modules()):
            m.__dict__.pop("_derived_cache", None)                    # device copies derived from theta's weights: not part of omega
        self.omega_st = copy.deepcopy(model.style_transformer).train()
        self.omega_dec = copy.deepcopy(model.decoder).train()
        self.optimizer = FusedAdam(list(self.omega_st.parameters()) + list(self.omega_dec.parameters()), lr=inner_lr)     # train.py:398
        self.on_inner_step: Optional[Callable[[int, Tensor], None]] = None       # hook(inner index, total loss) after backward, before step

    def _omega_forward(self, content: Tensor, style: Tensor, k: int) -> Tensor:
        from . import train_forward as TF
        B = content.shape[0]
        enc = self.model.swin_encoder
        was = enc.training
        enc.train(True)                                               # the reference's model is in train mode (StochasticDepth active)
        try:
            with torch.no_grad():
                x, _, h, w = enc.run(content, style)
        finally:
            enc.train(was)                                            # also when an explicit mask list runs out mid-call
        fc, fs = x[: B * h * w], x[B * h * w:]
        fcs = TF.style_transformer(self.omega_st, fc, fs, k, B, h, w, TF.sd_masks)
        return TF.image_decoder(self.omega_dec, fcs.view(B, h, w, -1))

    def outer_step(self, style_batch: Tensor, content_batches: Sequence[Tensor], layer_counts: Optional[Sequence[int]] = None) -> List[float]:
        """one outer iteration; content_batches: num_inner_updates batches.  Returns the inner losses."""
        assert len(content_batches) == self.num_inner_updates
        self.omega_st.load_state_dict(self.model.style_transformer.state_dict())       # omega <- theta  (train.py:432-433)
        self.omega_dec.load_state_dict(self.model.decoder.state_dict())
        losses = []
        for i, content in enumerate(content_batches):
            k = layer_counts[i] if layer_counts is not None else random.randint(1, self.max_layers)      # train.py:448
            out = self._omega_forward(content, style_batch, k)
            total = self.loss_fn(content, style_batch, out)
            self.optimizer.zero_grad()
            total.backward()
            if self.on_inner_step is not None:
                self.on_inner_step(i, total)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:      # data-parallel meta-training
                allreduce_gradients(list(self.omega_st.parameters()) + list(self.omega_dec.parameters()))
            self.optimizer.step()
            losses.append(float(total.detach()))
        # theta += outer_lr * (omega - theta): every tensor of both modules in one launch  (train.py:524-534)
        theta = list(self.model.style_transformer.parameters()) + list(self.model.decoder.parameters())
        omega = list(self.omega_st.parameters()) + list(self.omega_dec.parameters())
        reptile_update([p.data for p in theta], [p.data for p in omega], self.outer_lr)
        return losses
