"""TEST INFRASTRUCTURE - CPU oracle of one training step of the reference (train_only_inner_loop.py:519-575): train-mode forward
(StochasticDepth row masks given explicitly), perceptual loss (codes/loss.py:201-262), autograd backward, Adam.

    k = num_layers;  out = decoder(StyleTransformer(Fc, Fs, k));  total = content + lambda * style;  total.backward();  Adam.step()

The Swin encoder is frozen (train_only_inner_loop.py:254-257) and VGG-19 is frozen (loss.py:173-175): gradients exist for the
StyleTransformer (3 422 424 parameters) and the image decoder (877 091).  The arithmetic is the pure-torch restatement of
master_oracle.py / loss_oracle.py run under autograd, so its backward is torch's own.

Pinned by tests/golden/make_golden_train.py: the unmodified reference modules (StyleTransformer, Decoder, custom_loss) run the same
step with the same masks in this container; loss and every parameter gradient agreed to <= 1e-6 relative.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import loss_oracle as L
from . import master_oracle as O

Tensor = torch.Tensor


def train_forward_loss(fc: Tensor, fs: Tensor, content: Tensor, style: Tensor, k: int, st_sd: Dict[str, Tensor],
                       dec_sd: Dict[str, Tensor], vgg_sd: Dict[str, Tensor], masks: Optional[Sequence[Tensor]],
                       lambda_value: float = 10.0, distance_content: str = "euclidian", distance_style: str = "euclidian", cfg=None):
    """features Fc, Fs (NHWC, from the frozen encoder) -> (total, content, style, output image) with grads attached"""
    sd = O.SDMasks(masks) if masks is not None else None
    fcs = O.style_transformer(fc, fs, k, st_sd, cfg, sd).permute(0, 3, 1, 2)
    out = O.image_decoder(fcs, dec_sd)
    total, c, s = L.overall_loss(content, style, out, vgg_sd, lambda_value, distance_content, distance_style)
    return total, c, s, out


def train_step_grads(content: Tensor, style: Tensor, k: int, swin_sd, st_sd, dec_sd, vgg_sd, masks=None, lambda_value: float = 10.0,
                     distance_content: str = "euclidian", distance_style: str = "euclidian", cfg=None, trainable=None, swin_masks=None,
                     swin_heads=(4, 8)):
    """One forward + backward.  Returns (losses dict, grads dict keyed 'style_transformer.<name>' / 'decoder.<name>').
    `trainable`: optional predicate on the key (fast adaptation trains only 'style_transformer.encoder.*',
    train_only_inner_loop.py:306-318).  `swin_masks`: the 2 x O.SWIN_SD_CALLS StochasticDepth masks of the two frozen-encoder calls
    (content's six, then style's six), None = identity.  The computation runs in the dtype of `content` (float64: the reference
    the fp32 implementations are measured against)."""
    dt = content.dtype
    cast = lambda sd: {n: (t.to(dt) if t.is_floating_point() else t) for n, t in sd.items()}
    swin_sd, st_sd, dec_sd, vgg_sd = cast(swin_sd), cast(st_sd), cast(dec_sd), cast(vgg_sd)
    style = style.to(dt)
    if masks is not None:
        masks = [m.to(dt) for m in masks]
    if swin_masks is not None:
        swin_masks = [m.to(dt) for m in swin_masks]
    with torch.no_grad():
        sdc = O.SDMasks(swin_masks[:O.SWIN_SD_CALLS]) if swin_masks is not None else None
        sds = O.SDMasks(swin_masks[O.SWIN_SD_CALLS:]) if swin_masks is not None else None
        fc = O.swin_encoder(content, swin_sd, heads=swin_heads, sd=sdc)          # (3, 6) heads: swin_T / swin_S
        fs = O.swin_encoder(style, swin_sd, heads=swin_heads, sd=sds)

    def leaf(prefix, sd):
        out = {}
        for n, t in sd.items():
            t = t.detach().clone()
            if t.is_floating_point() and (trainable is None or trainable(prefix + n)):
                t.requires_grad_(True)
            out[n] = t
        return out
    st, dec = leaf("style_transformer.", st_sd), leaf("decoder.", dec_sd)
    total, c, s, out = train_forward_loss(fc, fs, content, style, k, st, dec, vgg_sd, masks, lambda_value, distance_content,
                                          distance_style, cfg)
    total.backward()
    grads = {}
    for prefix, sd in (("style_transformer.", st), ("decoder.", dec)):
        for n, t in sd.items():
            if t.requires_grad:
                grads[prefix + n] = t.grad if t.grad is not None else torch.zeros_like(t)
    return {"total": total.detach(), "content": c.detach(), "style": s.detach(), "output": out.detach()}, grads


def adam_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], state: Dict[str, Dict[str, Tensor]], lr: float, step: int,
              betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0) -> None:
    """torch.optim.Adam defaults (train_only_inner_loop.py:468-478 passes only lr): in-place on `params`; `step` is 1-based."""
    b1, b2 = betas
    for n, g in grads.items():
        p = params[n]
        if weight_decay:
            g = g + weight_decay * p
        st = state.setdefault(n, {"m": torch.zeros_like(p), "v": torch.zeros_like(p)})
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (st["v"].sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(st["m"], denom, value=-lr / bc1)
