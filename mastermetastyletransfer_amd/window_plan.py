"""Host-side window geometry for the attention kernel: the reference's pad -> cyclic roll -> window
partition (codes/style_transformer.py:75-111) and its shift-mask regions (:134-147) as two small
per-window tables, cached per (H, W, window, shift, device).  Pure index math, no activations."""
from __future__ import annotations

from functools import lru_cache
from typing import Optional, Tuple

import numpy as np
import torch

SLOTS = 64  # the kernels tile a window as 64 slots; N = wh*ww <= 64 of them are used


class WindowPlan:
    __slots__ = ("src", "region", "nW", "N", "pH", "pW", "n_pad", "shift", "pad_ord")

    def __init__(self, src, region, nW, N, pH, pW, n_pad, shift, pad_ord=None):
        self.src, self.region, self.nW, self.N = src, region, nW, N
        self.pH, self.pW, self.n_pad, self.shift = pH, pW, n_pad, shift
        self.pad_ord = pad_ord      # int32 [nW,64]: ordinal of a zero-padded slot among the image's n_pad padded slots (backward only)


def _plan_numpy(H: int, W: int, ws: Tuple[int, int], shift: Tuple[int, int]):
    wh, ww = ws
    N = wh * ww
    if N > SLOTS:
        raise NotImplementedError(f"window {ws} has {N} tokens; the HIP attention kernel tiles at most {SLOTS} "
                                  "(the reference's CLI default and this build's target is 7x7)")
    pH, pW = -(-H // wh) * wh, -(-W // ww) * ww
    sh = 0 if wh >= pH else shift[0]           # no shift along an axis one window already covers
    sw = 0 if ww >= pW else shift[1]
    rows = (np.arange(pH) + sh) % pH           # roll by -s: rolled[i] = padded[(i + s) % n]
    cols = (np.arange(pW) + sw) % pW
    real = (rows[:, None] < H) & (cols[None, :] < W)
    flat = np.where(real, rows[:, None] * W + cols[None, :], -1).astype(np.int32)
    nWh, nWw = pH // wh, pW // ww
    nW = nWh * nWw

    def windows(a):
        return a.reshape(nWh, wh, nWw, ww).transpose(0, 2, 1, 3).reshape(nW, N)

    src = np.full((nW, SLOTS), -1, np.int32)
    src[:, :N] = windows(flat)
    region = None
    if sh + sw > 0:
        # same literal slice assignment as torchvision / the reference, so that the one-axis-shift
        # corner case (a slice ending at -0) labels regions identically
        m = np.zeros((pH, pW), np.uint8)
        c = 0
        for a in ((0, -wh), (-wh, -sh), (-sh, None)):
            for b in ((0, -ww), (-ww, -sw), (-sw, None)):
                m[a[0]:a[1], b[0]:b[1]] = c
                c += 1
        region = np.zeros((nW, SLOTS), np.uint8)
        region[:, :N] = windows(m)
    return src, region, nW, N, pH, pW, pH * pW - H * W, (sh, sw)


@lru_cache(maxsize=128)
def _plan_cached(H, W, ws, shift, device_str):
    src, region, nW, N, pH, pW, n_pad, eff = _plan_numpy(H, W, ws, shift)
    dev = torch.device(device_str)
    pad_ord = np.full(src.shape, -1, np.int32)
    is_pad = np.zeros(src.shape, bool)
    is_pad[:, :N] = src[:, :N] < 0
    pad_ord[is_pad] = np.arange(int(is_pad.sum()), dtype=np.int32)
    assert int(is_pad.sum()) == n_pad
    return WindowPlan(torch.from_numpy(src).to(dev), None if region is None else torch.from_numpy(region).to(dev),
                      nW, N, pH, pW, n_pad, eff, torch.from_numpy(pad_ord).to(dev))


def window_plan(H: int, W: int, ws, shift, device) -> WindowPlan:
    return _plan_cached(int(H), int(W), (int(ws[0]), int(ws[1])), (int(shift[0]), int(shift[1])), str(device))


def relative_position_index(ws) -> torch.Tensor:
    """int64 [N*N] buffer with the reference's values (codes/style_transformer.py:227-239)."""
    wh, ww = int(ws[0]), int(ws[1])
    ys, xs = np.divmod(np.arange(wh * ww), ww)
    dy = ys[:, None] - ys[None, :] + wh - 1
    dx = xs[:, None] - xs[None, :] + ww - 1
    return torch.from_numpy((dy * (2 * ww - 1) + dx).reshape(-1).astype(np.int64))
