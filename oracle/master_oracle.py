"""CPU oracle for the Master stylization hot path (TEST INFRASTRUCTURE - not product code).

A pure-torch fp32 restatement of what the reference computes on the path
``MasterStyleTransferModel.forward`` (reference codes/full_model.py:214-226):

    swin_b.features[:4] encoder  ->  StyleTransformer (k shared-weight layers)  ->  conv Decoder

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module, and only as the checker / the CPU number reported beside the GPU one.
The product package ``mastermetastyletransfer_amd`` never imports it.

Parity pinning (see DESIGN.md "Oracle"):
  * StyleTransformer + Decoder: pinned against the *imported reference modules* by
    ``tests/golden/make_golden.py`` (run in the build container, where /root/reference is
    mounted); the resulting input/output vectors are committed under ``tests/golden``.
  * Swin-B ``features[:4]``: the arithmetic lives in torchvision (absent from the reference tree and
    from this image, version not pinned by the reference) -> cannot be pinned against torchvision here.
    Since r05 the whole stage is pinned against an INDEPENDENT implementation of the published architecture that is in the image,
    Hugging Face transformers' SwinModel (tests/golden/make_golden_swin_hf.py -> golden_swin_hf.npz: five geometries, agreement 4e-6).
    It is restated from the reference's own modified copy of torchvision's window attention
    (codes/style_transformer.py:37-169, fused-qkv original kept in comments at :203,:260-272) and
    the block layout documented in codes/load_pretrained_weights_to_style_transformer.py:29-60.
    ``make_golden.py`` additionally cross-checks the Swin block restatement against the reference's
    ``shifted_window_attention`` by feeding q=k=v and the split qkv weight.

Everything is functional: weights come in as state dicts that use the reference's key names.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
EPS = 1e-5  # nn.LayerNorm / nn.InstanceNorm2d default (style_transformer.py:340-343, :986)


# --------------------------------------------------------------------------------------
# window geometry (pad -> cyclic roll -> partition), as index tables
# --------------------------------------------------------------------------------------
@lru_cache(maxsize=64)
def window_plan(H: int, W: int, ws: Tuple[int, int], shift: Tuple[int, int]):
    """Index form of reference style_transformer.py:75-111 (pad, roll, partition) and
    :134-147 (shift mask regions).

    Returns (src, region, pH, pW):
      src    int64 [nW, N]  flat index h*W+w of the real token feeding window slot (w, t),
                            or H*W for a zero-padded slot (the caller appends one zero row);
      region int64 [nW, N]  mask region id of every slot, or None when no shift is active
                            (mask[i, j] = 0 if region[i] == region[j] else -100).
    """
    wh, ww = ws
    pH = (H + wh - 1) // wh * wh            # :77-85   pad bottom/right to a multiple of the window
    pW = (W + ww - 1) // ww * ww
    sh = 0 if wh >= pH else shift[0]        # :91-94   no shift when one window covers the axis
    sw = 0 if ww >= pW else shift[1]
    ph = torch.arange(pH)
    pw = torch.arange(pW)
    # torch.roll(x, -s)[i] == x[(i + s) % n]                                         (:97-100)
    sh_src = (ph + sh) % pH
    sw_src = (pw + sw) % pW
    real = (sh_src[:, None] < H) & (sw_src[None, :] < W)
    flat = torch.where(real, sh_src[:, None] * W + sw_src[None, :], torch.full((pH, pW), H * W))
    nWh, nWw = pH // wh, pW // ww
    src = flat.view(nWh, wh, nWw, ww).permute(0, 2, 1, 3).reshape(nWh * nWw, wh * ww)   # :105-111
    region = None
    if sh + sw > 0:                          # :134-147, literal slice assignment (keeps the
        m = torch.zeros(pH, pW, dtype=torch.int64)  # torchvision quirk when only one axis shifts)
        hs = ((0, -wh), (-wh, -sh), (-sh, None))
        wss = ((0, -ww), (-ww, -sw), (-sw, None))
        c = 0
        for a in hs:
            for b in wss:
                m[a[0]:a[1], b[0]:b[1]] = c
                c += 1
        region = m.view(nWh, wh, nWw, ww).permute(0, 2, 1, 3).reshape(nWh * nWw, wh * ww)
    return src, region, pH, pW


def relative_position_index(ws: Sequence[int]) -> Tensor:
    """int64 [N*N] table (style_transformer.py:227-239)."""
    wh, ww = ws
    ch, cw = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
    ch, cw = ch.flatten(), cw.flatten()
    dh = ch[:, None] - ch[None, :] + (wh - 1)
    dw = cw[:, None] - cw[None, :] + (ww - 1)
    return (dh * (2 * ww - 1) + dw).flatten()


def relative_position_bias(table: Tensor, index: Tensor, ws: Sequence[int]) -> Tensor:
    """[heads, N, N] (style_transformer.py:21-28, without the leading unsqueeze)."""
    N = ws[0] * ws[1]
    return table[index].view(N, N, -1).permute(2, 0, 1).contiguous()


def _to_windows(x: Tensor, src: Tensor) -> Tensor:
    """[B,H,W,C] -> [B*nW, N, C]; padded slots are zero rows."""
    B, H, W, C = x.shape
    flat = torch.cat([x.reshape(B, H * W, C), x.new_zeros(B, 1, C)], dim=1)
    return flat[:, src.reshape(-1)].reshape(B * src.shape[0], src.shape[1], C)


def _from_windows(xw: Tensor, src: Tensor, B: int, H: int, W: int) -> Tensor:
    """Inverse of _to_windows for the real tokens (reverse windows, reverse roll, un-pad;
    style_transformer.py:160-168)."""
    C = xw.shape[-1]
    s = src.reshape(-1)
    keep = s < H * W
    out = xw.new_empty(B, H * W, C)
    out[:, s[keep]] = xw.reshape(B, -1, C)[:, keep]
    return out.view(B, H, W, C)


def _heads(x: Tensor, nh: int) -> Tensor:
    Bw, N, C = x.shape
    return x.view(Bw, N, nh, C // nh).permute(0, 2, 1, 3)


def _attn_probs(q: Tensor, k: Tensor, bias: Tensor, region: Optional[Tensor], nh: int) -> Tensor:
    """softmax(scale*q k^T + rel-pos bias + shift mask)  (style_transformer.py:127-152)."""
    hd = q.shape[-1]
    a = (q * hd ** -0.5) @ k.transpose(-2, -1)
    a = a + bias.unsqueeze(0)
    if region is not None:
        nW, N = region.shape
        mask = (region[:, None, :] != region[:, :, None]).to(a.dtype) * -100.0   # [nW, N, N]
        a = (a.view(-1, nW, nh, N, N) + mask[None, :, None]).view(-1, nh, N, N)
    return F.softmax(a, dim=-1)


def shifted_window_attention(xq: Tensor, xk: Tensor, xv: Tensor, p: Dict[str, Tensor], prefix: str,
                             ws: Sequence[int], shift: Sequence[int], nh: int) -> Tensor:
    """reference shifted_window_attention, style_transformer.py:37-169 (three inputs, Wq/Wk/Wv)."""
    B, H, W, C = xq.shape
    src, region, _, _ = window_plan(H, W, tuple(ws), tuple(shift))
    q = F.linear(_to_windows(xq, src), p[prefix + "Wq.weight"], p[prefix + "Wq.bias"])
    k = F.linear(_to_windows(xk, src), p[prefix + "Wk.weight"], p[prefix + "Wk.bias"])
    v = F.linear(_to_windows(xv, src), p[prefix + "Wv.weight"], p[prefix + "Wv.bias"])
    bias = relative_position_bias(p[prefix + "relative_position_bias_table"],
                                  p[prefix + "relative_position_index"], ws)
    a = _attn_probs(_heads(q, nh), _heads(k, nh), bias, region, nh)
    x = (a @ _heads(v, nh)).transpose(1, 2).reshape(q.shape[0], q.shape[1], C)
    x = F.linear(x, p[prefix + "proj.weight"], p[prefix + "proj.bias"])
    return _from_windows(x, src, B, H, W)


def instance_norm_nhwc(x: Tensor) -> Tensor:
    """nn.InstanceNorm2d(affine=False) applied through the NHWC<->NCHW permutes
    (style_transformer.py:1056-1057): per (b, c) over H*W, biased variance, eps 1e-5."""
    B, H, W, C = x.shape
    y = F.instance_norm(x.reshape(B, H * W, C).permute(0, 2, 1), eps=EPS)
    return y.permute(0, 2, 1).reshape(B, H, W, C)


def instance_norm_nhwc_affine(x: Tensor, p: Dict[str, Tensor], key: Optional[str]) -> Tensor:
    """instance_norm_nhwc, then the per-channel affine of nn.InstanceNorm2d(dim, affine=True) when `key` names its parameters
    (decoder_use_instance_norm_with_affine, style_transformer.py:983-984)"""
    y = instance_norm_nhwc(x)
    return y if key is None else y * p[key + "weight"] + p[key + "bias"]


def sigma_mu_attention(xq: Tensor, xk: Tensor, xs: Tensor, xsh: Tensor, p: Dict[str, Tensor],
                       prefix: str, ws: Sequence[int], shift: Sequence[int], nh: int, in_q: Optional[str] = None, in_k: Optional[str] = None,
                       key_in_after_linear: bool = True):
    """reference shifted_window_attention_for_decoder_last_MHA (use_q_proj=False): style_transformer.py:414-611.
    in_q / in_k: state-dict prefixes of affine InstanceNorm parameters (None: affine=False, the default);
    key_in_after_linear: use_Key_instance_norm_after_linear_transformation (default True)."""
    B, H, W, C = xq.shape
    src, region, pH, pW = window_plan(H, W, tuple(ws), tuple(shift))
    q = _to_windows(instance_norm_nhwc_affine(xq, p, in_q), src)     # :468 (second IN of Query), no Wq
    if not key_in_after_linear:
        xk = instance_norm_nhwc_affine(xk, p, in_k)                  # :470-472
    k = F.linear(_to_windows(xk, src), p[prefix + "Wk.weight"], p[prefix + "Wk.bias"])
    vs = F.linear(_to_windows(xs, src), p[prefix + "Wv_scale.weight"], p[prefix + "Wv_scale.bias"])
    vsh = F.linear(_to_windows(xsh, src), p[prefix + "Wv_shift.weight"], p[prefix + "Wv_shift.bias"])
    # :520-530  Key un-windowed, InstanceNorm2d over the whole padded pH x pW map (bias-only rows
    # of the padding take part in the statistics), re-windowed.  Statistics do not depend on order.
    nW, N = src.shape
    if key_in_after_linear:
        k = F.instance_norm(k.reshape(B, nW * N, C).permute(0, 2, 1), eps=EPS).permute(0, 2, 1)
        if in_k is not None:
            k = k * p[in_k + "weight"] + p[in_k + "bias"]
        k = k.reshape(B * nW, N, C)
    bias = relative_position_bias(p[prefix + "relative_position_bias_table"],
                                  p[prefix + "relative_position_index"], ws)
    a = _attn_probs(_heads(q, nh), _heads(k, nh), bias, region, nh)
    outs = []
    for v in (vs, vsh):                                               # :575-607, shared proj
        x = (a @ _heads(v, nh)).transpose(1, 2).reshape(B * nW, N, C)
        x = F.linear(x, p[prefix + "proj.weight"], p[prefix + "proj.bias"])
        outs.append(_from_windows(x, src, B, H, W))
    return outs[0], outs[1]


def mlp(x: Tensor, p: Dict[str, Tensor], prefix: str) -> Tensor:
    """torchvision.ops.MLP = Linear, GELU(erf), Dropout(0), Linear, Dropout(0); keys .0.* / .3.*
    (call sites style_transformer.py:366, :839-841, :991)."""
    h = F.gelu(F.linear(x, p[prefix + "0.weight"], p[prefix + "0.bias"]))
    return F.linear(h, p[prefix + "3.weight"], p[prefix + "3.bias"])


# --------------------------------------------------------------------------------------
# StyleTransformer.  Eval mode: StochasticDepth is the identity (sd=None).  Train mode: `sd` is an
# SDMasks object that multiplies every residual branch by the next per-sample row mask, in the order
# the reference's modules call torchvision StochasticDepth(p, "row") (style_transformer.py:390-397,
# :865-882, :1125): noise = bernoulli(1-p) / (1-p), one value per sample.
# --------------------------------------------------------------------------------------
class SDMasks:
    """Explicit StochasticDepth row masks, consumed in call order (the GPU and CPU RNG streams differ, so
    parity tests draw the masks once and hand the same list to the oracle and to the HIP path)."""

    def __init__(self, masks: Sequence[Tensor]):
        self.masks, self.i = list(masks), 0

    def __call__(self, x: Tensor) -> Tensor:
        m = self.masks[self.i]
        self.i += 1
        return x * m.to(x.dtype).view(-1, *([1] * (x.dim() - 1)))

    @staticmethod
    def draw(count: int, batch: int, p: float, generator=None) -> List[Tensor]:
        """torchvision.ops.stochastic_depth row mode: noise = empty([B]).bernoulli_(1-p).div_(1-p)"""
        keep = 1.0 - p
        return [torch.empty(batch).bernoulli_(keep, generator=generator).div_(keep) for _ in range(count)]


SD_CALLS_PER_LAYER = 9     # encoder: attn/MLP x (Key, Scale, Shift); decoder: self-attn, its MLP, last_MLP


def _sd(sd: Optional["SDMasks"], x: Tensor) -> Tensor:
    return x if sd is None else sd(x)


def style_encoder(key: Tensor, scale: Tensor, shift_t: Tensor, p: Dict[str, Tensor], cfg, sd=None) -> Tuple[Tensor, Tensor, Tensor]:
    """StyleEncoder.forward, style_transformer.py:855-912 (both orderings)."""
    a = "encoder.shared_MHA_without_MLP.attn."
    ws, sh, nh = cfg["encoder_window_size"], cfg["encoder_shift_size"], cfg["encoder_num_heads"]

    def attn(q, k, v):
        return shifted_window_attention(q, k, v, p, a, ws, sh, nh)

    def do_key(key):
        key = key + _sd(sd, attn(key, key, key))                       # residual = input_q (:383-384), SD :395
        return key + _sd(sd, mlp(key, p, "encoder.encoder_MLP_Key."))  # :865

    def do_val(key, val, name):
        val = val + _sd(sd, attn(key, key, val))                       # residual = input_v (:385-386)
        return val + _sd(sd, mlp(val, p, f"encoder.encoder_MLP_{name}."))

    if cfg.get("encoder_if_use_processed_Key_in_Scale_and_Shift_calculation", True):
        key = do_key(key)
        scale = do_val(key, scale, "Scale")
        shift_t = do_val(key, shift_t, "Shift")
    else:
        scale = do_val(key, scale, "Scale")
        shift_t = do_val(key, shift_t, "Shift")
        key = do_key(key)
    return key, scale, shift_t


def style_decoder(fcs: Tensor, key: Tensor, scale: Tensor, shift_t: Tensor, p: Dict[str, Tensor], cfg, sd=None) -> Tensor:
    """StyleDecoder.forward default branch, style_transformer.py:1045-1059, :1123-1128."""
    C = fcs.shape[-1]
    ws, sh, nh = cfg["decoder_window_size"], cfg["decoder_shift_size"], cfg["decoder_num_heads"]
    b = "decoder.MHA_self_attn."
    n1 = F.layer_norm(fcs, (C,), p[b + "norm1.weight"], p[b + "norm1.bias"], EPS)
    q = fcs + _sd(sd, shifted_window_attention(n1, n1, n1, p, b + "attn.", ws, sh, nh))  # :390
    if not cfg.get("decoder_exclude_MLP_after_Fcs_self_MHA", False):
        q = q + _sd(sd, mlp(F.layer_norm(q, (C,), p[b + "norm2.weight"], p[b + "norm2.bias"], EPS), p, b + "mlp."))  # :392
    affine = cfg.get("decoder_use_instance_norm_with_affine", False)
    in_q = "decoder.instance_norm_Query." if affine else None
    in_k = "decoder.instance_norm_Key." if affine else None
    after = cfg.get("decoder_use_Key_instance_norm_after_linear_transformation", True)
    if not cfg.get("decoder_use_regular_MHA_instead_of_Swin_at_the_end", False):
        q_in = instance_norm_nhwc_affine(q, p, in_q)                                     # :1056
        key_in = instance_norm_nhwc_affine(key, p, in_k)                                 # :1057
        sigma, mu = sigma_mu_attention(q_in, key_in, scale, shift_t, p, "decoder.decoder_MHA_for_sigma_and_mu.", ws, sh, nh,
                                       in_q, in_k, after)                                # :1059
    else:
        # :1060-1119  one global single-head attention over all tokens of the image (no windows, no bias, no mask)
        # The reference hands nn.InstanceNorm2d a 3-D tensor [B, C, H*W] here (:1073-1090); torch treats that as ONE un-batched
        # sample with B "channels", so every image is normalised over ALL its channels and tokens jointly (biased variance, eps 1e-5)
        # - reproduced as is.  (With affine=True the [dim] weight would meet B channels: the reference raises unless B == dim.)
        if affine:
            raise ValueError("decoder_use_regular_MHA_instead_of_Swin_at_the_end with affine InstanceNorm fails in the reference "
                             "(InstanceNorm2d weight [dim] against B channels)")
        B, H, W, _ = q.shape
        lin = lambda t, n: F.linear(t, p[f"decoder.{n}.weight"], p[f"decoder.{n}.bias"])

        def image_norm(t):
            mean = t.mean(dim=(1, 2, 3), keepdim=True)
            var = ((t - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
            return (t - mean) / torch.sqrt(var + EPS)
        q_in = image_norm(q)
        if after:
            k2 = image_norm(lin(key, "linear_transformation_Key"))
        else:
            k2 = lin(image_norm(key), "linear_transformation_Key")
        v_s, v_sh = lin(scale, "linear_transformation_Scale"), lin(shift_t, "linear_transformation_Shift")
        a = F.softmax((q_in.reshape(B, H * W, C) * C ** -0.5) @ k2.reshape(B, H * W, C).transpose(-2, -1), dim=-1)
        sigma = lin(a @ v_s.reshape(B, H * W, C), "proj_sigma").view(B, H, W, C)
        mu = lin(a @ v_sh.reshape(B, H * W, C), "proj_mu").view(B, H, W, C)
    q = q * sigma + mu                                                                   # :1123 (un-normalised Query)
    return q + _sd(sd, mlp(q, p, "decoder.last_MLP."))                                   # :1125


DEFAULT_ST_CFG = dict(
    encoder_window_size=(7, 7), decoder_window_size=(7, 7),
    encoder_shift_size=(4, 4), decoder_shift_size=(4, 4),
    encoder_num_heads=8, decoder_num_heads=8,
)


def style_transformer(fc: Tensor, fs: Tensor, k: int, p: Dict[str, Tensor], cfg=None, sd=None) -> Tensor:
    """StyleTransformer.forward, style_transformer.py:1229-1245."""
    cfg = cfg or DEFAULT_ST_CFG
    scale = shift_t = fs
    for _ in range(k):
        fs, scale, shift_t = style_encoder(fs, scale, shift_t, p, cfg, sd)
        fc = style_decoder(fc, fs, scale, shift_t, p, cfg, sd)
    return fc


# --------------------------------------------------------------------------------------
# Image decoder (codes/decoder.py:23-55)
# --------------------------------------------------------------------------------------
DECODER_CONV_IDX = (0, 3, 5, 7, 9, 12, 14, 17, 19)       # positions of the Conv2d's in the nn.Sequential
DECODER_UPSAMPLE_BEFORE = (3, 12, 17)                    # convs preceded by Upsample(x2, nearest)


def image_decoder(x_nchw: Tensor, p: Dict[str, Tensor], prefix: str = "decoder.") -> Tensor:
    x = x_nchw
    for i in DECODER_CONV_IDX:
        if i in DECODER_UPSAMPLE_BEFORE:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), p[f"{prefix}{i}.weight"], p[f"{prefix}{i}.bias"])
        if i != DECODER_CONV_IDX[-1]:
            x = F.relu(x)
    return x


# --------------------------------------------------------------------------------------
# Swin-B features[:4]  (torchvision arithmetic, restated; see module docstring)
# --------------------------------------------------------------------------------------
def swin_window_attention(x: Tensor, p: Dict[str, Tensor], prefix: str, ws, shift, nh: int) -> Tensor:
    """Fused-qkv form of the window attention: one [3C, C] weight split as q|k|v thirds
    (split order documented in load_pretrained_weights_to_style_transformer.py:52-60)."""
    B, H, W, C = x.shape
    src, region, _, _ = window_plan(H, W, tuple(ws), tuple(shift))
    qkv = F.linear(_to_windows(x, src), p[prefix + "qkv.weight"], p[prefix + "qkv.bias"])
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    bias = relative_position_bias(p[prefix + "relative_position_bias_table"],
                                  p[prefix + "relative_position_index"], ws)
    a = _attn_probs(_heads(q, nh), _heads(k, nh), bias, region, nh)
    o = (a @ _heads(v, nh)).transpose(1, 2).reshape(q.shape[0], q.shape[1], C)
    o = F.linear(o, p[prefix + "proj.weight"], p[prefix + "proj.bias"])
    return _from_windows(o, src, B, H, W)


def swin_block(x: Tensor, p: Dict[str, Tensor], prefix: str, ws, shift, nh: int, sd=None) -> Tensor:
    """x + SD(attn(norm1 x)); x + SD(mlp(norm2 x))   (layout: load_pretrained...py:29-49; torchvision's block applies
    StochasticDepth(p_block, "row") to both branches - identity in eval mode, sd=None)."""
    C = x.shape[-1]
    n = F.layer_norm(x, (C,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], EPS)
    x = x + _sd(sd, swin_window_attention(n, p, prefix + "attn.", ws, shift, nh))
    n = F.layer_norm(x, (C,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], EPS)
    return x + _sd(sd, mlp(n, p, prefix + "mlp."))


def patch_merging(x: Tensor, p: Dict[str, Tensor], prefix: str) -> Tensor:
    """pad to even, concat x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2] -> LN(4C) -> Linear(4C,2C,no bias)."""
    B, H, W, C = x.shape
    x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], dim=-1)
    x = F.layer_norm(x, (4 * C,), p[prefix + "norm.weight"], p[prefix + "norm.bias"], EPS)
    return F.linear(x, p[prefix + "reduction.weight"])


SWIN_B_SD_PROBS = tuple(0.5 * n / 23.0 for n in range(4))   # torchvision swin_b: stochastic_depth_prob 0.5 spread linearly over 24 blocks
SWIN_SD_CALLS = 6                                            # per encoder call in train mode: blocks 1, 2, 3 (block 0 has p = 0: no draw) x 2


def swin_encoder(img: Tensor, p: Dict[str, Tensor], prefix: str = "", ws=(7, 7), heads=(4, 8), depths=(2, 2), sd=None) -> Tensor:
    """swin_b.features[:4]: NCHW image -> NHWC [B, H/8, W/8, 256] (built in reference codes/utils.py:89-93,
    called at codes/full_model.py:219-220).  sd: train-mode StochasticDepth masks (the frozen encoder is still in train mode in the
    reference's trainer); a block whose probability is 0 draws nothing, like torchvision."""
    x = F.conv2d(img, p[prefix + "0.0.weight"], p[prefix + "0.0.bias"], stride=4).permute(0, 2, 3, 1)
    x = F.layer_norm(x, (x.shape[-1],), p[prefix + "0.2.weight"], p[prefix + "0.2.bias"], EPS)
    n = 0
    for i in range(depths[0]):
        sh = (0, 0) if i % 2 == 0 else (ws[0] // 2, ws[1] // 2)
        x = swin_block(x, p, f"{prefix}1.{i}.", ws, sh, heads[0], sd if n > 0 else None)
        n += 1
    x = patch_merging(x, p, prefix + "2.")
    for i in range(depths[1]):
        sh = (0, 0) if i % 2 == 0 else (ws[0] // 2, ws[1] // 2)
        x = swin_block(x, p, f"{prefix}3.{i}.", ws, sh, heads[1], sd if n > 0 else None)
        n += 1
    return x


# --------------------------------------------------------------------------------------
# full forward (codes/full_model.py:214-226)
# --------------------------------------------------------------------------------------
def full_forward(content: Tensor, style: Tensor, k: int, swin_sd, st_sd, dec_sd, cfg=None, swin_heads=(4, 8)) -> Tensor:
    """swin_heads: (4, 8) for swin_B, (3, 6) for swin_T / swin_S (head_dim 32 in all three: codes/utils.py:82-88)"""
    fc = swin_encoder(content, swin_sd, heads=swin_heads)
    fs = swin_encoder(style, swin_sd, heads=swin_heads)
    fcs = style_transformer(fc, fs, k, st_sd, cfg).permute(0, 3, 1, 2)
    return image_decoder(fcs, dec_sd)


# --------------------------------------------------------------------------------------
# state-dict shapes (used by tests to build seeded weights without the reference)
# --------------------------------------------------------------------------------------
def style_transformer_shapes(dim=256, heads=8, ws=(7, 7), mlp_ratio=4.0, affine_in=False, regular_mha=False) -> Dict[str, Tuple[int, ...]]:
    hid = int(dim * mlp_ratio)
    T = (2 * ws[0] - 1) * (2 * ws[1] - 1)
    N = ws[0] * ws[1]
    s: Dict[str, Tuple[int, ...]] = {}

    def attn(pfx, names):
        s[pfx + "relative_position_bias_table"] = (T, heads)
        s[pfx + "relative_position_index"] = (N * N,)
        for n in names:
            s[f"{pfx}{n}.weight"] = (dim, dim)
            s[f"{pfx}{n}.bias"] = (dim,)

    def mlp_(pfx):
        s[pfx + "0.weight"] = (hid, dim); s[pfx + "0.bias"] = (hid,)
        s[pfx + "3.weight"] = (dim, hid); s[pfx + "3.bias"] = (dim,)

    attn("encoder.shared_MHA_without_MLP.attn.", ("Wq", "Wk", "Wv", "proj"))
    for n in ("Key", "Scale", "Shift"):
        mlp_(f"encoder.encoder_MLP_{n}.")
    for n in ("norm1", "norm2"):
        s[f"decoder.MHA_self_attn.{n}.weight"] = (dim,); s[f"decoder.MHA_self_attn.{n}.bias"] = (dim,)
    attn("decoder.MHA_self_attn.attn.", ("Wq", "Wk", "Wv", "proj"))
    mlp_("decoder.MHA_self_attn.mlp.")
    mlp_("decoder.last_MLP.")
    if affine_in:
        for n in ("instance_norm_Query", "instance_norm_Key"):
            s[f"decoder.{n}.weight"] = (dim,); s[f"decoder.{n}.bias"] = (dim,)
    if regular_mha:
        for n in ("linear_transformation_Key", "linear_transformation_Scale", "linear_transformation_Shift", "proj_sigma", "proj_mu"):
            s[f"decoder.{n}.weight"] = (dim, dim); s[f"decoder.{n}.bias"] = (dim,)
    else:
        attn("decoder.decoder_MHA_for_sigma_and_mu.", ("Wk", "Wv_scale", "Wv_shift", "proj"))
        if affine_in:        # the attention module registers the two norm modules it is handed as its own children (:645-646)
            for n, src in (("instance_norm_q", "instance_norm_Query"), ("instance_norm_k", "instance_norm_Key")):
                s[f"decoder.decoder_MHA_for_sigma_and_mu.{n}.weight"] = (dim,); s[f"decoder.decoder_MHA_for_sigma_and_mu.{n}.bias"] = (dim,)
    return s


def decoder_shapes(ch=256) -> Dict[str, Tuple[int, ...]]:
    chans = [(ch, ch // 2), (ch // 2, ch // 2), (ch // 2, ch // 2), (ch // 2, ch // 2), (ch // 2, ch // 4),
             (ch // 4, ch // 4), (ch // 4, ch // 8), (ch // 8, ch // 8), (ch // 8, 3)]
    s = {}
    for i, (ci, co) in zip(DECODER_CONV_IDX, chans):
        s[f"decoder.{i}.weight"] = (co, ci, 3, 3)
        s[f"decoder.{i}.bias"] = (co,)
    return s


def swin_shapes(embed=128, heads=(4, 8), ws=(7, 7), depths=(2, 2), mlp_ratio=4.0) -> Dict[str, Tuple[int, ...]]:
    T = (2 * ws[0] - 1) * (2 * ws[1] - 1)
    N = ws[0] * ws[1]
    s = {"0.0.weight": (embed, 3, 4, 4), "0.0.bias": (embed,), "0.2.weight": (embed,), "0.2.bias": (embed,)}

    def block(pfx, dim, nh):
        hid = int(dim * mlp_ratio)
        s[pfx + "norm1.weight"] = (dim,); s[pfx + "norm1.bias"] = (dim,)
        s[pfx + "attn.relative_position_bias_table"] = (T, nh)
        s[pfx + "attn.relative_position_index"] = (N * N,)
        s[pfx + "attn.qkv.weight"] = (3 * dim, dim); s[pfx + "attn.qkv.bias"] = (3 * dim,)
        s[pfx + "attn.proj.weight"] = (dim, dim); s[pfx + "attn.proj.bias"] = (dim,)
        s[pfx + "norm2.weight"] = (dim,); s[pfx + "norm2.bias"] = (dim,)
        s[pfx + "mlp.0.weight"] = (hid, dim); s[pfx + "mlp.0.bias"] = (hid,)
        s[pfx + "mlp.3.weight"] = (dim, hid); s[pfx + "mlp.3.bias"] = (dim,)

    for i in range(depths[0]):
        block(f"1.{i}.", embed, heads[0])
    s["2.reduction.weight"] = (2 * embed, 4 * embed)
    s["2.norm.weight"] = (4 * embed,); s["2.norm.bias"] = (4 * embed,)
    for i in range(depths[1]):
        block(f"3.{i}.", 2 * embed, heads[1])
    return s
