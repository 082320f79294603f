"""Train-mode forward of the stylization path (reference train_only_inner_loop.py:519-527: `.train()` model, StochasticDepth row
masks active) built from the autograd Functions of autograd.py, so that `total_loss.backward()` runs the HIP backward kernels and
fills `.grad` of the reference-named parameters.

Same arithmetic as the inference `run` methods of modules.py, with two differences the reference has in train mode:
  * every residual branch goes through torchvision StochasticDepth(p, "row") (style_transformer.py:361,:390-397,:819,:865-882,:989,:1125):
    out = residual + mask[b] * branch with mask = bernoulli(1-p)/(1-p) per sample (master_rowscale_add);
  * parameters are used directly (no detached/derived copies), fused projections are torch.cat views of them.
The tensor-identity fusions that stay valid under per-call masks are kept: q = k = v -> one [3C,C] projection; q = k -> [2C,C]; the
Scale and Shift attention calls share q, k and the softmax (two value streams in one launch), and in layer 1 (Scale is Shift is Fs) the
attention output is computed once and only the two row masks differ.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
from torch import Tensor, nn

from . import ops
from .autograd import (ChannelAffineFn, ConvReflectFn, InstNormFn, LayerNormFn, LinearFn, MLPFn, MulAddFn, RowScaleAddFn, SoftmaxRowsFn,
                       WindowAttnFn)
from .window_plan import window_plan

HEAD_DIM = 32


class StochasticDepthMasks:
    """Source of the row masks.  Default: drawn on the device the way torchvision.ops.stochastic_depth does
    (noise = empty([B]).bernoulli_(1-p).div_(1-p)), in the order the reference's modules call it.  Parity tests install an explicit
    list with `set_masks` (CPU and GPU random streams differ), consumed in the same order."""

    def __init__(self):
        self._explicit: Optional[List[Tensor]] = None
        self.generator: Optional[torch.Generator] = None

    def set_masks(self, masks: Optional[Sequence[Tensor]]):
        self._explicit = list(masks) if masks is not None else None

    def remaining(self) -> int:
        return len(self._explicit) if self._explicit is not None else 0

    def draw(self, B: int, p: float, device) -> Tensor:
        if self._explicit is not None:
            if not self._explicit:
                raise RuntimeError("StochasticDepthMasks: the explicit mask list is exhausted")
            return self._explicit.pop(0).to(device=device, dtype=torch.float32).contiguous()
        q = self.__dict__.get("_queue")
        if q:
            (qB, qp, qdev), row = q[0]
            if qB == B and qp == p and qdev == device:
                q.pop(0)
                return row
            q.clear()                                    # the caller left the announced order: fall back to one draw per call
        keep = 1.0 - p
        noise = torch.empty((B,), dtype=torch.float32, device=device).bernoulli_(keep, generator=self.generator)
        if keep > 0.0:
            noise.div_(keep)
        return noise

    def prefetch(self, ps: Sequence[float], B: int, device) -> None:
        """Draw the masks of the next len(ps) draw(B, ps[i], device) calls in TWO launches (one Bernoulli over a [n, B] matrix of keep
        probabilities, one division) instead of two per mask: a train step draws 34-50 masks of 8 floats each.  Same distribution as
        torchvision's per-call `empty([B]).bernoulli_(1 - p).div_(1 - p)`; the device random stream is consumed in one block instead of
        n (no CPU / reference stream is reproducible on the device either way - parity tests install explicit masks)."""
        if self._explicit is not None or not ps:
            return
        # the keep probabilities live on the device, one upload per distinct draw order (a step captured in a hipGraph must not copy
        # from pageable host memory, and k = 1..4 give four orders)
        cache = self.__dict__.setdefault("_keep_cache", {})
        ck = (tuple(float(p) for p in ps), str(device))
        keep = cache.get(ck)
        if keep is None:
            keep = cache[ck] = torch.tensor([1.0 - float(p) for p in ps], dtype=torch.float32).clamp_(min=0.0).to(device)
        keep = keep[:, None].expand(len(ps), B)
        noise = torch.bernoulli(keep, generator=self.generator)
        noise = noise / keep.clamp(min=1e-30)            # p = 1 rows are all zero either way
        self.__dict__["_queue"] = [((B, float(p), device), noise[i]) for i, p in enumerate(ps)]


sd_masks = StochasticDepthMasks()          # the process-wide source (one model per process; parity tests install explicit masks here)


def _residual(branch: Tensor, res: Tensor, p: float, training: bool, sd: StochasticDepthMasks, B: int, T: int) -> Tensor:
    """res + StochasticDepth(branch)   (torchvision: identity when not training or p == 0 - no random draw then)"""
    if training and p > 0.0:
        mask = sd.draw(B, p, branch.device)
    else:
        mask = _ones(B, branch.device)
    return RowScaleAddFn.apply(branch, res, mask, B, T)


_ONES = {}


def _ones(B: int, device) -> Tensor:
    key = (B, str(device))
    if key not in _ONES:
        _ONES[key] = torch.ones((B,), dtype=torch.float32, device=device)
    return _ONES[key]


# Tensors derived from parameters inside ONE StyleTransformer call (k layers share their weights, :1239: the shared encoder block runs
# 2k times, the decoder's two attention modules k times each): the [heads,64,64] bias tiles and the Wq|Wk|Wv concatenations are built
# once per call and reused as the same autograd node - its backward (index scatter into the table, splits of the fused weight
# gradient) then runs once per module instead of once per use, and autograd sums the uses' gradients in between.
_MEMO: Optional[dict] = None


def _memo(key, build: Callable[[], Tensor]) -> Tensor:
    if _MEMO is None:
        return build()
    hit = _MEMO.get(key)
    if hit is None:
        hit = _MEMO[key] = build()
    return hit


def _cat(tag: str, owner, *params: Tensor) -> Tensor:
    return _memo((tag, id(owner)), lambda: torch.cat(list(params), dim=0))


def _bias64(attn) -> Tensor:
    """differentiable [heads,64,64] bias tile from the relative-position table (style_transformer.py:21-28); key slots >= N = -1e30"""
    return _memo(("bias64", id(attn)), lambda: _bias64_build(attn))


def _bias64_build(attn) -> Tensor:
    from .autograd import BiasTileFn
    N = attn.window_size[0] * attn.window_size[1]
    return BiasTileFn.apply(attn.relative_position_bias_table, attn.relative_position_index, N)


def _sd_mask(p: float, training: bool, sd: StochasticDepthMasks, B: int, device) -> Tensor:
    """the [B] row mask of one residual branch (torchvision StochasticDepth "row": identity - ones - when not training or p == 0)"""
    return sd.draw(B, p, device) if (training and p > 0.0) else _ones(B, device)


def _mlp(mlp: nn.Sequential, x: Tensor, norm: Optional[nn.LayerNorm] = None, fork: bool = False, mask: Optional[Tensor] = None, T: int = 0):
    return MLPFn.apply(x, mlp[0].weight, mlp[0].bias, mlp[3].weight, mlp[3].bias,
                       norm.weight if norm is not None else None, norm.bias if norm is not None else None,
                       norm.eps if norm is not None else 1e-5, fork, mask, T)


def _mlp_residual(mlp: nn.Sequential, x: Tensor, norm: Optional[nn.LayerNorm], p: float, training: bool, sd: StochasticDepthMasks, B: int,
                  T: int, mask: Optional[Tensor] = None) -> Tensor:
    """x + StochasticDepth(mlp(norm(x))) as ONE autograd Function: fc2's epilogue adds the skip connection and applies the row mask, the
    backward kernels apply the mask to the branch's gradient and add the skip gradient (no rowscale_add launch, no autograd add).
    `mask`: drawn by the caller when the reference's draw order asks for it earlier."""
    if mask is None:
        mask = _sd_mask(p, training, sd, B, x.device)
    return _mlp(mlp, x, norm, mask=mask, T=T)


# ---------------------------------------------------------------------------------------------------------------------------------------
# ShiftedWindowAttention (three inputs) -> projected attention outputs, one per value stream (no residual)
# ---------------------------------------------------------------------------------------------------------------------------------------
def attention(attn, xq: Tensor, xk: Tensor, xvs: Sequence[Tensor], B: int, H: int, W: int, norm: Optional[nn.LayerNorm] = None,
              skips: Optional[dict] = None, res_masks: Optional[Sequence] = None) -> List[Tensor]:
    """skips (out): id(input tensor) -> the alias of it that the LayerNorm handed back (LayerNormFn fork).  A caller that uses the alias
    for everything else the tensor feeds (the skip connection of the block) leaves the norm as the tensor's only consumer.
    res_masks: [(residual tensor, StochasticDepth row mask [B])] per value stream: the function then returns the finished residual
    blocks  res + mask * proj(attention)  instead of the bare projections (the residual is looked up in `skips` first)."""
    C = xq.shape[1]
    T = H * W
    plan = window_plan(H, W, attn.window_size, attn.shift_size, xq.device)
    scale = HEAD_DIM ** -0.5
    if norm is not None:
        memo = []

        def ln(t):
            for src, o in memo:
                if src is t:
                    return o
            if skips is not None:
                o, alias = LayerNormFn.apply(t, norm.weight, norm.bias, norm.eps, True)
                skips[id(t)] = alias
            else:
                o = LayerNormFn.apply(t, norm.weight, norm.bias, norm.eps)
            memo.append((t, o))
            return o
        xq, xk, xvs = ln(xq), ln(xk), [ln(v) for v in xvs]
    uniq_v: List[Tensor] = []
    vmap = []
    for v in xvs:
        for i, s in enumerate(uniq_v):
            if s is v:
                vmap.append(i)
                break
        else:
            vmap.append(len(uniq_v)); uniq_v.append(v)
    if len(uniq_v) > 2:
        raise NotImplementedError("at most two value streams per attention call")
    bias64 = _bias64(attn)
    if xq is xk and len(uniq_v) == 1 and uniq_v[0] is xq:
        w = _cat("wqkv", attn, attn.Wq.weight, attn.Wk.weight, attn.Wv.weight)
        b = _cat("bqkv", attn, attn.Wq.bias, attn.Wk.bias, attn.Wv.bias)
        qkv = LinearFn.apply(xq, w, b)
        o0, _ = WindowAttnFn.apply(qkv, qkv, qkv, None, attn.Wk.bias, attn.Wv.bias, None, bias64, (0, C, 2 * C, 0), plan, B, T,
                                   attn.num_heads, scale)
        cores = [o0]
    else:
        if xq is xk:
            qk = LinearFn.apply(xq, _cat("wqk", attn, attn.Wq.weight, attn.Wk.weight), _cat("bqk", attn, attn.Wq.bias, attn.Wk.bias))
            tq, tk, oq, ok = qk, qk, 0, C
        else:
            tq, tk, oq, ok = LinearFn.apply(xq, attn.Wq.weight, attn.Wq.bias), LinearFn.apply(xk, attn.Wk.weight, attn.Wk.bias), 0, 0
        vs = [LinearFn.apply(v, attn.Wv.weight, attn.Wv.bias) for v in uniq_v]
        v1 = vs[1] if len(vs) > 1 else None
        o0, o1 = WindowAttnFn.apply(tq, tk, vs[0], v1, attn.Wk.bias, attn.Wv.bias, attn.Wv.bias if v1 is not None else None, bias64,
                                    (oq, ok, 0, 0), plan, B, T, attn.num_heads, scale)
        cores = [o0] if v1 is None else [o0, o1]
    if res_masks is not None:
        # projection + StochasticDepth row mask + residual add of every value stream in one launch each (LinearFn with res / mask).  In
        # the first layer Scale and Shift are the same tensor (style_transformer.py:1235-1236): one attention core, projected once per
        # stream because each stream has its own mask and residual
        outs = []
        for j, i in enumerate(vmap):
            res, mask = res_masks[j]
            res = skips.get(id(res), res) if skips is not None else res
            outs.append(LinearFn.apply(cores[i], attn.proj.weight, attn.proj.bias, res, mask, T))
        return outs
    projected = [LinearFn.apply(c, attn.proj.weight, attn.proj.bias) for c in cores]
    return [projected[i] for i in vmap]


def block(blk, xq: Tensor, xk: Tensor, xvs: Sequence[Tensor], residuals: Sequence[Tensor], B, H, W, sd: StochasticDepthMasks) -> List[Tensor]:
    """StyleSwinTransformerBlock.forward (:376-398) for one q/k pair and one or two value streams"""
    T = H * W
    p = blk.stochastic_depth_prob
    skips: dict = {}
    # masks in the reference's draw order: the attention branches first, then their MLPs
    masks = [_sd_mask(p, blk.training, sd, B, xq.device) for _ in residuals]
    outs = attention(blk.attn, xq, xk, xvs, B, H, W, norm=blk.norm1 if blk.use_norm else None, skips=skips, res_masks=list(zip(residuals, masks)))
    if not blk.exclude_MLP_after:
        outs = [_mlp_residual(blk.mlp, o, blk.norm2 if blk.use_norm else None, p, blk.training, sd, B, T) for o in outs]
    return outs


def style_encoder(enc, key: Tensor, scale: Tensor, shift: Tensor, B, H, W, sd: StochasticDepthMasks):
    """StyleEncoder.forward (:855-912).  Mask order of the default branch: attn(Key), MLP_Key, attn(Scale), MLP_Scale, attn(Shift),
    MLP_Shift - the fused Scale/Shift attention launch draws its two masks in that order around the MLP masks."""
    blk = enc.shared_MHA_without_MLP
    T = H * W
    p_blk, p_enc, tr = blk.stochastic_depth_prob, enc.encoder_stochastic_depth_prob, enc.training

    def do_key(key):
        key = block(blk, key, key, [key], [key], B, H, W, sd)[0]                                       # :859-864
        return _mlp_residual(enc.encoder_MLP_Key, key, None, p_enc, tr, sd, B, T)                      # :865

    def do_scale_shift(key, scale, shift):
        # :867-882: both calls use q = k = Key: one softmax, two value streams; the residual adds draw their masks in the reference's
        # order (attn Scale, MLP Scale, attn Shift, MLP Shift).  Returns the Key to hand on as well: the alias its LayerNorm gave back
        skips: dict = {}
        dev = key.device
        m_s, m_smlp = _sd_mask(p_blk, blk.training, sd, B, dev), _sd_mask(p_enc, tr, sd, B, dev)      # the four masks in the reference's order,
        m_sh, m_shmlp = _sd_mask(p_blk, blk.training, sd, B, dev), _sd_mask(p_enc, tr, sd, B, dev)    # drawn before the fused launches use them
        s, sh = attention(blk.attn, key, key, [scale, shift], B, H, W, norm=blk.norm1 if blk.use_norm else None, skips=skips,
                          res_masks=[(scale, m_s), (shift, m_sh)])
        s = _mlp_residual(enc.encoder_MLP_Scale, s, None, p_enc, tr, sd, B, T, mask=m_smlp)
        sh = _mlp_residual(enc.encoder_MLP_Shift, sh, None, p_enc, tr, sd, B, T, mask=m_shmlp)
        return s, sh, skips.get(id(key), key)

    if enc.if_use_processed_Key_in_Scale_and_Shift_calculation:
        key = do_key(key)
        scale, shift, key = do_scale_shift(key, scale, shift)
    else:
        scale, shift, key = do_scale_shift(key, scale, shift)
        key = do_key(key)
    return key, scale, shift


def sigma_mu(att, xq: Tensor, xk: Tensor, xs: Tensor, xsh: Tensor, B, H, W):
    """ShiftedWindowAttention_for_decoder_last_MHA (:414-611), default flags: returns (sigma, mu)"""
    T = H * W
    C = xq.shape[1]
    plan = window_plan(H, W, att.window_size, att.shift_size, xq.device)
    nq, nk = getattr(att, "instance_norm_q", None), getattr(att, "instance_norm_k", None)           # affine norms, or None
    aff = lambda t, n: t if n is None else ChannelAffineFn.apply(t, n.weight, n.bias)
    q = aff(InstNormFn.apply(xq, None, B, T, 0, 1e-5)[0], nq)                                       # :468
    if not att.use_Key_instance_norm_after_linear_transformation:
        xk = aff(InstNormFn.apply(xk, None, B, T, 0, 1e-5)[0], nk)                                  # :470-472
    k = LinearFn.apply(xk, att.Wk.weight, att.Wk.bias)
    k_pad = att.Wk.bias
    if att.use_Key_instance_norm_after_linear_transformation:
        k, k_pad_b = InstNormFn.apply(k, att.Wk.bias, B, T, plan.n_pad, 1e-5)                       # :520-530 (padded map statistics)
        k = aff(k, nk)
        if k_pad_b is not None:
            k_pad = k_pad_b if nk is None else k_pad_b * nk.weight + nk.bias                        # [B,C]: the normalised pad row through the affine
    vs = LinearFn.apply(xs, att.Wv_scale.weight, att.Wv_scale.bias)
    vsh = LinearFn.apply(xsh, att.Wv_shift.weight, att.Wv_shift.bias)
    o_s, o_sh = WindowAttnFn.apply(q, k, vs, vsh, k_pad, att.Wv_scale.bias, att.Wv_shift.bias, _bias64(att), (0, 0, 0, 0), plan, B, T,
                                   att.num_heads, HEAD_DIM ** -0.5)
    return LinearFn.apply(o_s, att.proj.weight, att.proj.bias), LinearFn.apply(o_sh, att.proj.weight, att.proj.bias)   # :575-576,:594-595


def _image_norm(x: Tensor, B: int, T: int) -> Tensor:
    """the regular-MHA branch's InstanceNorm2d on a 3-D tensor = per-image normalisation over all channels and tokens (modules.py
    StyleDecoder._image_norm).  Train form: [B] statistics by torch reductions of the [B*T, C] matrix (autograd differentiates them)."""
    xb = x.view(B, -1)
    mean = xb.mean(dim=1, keepdim=True)
    var = ((xb - mean) ** 2).mean(dim=1, keepdim=True)
    return ((xb - mean) * torch.rsqrt(var + 1e-5)).view_as(x)


def regular_mha(dec, query: Tensor, key: Tensor, scale: Tensor, shift: Tensor, B, H, W):
    """:1060-1119 (decoder_use_regular_MHA_instead_of_Swin_at_the_end): returns (sigma, mu)"""
    T, C = H * W, query.shape[1]
    if dec.decoder_use_instance_norm_with_affine:
        raise ValueError(dec.REGULAR_MHA_AFFINE_ERROR.format(dim=C, B=B))
    if T % 4:
        raise NotImplementedError("the regular-MHA decoder end needs H*W % 4 == 0 tokens per image")
    lin = lambda t, m: LinearFn.apply(t, m.weight, m.bias)
    q_in = _image_norm(query, B, T)
    if dec.decoder_use_Key_instance_norm_after_linear_transformation:
        k = _image_norm(lin(key, dec.linear_transformation_Key), B, T)
    else:
        k = lin(_image_norm(key, B, T), dec.linear_transformation_Key)
    vs, vsh = lin(scale, dec.linear_transformation_Scale), lin(shift, dec.linear_transformation_Shift)
    Tp = -(-T // 32) * 32
    pad = lambda t: torch.nn.functional.pad(t, (0, 0, 0, Tp - T))                                   # [T,C] -> [Tp,C] zero rows
    outs_s, outs_sh = [], []
    for i in range(B):
        sl = slice(i * T, (i + 1) * T)
        p = SoftmaxRowsFn.apply(LinearFn.apply(q_in[sl], pad(k[sl]), None), C ** -0.5, T)           # [T, Tp], padding columns zero
        outs_s.append(LinearFn.apply(p, pad(vs[sl]).t().contiguous(), None))
        outs_sh.append(LinearFn.apply(p, pad(vsh[sl]).t().contiguous(), None))
    return lin(torch.cat(outs_s), dec.proj_sigma), lin(torch.cat(outs_sh), dec.proj_mu)


def style_decoder(dec, fcs: Tensor, key: Tensor, scale: Tensor, shift: Tensor, B, H, W, sd: StochasticDepthMasks) -> Tensor:
    """StyleDecoder.forward (:1045-1128)"""
    T = H * W
    query = block(dec.MHA_self_attn, fcs, fcs, [fcs], [fcs], B, H, W, sd)[0]                          # :1047
    if dec.decoder_use_regular_MHA_instead_of_Swin_at_the_end:
        sigma, mu = regular_mha(dec, query, key, scale, shift, B, H, W)
    else:
        q_in, _ = InstNormFn.apply(query, None, B, T, 0, 1e-5)                                        # :1056
        k_in, _ = InstNormFn.apply(key, None, B, T, 0, 1e-5)                                          # :1057
        if dec.decoder_use_instance_norm_with_affine:
            q_in = ChannelAffineFn.apply(q_in, dec.instance_norm_Query.weight, dec.instance_norm_Query.bias)
            k_in = ChannelAffineFn.apply(k_in, dec.instance_norm_Key.weight, dec.instance_norm_Key.bias)
        sigma, mu = sigma_mu(dec.decoder_MHA_for_sigma_and_mu, q_in, k_in, scale, shift, B, H, W)     # :1059
    query = MulAddFn.apply(query, sigma, mu)                                                          # :1123
    p = dec.decoder_stochastic_depth_prob                                                             # (:989)
    return _mlp_residual(dec.last_MLP, query, None, p, dec.training, sd, B, T)                        # :1125


def _mask_order(st, k: int) -> List[float]:
    """StochasticDepth probabilities of one StyleTransformer call in draw order (only the residuals that draw: training and p > 0)"""
    enc, dec = st.encoder, st.decoder
    eb, db = enc.shared_MHA_without_MLP, dec.MHA_self_attn
    pe_blk = eb.stochastic_depth_prob if eb.training else 0.0
    pe = enc.encoder_stochastic_depth_prob if enc.training else 0.0
    pd_blk = db.stochastic_depth_prob if db.training else 0.0
    pd = dec.decoder_stochastic_depth_prob if dec.training else 0.0
    key = [pe_blk, pe]
    scale_shift = [pe_blk, pe, pe_blk, pe]
    per_layer = (key + scale_shift) if enc.if_use_processed_Key_in_Scale_and_Shift_calculation else (scale_shift + key)
    per_layer = per_layer + [pd_blk] + ([] if db.exclude_MLP_after else [pd_blk]) + [pd]
    return [p for p in per_layer * k if p > 0.0]


def style_transformer(st, fc: Tensor, fs: Tensor, k: int, B, H, W, sd: StochasticDepthMasks) -> Tensor:
    global _MEMO
    from .autograd import shared_grads
    outer, _MEMO = _MEMO, {}
    scope = shared_grads.begin() if (outer is None and torch.is_grad_enabled()) else None
    try:
        sd.prefetch(_mask_order(st, k), B, fc.device)
        scale = shift = fs                                                                            # :1235-1236
        for _ in range(k):
            fs, scale, shift = style_encoder(st.encoder, fs, scale, shift, B, H, W, sd)
            fc = style_decoder(st.decoder, fc, fs, scale, shift, B, H, W, sd)
        return fc
    finally:
        _MEMO = outer
        if scope is not None:
            shared_grads.end(scope)


def image_decoder(dec, x_nhwc: Tensor) -> Tensor:
    """Decoder.forward (decoder.py:23-55,75) on an NHWC map: returns the fp32 NCHW image"""
    last = dec.CONV_IDX[-1]
    x = x_nhwc
    for i in dec.CONV_IDX:
        conv = dec.decoder[i]
        w, b = conv.weight, conv.bias
        co, ci = dec.padded_channels(i)                      # channel counts the kernels take (multiples of 32): zero-padded weights
        if (co, ci) != tuple(w.shape[:2]):
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, ci - w.shape[1], 0, co - w.shape[0]))
            b = torch.nn.functional.pad(b, (0, co - b.shape[0]))
        x = ConvReflectFn.apply(x, w, b, i in dec.UPSAMPLE_BEFORE, i != last, i == last)
    return x
