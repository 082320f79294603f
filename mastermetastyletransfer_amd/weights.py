"""Checkpoint interchange: host-side state-dict manipulation, no kernels.

remap_ms_swin_block: the reference's PretrainedSwinWeightsLoader (codes/load_pretrained_weights_to_style_transformer.py:16-733) as one
table.  Source: the state dict of ONE Microsoft-Swin SwinTransformerBlock saved as an nn.Sequential/ModuleList slice
("0.*" norm1, "1.*" WindowAttention with fused qkv, "3.*" norm2, "4.fc1/fc2" MLP; :29-49).  The fused qkv rows are split into thirds
q | k | v (:52-60) and the block is copied into EVERY attention / MLP of the StyleTransformer (:688-733): the encoder's shared attention
and its three MLPs, the decoder's self-attention block (with its two LayerNorms and MLP), the last MLP, and the sigma/mu attention
(Wk <- k, Wv_scale and Wv_shift <- v, proj; no Wq).  Pinned against the imported reference by tests/golden/make_golden_remap.py.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import torch
from torch import Tensor


def split_block(block_sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    qkv_w, qkv_b = block_sd["1.qkv.weight"].cpu(), block_sd["1.qkv.bias"].cpu()
    n = qkv_w.shape[0] // 3
    return {
        "norm1.weight": block_sd["0.weight"].cpu(), "norm1.bias": block_sd["0.bias"].cpu(),
        "table": block_sd["1.relative_position_bias_table"].cpu(), "index": block_sd["1.relative_position_index"].cpu().flatten(),
        "q.weight": qkv_w[:n], "k.weight": qkv_w[n:2 * n], "v.weight": qkv_w[2 * n:],
        "q.bias": qkv_b[:n], "k.bias": qkv_b[n:2 * n], "v.bias": qkv_b[2 * n:],
        "proj.weight": block_sd["1.proj.weight"].cpu(), "proj.bias": block_sd["1.proj.bias"].cpu(),
        "norm2.weight": block_sd["3.weight"].cpu(), "norm2.bias": block_sd["3.bias"].cpu(),
        "fc1.weight": block_sd["4.fc1.weight"].cpu(), "fc1.bias": block_sd["4.fc1.bias"].cpu(),
        "fc2.weight": block_sd["4.fc2.weight"].cpu(), "fc2.bias": block_sd["4.fc2.bias"].cpu(),
    }


def _attn(prefix: str, qkv_bias: bool, proj_bias: bool) -> Dict[str, str]:
    m = {f"{prefix}relative_position_bias_table": "table", f"{prefix}relative_position_index": "index",
         f"{prefix}Wq.weight": "q.weight", f"{prefix}Wk.weight": "k.weight", f"{prefix}Wv.weight": "v.weight", f"{prefix}proj.weight": "proj.weight"}
    if qkv_bias:
        m.update({f"{prefix}Wq.bias": "q.bias", f"{prefix}Wk.bias": "k.bias", f"{prefix}Wv.bias": "v.bias"})
    if proj_bias:
        m[f"{prefix}proj.bias"] = "proj.bias"
    return m


def _mlp(prefix: str) -> Dict[str, str]:
    return {f"{prefix}0.weight": "fc1.weight", f"{prefix}0.bias": "fc1.bias", f"{prefix}3.weight": "fc2.weight", f"{prefix}3.bias": "fc2.bias"}


def remap_table(encoder_qkv_bias=True, decoder_qkv_bias=True, encoder_proj_bias=True, decoder_proj_bias=True, encoder_norm_layer=None,
                decoder_norm_layer=True, decoder_exclude_MLP_after_Fcs_self_MHA=False) -> Dict[str, str]:
    """destination StyleTransformer key -> source piece of the block"""
    t: Dict[str, str] = {}
    e = "encoder.shared_MHA_without_MLP."
    t.update(_attn(e + "attn.", encoder_qkv_bias, encoder_proj_bias))
    if encoder_norm_layer:
        t.update({e + "norm1.weight": "norm1.weight", e + "norm1.bias": "norm1.bias"})      # (no norm2: the encoder block has no MLP)
    for n in ("Key", "Scale", "Shift"):
        t.update(_mlp(f"encoder.encoder_MLP_{n}."))
    d = "decoder.MHA_self_attn."
    t.update(_attn(d + "attn.", decoder_qkv_bias, decoder_proj_bias))
    if decoder_norm_layer:
        t.update({d + "norm1.weight": "norm1.weight", d + "norm1.bias": "norm1.bias"})
        if not decoder_exclude_MLP_after_Fcs_self_MHA:
            t.update({d + "norm2.weight": "norm2.weight", d + "norm2.bias": "norm2.bias"})
    if not decoder_exclude_MLP_after_Fcs_self_MHA:
        t.update(_mlp(d + "mlp."))
    t.update(_mlp("decoder.last_MLP."))
    s = "decoder.decoder_MHA_for_sigma_and_mu."
    t.update({s + "relative_position_bias_table": "table", s + "relative_position_index": "index", s + "Wk.weight": "k.weight",
              s + "Wv_scale.weight": "v.weight", s + "Wv_shift.weight": "v.weight", s + "proj.weight": "proj.weight"})
    if decoder_qkv_bias:
        t.update({s + "Wk.bias": "k.bias", s + "Wv_scale.bias": "v.bias", s + "Wv_shift.bias": "v.bias"})
    if decoder_proj_bias:
        t[s + "proj.bias"] = "proj.bias"
    return t


def remap_ms_swin_block(block_sd: Dict[str, Tensor], st_state_dict: "OrderedDict[str, Tensor]", encoder_dim=256, decoder_dim=256,
                        encoder_mlp_ratio=4, decoder_mlp_ratio=4, encoder_window_size=(7, 7), decoder_window_size=(7, 7), **flags):
    """returns st_state_dict with the block's tensors written into every destination key that exists in it
    (shape or dtype mismatches raise ValueError like the reference's guards, :441-681)"""
    assert encoder_dim == 256, "encoder_dim should be 256 for pre-trained weight loading"
    assert decoder_dim == 256, "decoder_dim should be 256 for pre-trained weight loading"
    assert encoder_mlp_ratio == 4, "encoder_mlp_ratio should be 4 for pre-trained weight loading"
    assert decoder_mlp_ratio == 4, "decoder_mlp_ratio should be 4 for pre-trained weight loading"
    assert list(encoder_window_size) == [7, 7], "encoder_window_size should be [7, 7] for pre-trained weight loading"
    assert list(decoder_window_size) == [7, 7], "decoder_window_size should be [7, 7] for pre-trained weight loading"
    pieces = split_block(block_sd)
    out = OrderedDict(st_state_dict)
    for dst, src in remap_table(**flags).items():
        if dst not in out:
            continue
        new = pieces[src]
        if tuple(out[dst].shape) != tuple(new.shape):
            raise ValueError(f"shape mismatch for {dst} (original: {tuple(out[dst].shape)}, new: {tuple(new.shape)})")
        if out[dst].dtype != new.dtype:
            raise ValueError(f"dtype mismatch for {dst} (original: {out[dst].dtype}, new: {new.dtype})")
        out[dst] = new.clone()
    return out
