"""Generate golden vectors from the *imported reference* (build container only).

Run:  python tests/golden/make_golden.py        (needs /root/reference; never runs on the GPU box)

The unmodified reference files codes/style_transformer.py and codes/decoder.py are imported with an
in-process stand-in for the two torchvision helpers they need (StochasticDepth, MLP - both absent
from this image).  Weights are overwritten with oracle.seeded tensors so tests can re-create them
without the reference; only seeds and expected outputs are stored (float32 .npz).

It also asserts, while generating, that oracle/master_oracle.py agrees with the reference
(<= 2e-5 max-abs) - that is the pinning of the oracle.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
REF = os.environ.get("MASTER_REFERENCE", "/root/reference")


def _install_torchvision_stub():
    class StochasticDepth(nn.Module):
        def __init__(self, p, mode):
            super().__init__(); self.p, self.mode = p, mode
        def forward(self, x):
            assert not self.training, "golden vectors are eval-mode only"
            return x

    class MLP(nn.Sequential):
        def __init__(self, in_channels, hidden_channels, norm_layer=None, activation_layer=nn.ReLU,
                     inplace=None, bias=True, dropout=0.0):
            layers, d = [], in_channels
            for h in hidden_channels[:-1]:
                layers += [nn.Linear(d, h, bias=bias), activation_layer(), nn.Dropout(dropout)]
                d = h
            layers += [nn.Linear(d, hidden_channels[-1], bias=bias), nn.Dropout(dropout)]
            super().__init__(*layers)

    mods = {n: types.ModuleType(n) for n in ("torchvision", "torchvision.ops", "torchvision.ops.stochastic_depth",
                                             "torchvision.ops.misc", "torchvision.transforms")}
    mods["torchvision.ops.stochastic_depth"].StochasticDepth = StochasticDepth
    mods["torchvision.ops.misc"].MLP = MLP
    sys.modules.update(mods)


def main():
    _install_torchvision_stub()
    sys.path.insert(0, REF)
    from codes import style_transformer as R            # the reference, unmodified
    from codes.decoder import Decoder as RefDecoder
    from oracle import master_oracle as O
    from oracle.seeded import seeded_state_dict, seeded_input

    torch.set_grad_enabled(False)
    out = {}
    TOL = 2e-5

    def check(name, got, ref):
        err = (got - ref).abs().max().item()
        print(f"{name:48s} oracle-vs-reference max|d| = {err:.3e}   ref std {ref.std().item():.3f}")
        assert err <= TOL * max(1.0, ref.abs().max().item()), name
        out[name] = ref.numpy().astype(np.float32)

    # ---- 1. shifted_window_attention: {shift on/off} x {pad on/off}, B=2 -------------------------
    dim, nh = 64, 2
    for H, W, shift in ((7, 7, (0, 0)), (14, 14, (0, 0)), (14, 14, (3, 3)), (9, 9, (4, 4)), (9, 12, (4, 4)), (14, 9, (4, 0))):
        m = R.ShiftedWindowAttention(dim, nh, [7, 7], list(shift)).eval()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = seeded_state_dict(shapes, seed=11)
        m.load_state_dict(sd)
        tag = f"swa_H{H}_W{W}_s{shift[0]}{shift[1]}"
        xq, xk, xv = (seeded_input(f"{tag}_{n}", (2, H, W, dim), 11) for n in "qkv")
        ref = m(xq, xk, xv)
        got = O.shifted_window_attention(xq, xk, xv, sd, "", (7, 7), shift, nh)
        check(tag, got, ref)

    # ---- 1b. torchvision-style fused qkv block cross-check (q=k=v, split weight) -----------------
    m = R.ShiftedWindowAttention(dim, nh, [7, 7], [3, 3]).eval()
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=12)
    m.load_state_dict(sd)
    fused = {"qkv.weight": torch.cat([sd["Wq.weight"], sd["Wk.weight"], sd["Wv.weight"]]),
             "qkv.bias": torch.cat([sd["Wq.bias"], sd["Wk.bias"], sd["Wv.bias"]]),
             "proj.weight": sd["proj.weight"], "proj.bias": sd["proj.bias"],
             "relative_position_bias_table": sd["relative_position_bias_table"],
             "relative_position_index": sd["relative_position_index"]}
    x = seeded_input("fused_x", (2, 10, 10, dim), 12)
    check("swin_fused_qkv_H10_s33", O.swin_window_attention(x, fused, "", (7, 7), (3, 3), nh), m(x, x, x))

    # ---- 2. StyleSwinTransformerBlock {norm None/LN} x {exclude_MLP T/F} -------------------------
    for norm, excl in ((None, True), (nn.LayerNorm, False), (nn.LayerNorm, True), (None, False)):
        m = R.StyleSwinTransformerBlock(dim, nh, [7, 7], [4, 4], norm_layer=norm, exclude_MLP_after=excl).eval()
        sd = seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=13)
        m.load_state_dict(sd)
        tag = f"block_norm{int(norm is not None)}_excl{int(excl)}"
        xq, xk, xv = (seeded_input(f"{tag}_{n}", (2, 9, 9, dim), 13) for n in "qkv")
        ref = m(xq, xk, xv, calculating_Key_in_encoder=False)
        # oracle composition of the same block
        C = dim
        ln = (lambda t, n: torch.nn.functional.layer_norm(t, (C,), sd[f"{n}.weight"], sd[f"{n}.bias"], 1e-5)) if norm else (lambda t, n: t)
        res = xv if excl else xq
        y = res + O.shifted_window_attention(ln(xq, "norm1"), ln(xk, "norm1"), ln(xv, "norm1"), sd, "attn.", (7, 7), (4, 4), nh)
        if not excl:
            y = y + O.mlp(ln(y, "norm2"), sd, "mlp.")
        check(tag, y, ref)

    # ---- 3. sigma/mu attention, StyleEncoder (both orders), StyleDecoder, StyleTransformer -------
    def build_st(dim, nh, flag=True):
        m = R.StyleTransformer(dim, dim, nh, nh, [7, 7], [7, 7], [4, 4], [4, 4],
                               encoder_if_use_processed_Key_in_Scale_and_Shift_calculation=flag).eval()
        sd = seeded_state_dict(O.style_transformer_shapes(dim, nh), seed=14)
        assert set(sd) == set(m.state_dict()), set(sd) ^ set(m.state_dict())
        for k, v in m.state_dict().items():
            assert tuple(v.shape) == tuple(sd[k].shape), k
        m.load_state_dict(sd)
        return m, sd

    cfg = dict(O.DEFAULT_ST_CFG, encoder_num_heads=nh, decoder_num_heads=nh)
    m, sd = build_st(dim, nh)
    a, b, c, d = (seeded_input(f"sigmu_{n}", (2, 9, 9, dim), 14) for n in "abcd")
    rs, rm = m.decoder.decoder_MHA_for_sigma_and_mu(a, b, c, d)
    gs, gm = O.sigma_mu_attention(a, b, c, d, sd, "decoder.decoder_MHA_for_sigma_and_mu.", (7, 7), (4, 4), nh)
    check("sigma_mu_sigma", gs, rs); check("sigma_mu_mu", gm, rm)

    for flag in (True, False):
        m2, sd2 = build_st(dim, nh, flag)
        rk, rsc, rsh = m2.encoder(a, b, c)
        gk, gsc, gsh = O.style_encoder(a, b, c, sd2, dict(cfg, encoder_if_use_processed_Key_in_Scale_and_Shift_calculation=flag))
        for n, g_, r_ in (("key", gk, rk), ("scale", gsc, rsc), ("shift", gsh, rsh)):
            check(f"style_encoder_flag{int(flag)}_{n}", g_, r_)

    check("style_decoder", O.style_decoder(a, b, c, d, sd, cfg), m.decoder(a, b, c, d))

    fc, fs = seeded_input("st_fc", (2, 12, 16, dim), 14), seeded_input("st_fs", (2, 12, 16, dim), 14)
    for k in (1, 3):
        check(f"style_transformer_small_k{k}", O.style_transformer(fc, fs, k, sd, cfg), m(fc, fs, k))

    # full width (dim 256, 8 heads), 32x32 feature map = the 256x256 image configuration, B=1
    mf, sdf = build_st(256, 8)
    fc, fs = seeded_input("st256_fc", (1, 32, 32, 256), 15), seeded_input("st256_fs", (1, 32, 32, 256), 15)
    sdf = seeded_state_dict(O.style_transformer_shapes(256, 8), seed=15); mf.load_state_dict(sdf)
    for k in (1, 3):
        check(f"style_transformer_256_k{k}", O.style_transformer(fc, fs, k, sdf), mf(fc, fs, k))

    # ---- 4. image decoder --------------------------------------------------------------------------
    dm = RefDecoder().eval()
    dsd = seeded_state_dict(O.decoder_shapes(256), seed=16)
    dm.load_state_dict(dsd)
    for hw in (8, 32):
        x = seeded_input(f"dec_x{hw}", (1, 256, hw, hw), 16)
        check(f"decoder_{hw}", O.image_decoder(x, dsd), dm(x))

    # ---- 5. ST(256,k=1) -> permute -> decoder chain, the reference's own composition (full_model.py:222-224)
    x = mf(fc, fs, 1).permute(0, 3, 1, 2)
    check("st256_k1_then_decoder", O.image_decoder(O.style_transformer(fc, fs, 1, sdf).permute(0, 3, 1, 2), dsd), dm(x))

    path = os.path.join(HERE, "golden_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"{os.path.getsize(path)/1e6:.2f} MB, {len(out)} vectors")


if __name__ == "__main__":
    main()
