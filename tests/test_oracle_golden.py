"""CPU: the oracle reproduces every vector the imported reference produced (tests/golden)."""
import numpy as np
import torch

from oracle import master_oracle as O
from tests import golden_cases as G

TOL = 2e-5   # bit-exact in the generating container; other BLAS builds may differ in the last bits


def close(got, ref, name):
    ref = torch.from_numpy(ref)
    err = (got - ref).abs().max().item()
    assert err <= TOL * max(1.0, ref.abs().max().item()), f"{name}: max|d|={err}"


@torch.no_grad()
def test_window_attention(golden):
    for H, W, shift in G.SWA_CASES:
        tag, sd, (q, k, v) = G.swa_case(H, W, shift)
        close(O.shifted_window_attention(q, k, v, sd, "", (7, 7), shift, G.NH), golden[tag], tag)


@torch.no_grad()
def test_blocks(golden):
    import torch.nn.functional as F
    for norm, excl in G.BLOCK_CASES:
        tag, sd, (q, k, v) = G.block_case(norm, excl)
        ln = (lambda t, n: F.layer_norm(t, (G.DIM,), sd[f"{n}.weight"], sd[f"{n}.bias"], 1e-5)) if norm else (lambda t, n: t)
        y = (v if excl else q) + O.shifted_window_attention(ln(q, "norm1"), ln(k, "norm1"), ln(v, "norm1"), sd, "attn.", (7, 7), (4, 4), G.NH)
        if not excl:
            y = y + O.mlp(ln(y, "norm2"), sd, "mlp.")
        close(y, golden[tag], tag)


@torch.no_grad()
def test_style_transformer_parts(golden):
    sd, (a, b, c, d), fc, fs, cfg = G.st_small()
    s, m = O.sigma_mu_attention(a, b, c, d, sd, "decoder.decoder_MHA_for_sigma_and_mu.", (7, 7), (4, 4), G.NH)
    close(s, golden["sigma_mu_sigma"], "sigma"); close(m, golden["sigma_mu_mu"], "mu")
    for flag in (True, False):
        outs = O.style_encoder(a, b, c, sd, dict(cfg, encoder_if_use_processed_Key_in_Scale_and_Shift_calculation=flag))
        for n, o in zip(("key", "scale", "shift"), outs):
            close(o, golden[f"style_encoder_flag{int(flag)}_{n}"], n)
    close(O.style_decoder(a, b, c, d, sd, cfg), golden["style_decoder"], "style_decoder")
    for k in (1, 3):
        close(O.style_transformer(fc, fs, k, sd, cfg), golden[f"style_transformer_small_k{k}"], f"st_small_k{k}")


@torch.no_grad()
def test_style_transformer_256_and_decoder(golden):
    sd, fc, fs = G.st_256()
    y1 = O.style_transformer(fc, fs, 1, sd)
    close(y1, golden["style_transformer_256_k1"], "st256_k1")
    close(O.style_transformer(fc, fs, 3, sd), golden["style_transformer_256_k3"], "st256_k3")
    for hw in (8, 32):
        dsd, x = G.decoder_case(hw)
        close(O.image_decoder(x, dsd), golden[f"decoder_{hw}"], f"decoder_{hw}")
    close(O.image_decoder(y1.permute(0, 3, 1, 2), dsd), golden["st256_k1_then_decoder"], "st256_then_decoder")


@torch.no_grad()
def test_fused_qkv_swin_attention(golden):
    from oracle.seeded import seeded_input, seeded_state_dict
    sd = seeded_state_dict(G.attn_shapes(), seed=12)
    fused = {"qkv.weight": torch.cat([sd["Wq.weight"], sd["Wk.weight"], sd["Wv.weight"]]),
             "qkv.bias": torch.cat([sd["Wq.bias"], sd["Wk.bias"], sd["Wv.bias"]]),
             "proj.weight": sd["proj.weight"], "proj.bias": sd["proj.bias"],
             "relative_position_bias_table": sd["relative_position_bias_table"],
             "relative_position_index": sd["relative_position_index"]}
    x = seeded_input("fused_x", (2, 10, 10, G.DIM), 12)
    close(O.swin_window_attention(x, fused, "", (7, 7), (3, 3), G.NH), golden["swin_fused_qkv_H10_s33"], "fused")


@torch.no_grad()
def test_swin_encoder_shapes_and_determinism():
    from oracle.seeded import seeded_input, seeded_state_dict
    sd = seeded_state_dict(O.swin_shapes(), seed=21)
    img = seeded_input("img", (1, 3, 64, 64), 21, kind="image")
    y = O.swin_encoder(img, sd)
    assert y.shape == (1, 8, 8, 256) and torch.isfinite(y).all()
    assert torch.equal(y, O.swin_encoder(img, sd))


@torch.no_grad()
def test_non_default_decoder_branches():
    """affine InstanceNorm, regular-MHA decoder end, Key InstanceNorm before the linear map: oracle vs the reference's outputs
    (tests/golden/golden_branches.npz, written by make_golden_branches.py)"""
    import os, sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden_branches as MB
    gold = np.load(os.path.join(here, "golden", "golden_branches.npz"))
    for i, (tag, *_ ) in enumerate(MB.CASES):
        sd, fc, fs, cfg = MB.case(i)
        for k in (1, 2):
            close(O.style_transformer(fc, fs, k, sd, cfg), gold[f"{tag}_k{k}"], f"{tag}_k{k}")


# ---- the torchvision stage (Swin-B features[:4]) pinned against an independent implementation -------------------------------------------
def _swin_hf_fixture():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_swin_hf.npz"))
    cases = [c.split(":") for c in z["cases"].tolist()]
    return z, [(n, int(b), int(h), int(w), int(s)) for n, b, h, w, s in cases]


@torch.no_grad()
def test_swin_encoder_matches_the_transformers_fixture():
    """oracle.swin_encoder (the restatement of torchvision's swin_b.features[:4], which is absent here) against hidden states produced by
    Hugging Face transformers' SwinModel - an independent implementation of the same architecture - loaded with the same seeded weights
    (tests/golden/make_golden_swin_hf.py): five geometries incl. maps smaller than two windows, odd maps at the merge, a non-square image"""
    from oracle.seeded import seeded_input, seeded_state_dict
    z, cases = _swin_hf_fixture()
    sd = seeded_state_dict(O.swin_shapes(), int(z["weights_seed"]))
    assert len(cases) == 5
    for name, B, H, W, seed in cases:
        img = seeded_input("swin_hf_" + name, (B, 3, H, W), seed, kind="image")
        close(O.swin_encoder(img, sd), z["out_" + name], f"swin encoder vs transformers {z['transformers_version']} ({name})")


@torch.no_grad()
def test_swin_fixture_is_what_transformers_computes():
    """the committed fixture against a live run of the generating code, where `transformers` is installed (it is in this image): the fixture
    is not a copy of the oracle's own output"""
    import pytest
    pytest.importorskip("transformers")
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_swin_hf as M
    from oracle.seeded import seeded_input, seeded_state_dict
    z, cases = _swin_hf_fixture()
    tv = seeded_state_dict(O.swin_shapes(), int(z["weights_seed"]))
    m, _ = M.build_hf_model(tv)
    # the model's own index buffer (computed by transformers) equals the oracle's relative_position_index
    idx = [b for n, b in m.named_buffers() if n.endswith("relative_position_index")]
    assert idx and all(torch.equal(b.reshape(-1).long(), O.relative_position_index((7, 7)).reshape(-1).long()) for b in idx)
    for name, B, H, W, seed in cases[:3]:
        img = seeded_input("swin_hf_" + name, (B, 3, H, W), seed, kind="image")
        close(M.hf_features(m, img), z["out_" + name], f"transformers live vs fixture ({name})")


def test_swin_b_stochastic_depth_schedule_matches_transformers():
    """train-mode StochasticDepth of the (frozen, but train-mode) encoder: torchvision's swin_b spreads stochastic_depth_prob = 0.5 linearly
    over its 24 blocks (block n: 0.5 n / 23); transformers' SwinModel with drop_path_rate = 0.5 and Swin-B's depths builds the same schedule
    independently - the first four blocks are the ones features[:4] runs"""
    import pytest
    pytest.importorskip("transformers")
    from transformers import SwinConfig
    from transformers.models.swin.modeling_swin import SwinEncoder
    cfg = SwinConfig(image_size=224, patch_size=4, embed_dim=8, depths=[2, 2, 18, 2], num_heads=[1, 2, 4, 8], window_size=7, drop_path_rate=0.5)
    enc = SwinEncoder(cfg, (56, 56))
    probs = []
    for layer in enc.layers[:2]:
        for blk in layer.blocks:
            dp = blk.drop_path
            probs.append(float(getattr(dp, "drop_prob", 0.0) or 0.0))
    assert len(probs) == 4 and all(abs(a - b) < 1e-7 for a, b in zip(probs, O.SWIN_B_SD_PROBS)), (probs, O.SWIN_B_SD_PROBS)
