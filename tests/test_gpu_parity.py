"""GPU parity: every HIP kernel and every module of the stylization path against the CPU oracle, on
the same seeded inputs, through the C ABI.  fp32 mode bar: <= 1e-3 max-abs (north star), in practice
the kernel-level checks hold 1e-4 relative."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import master_oracle as O
from oracle.seeded import seeded_input, seeded_state_dict
from tests import golden_cases as G

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-3            # BASELINE.json: <= 1e-3 max-abs vs the CPU path, fp32


def assert_close(got, ref, name, tol=ABS_TOL, absolute=False):
    """kernel / module tests: tol relative to max(1, max|ref|).  absolute=True (the full-forward tests): the north star's bar as written,
    |d| <= tol (1e-3) ABSOLUTE, wherever the stylized image is O(1) (max|ref| <= 10: every k = 1 case); an output in the thousands (the
    seeded weights at k = 3) cannot carry an absolute 1e-3 in fp32 at all, there the bar is 1e-4 of max|ref| (r04, VERDICT r03 item 2:
    measured errors are 1.3e-5 ... 8.5e-5, so the literal bar holds with room)."""
    got = got.detach().float().cpu()
    ref = ref if isinstance(ref, torch.Tensor) else torch.from_numpy(ref)
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = (got - ref).abs().max().item()
    top = ref.abs().max().item()
    bound = (tol if top <= 10.0 else 0.1 * tol * top) if absolute else tol * max(1.0, top)
    assert err <= bound, f"{name}: max|d| = {err:.3e} > {bound:.3e}"
    return err


def load(mod, sd, dev):
    mod.load_state_dict(sd)
    return mod.to(dev).eval()


# ------------------------------------------------------------------ kernels through the C ABI ----
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 256, 256), (1024, 768, 256), (257, 1024, 256), (512, 256, 1024),
                                   (200, 64, 64), (130, 32, 96), (64, 3, 64), (4096, 384, 128)])
def test_linear_shapes(cuda, M, N, K):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    out = ops.linear(a.to(cuda), w.to(cuda), b.to(cuda))
    assert_close(out, F.linear(a, w, b), f"linear {M}x{N}x{K}", 1e-4)


def test_linear_epilogues_and_strides(cuda):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 333, 256, 256
    big = torch.randn(M, 3 * K, generator=g)
    a = big[:, K:2 * K]                                        # strided view, lda = 3K
    w, b = torch.randn(N, K, generator=g) / 16, torch.randn(N, generator=g)
    r1, r2 = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    ad = big.to(cuda)[:, K:2 * K]
    wd, bd = w.to(cuda), b.to(cuda)
    assert_close(ops.linear(ad, wd, bd, act=ops.ACT_GELU), F.gelu(F.linear(a, w, b)), "gelu", 1e-4)
    assert_close(ops.linear(ad, wd, bd, r2=r2.to(cuda)), F.linear(a, w, b) + r2, "residual", 1e-4)
    assert_close(ops.linear(ad, wd, bd, r1=r1.to(cuda), r2=r2.to(cuda)), r1 * F.linear(a, w, b) + r2, "mul-add", 1e-4)
    assert_close(ops.linear(ad, wd, None), F.linear(a, w), "no bias", 1e-4)
    assert_close(ops.linear(ad, wd, bd, act=ops.ACT_RELU), F.relu(F.linear(a, w, b)), "relu", 1e-4)


@pytest.mark.parametrize("B,H,W,Cin,Cout,up,relu,nchw", [(2, 8, 8, 256, 128, False, True, False), (1, 8, 8, 128, 128, True, True, False),
                                                          (2, 5, 7, 64, 64, True, True, False), (1, 16, 16, 64, 32, False, True, False),
                                                          (2, 12, 10, 32, 3, False, False, True), (1, 2, 2, 32, 32, False, False, False),
                                                          # <= 4 output channels: the direct convolution (16 x 16 pixel tiles with a halo in LDS): partial
                                                          # tiles, upsample folded into the tile gather, NHWC and NCHW outputs, 1..4 channels
                                                          (1, 40, 24, 32, 3, True, True, False), (2, 17, 33, 64, 4, False, True, True),
                                                          (1, 16, 16, 32, 1, False, False, False), (3, 5, 3, 96, 2, True, False, True)])
def test_conv3x3_reflect(cuda, B, H, W, Cin, Cout, up, relu, nchw):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(H * 31 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(F.pad(xi, (1, 1, 1, 1), mode="reflect"), w, b)
    ref = F.relu(ref) if relu else ref
    out = ops.conv3x3_reflect(x.permute(0, 2, 3, 1).contiguous().to(cuda), w.permute(0, 2, 3, 1).contiguous().to(cuda),
                              b.to(cuda), up, relu, out_nchw=nchw)
    assert_close(out if nchw else out.permute(0, 3, 1, 2), ref, "conv3x3", 1e-4)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 20, 12, 32, 3), (1, 16, 48, 64, 4), (1, 9, 9, 64, 128)])
def test_conv3x3_zero_padding(cuda, B, H, W, Cin, Cout):
    """nn.Conv2d(padding=1) + ReLU (the VGG-19 layers of the loss network) incl. the <= 4-channel direct form with zero rows for
    out-of-image taps"""
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(H * 13 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    out = ops.conv3x3_zero(x.permute(0, 2, 3, 1).contiguous().to(cuda), w.permute(0, 2, 3, 1).contiguous().to(cuda), b.to(cuda), relu=True)
    assert_close(out.permute(0, 3, 1, 2), F.relu(F.conv2d(x, w, b, padding=1)), "conv3x3 zero pad", 1e-4)


@pytest.mark.parametrize("M,C", [(100, 128), (33, 256), (7, 512), (5, 1024), (64, 64)])
def test_layernorm(cuda, M, C):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(C)
    x, w, b = torch.randn(M, C, generator=g) * 3 + 1, torch.randn(C, generator=g), torch.randn(C, generator=g)
    assert_close(ops.layernorm(x.to(cuda), w.to(cuda), b.to(cuda)), F.layer_norm(x, (C,), w, b, 1e-5), "layernorm", 1e-5)


@pytest.mark.parametrize("B,T,C,n_pad", [(2, 81, 64, 0), (3, 1024, 256, 0), (2, 81, 64, 115), (1, 1024, 256, 201),
                                          (2, 4096, 256, 0), (1, 4096, 256, 804), (40, 2000, 256, 0)])   # streaming form: 16- and 32-channel strips
def test_instnorm(cuda, B, T, C, n_pad):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(T + n_pad)
    x = torch.randn(B, T, C, generator=g) * 2 + 0.5
    row = torch.randn(C, generator=g)
    full = torch.cat([x, row.expand(B, n_pad, C)], dim=1) if n_pad else x
    ref = F.instance_norm(full.permute(0, 2, 1), eps=1e-5).permute(0, 2, 1)
    out, pad = ops.instnorm(x.reshape(B * T, C).to(cuda), B, T, row.to(cuda) if n_pad else None, n_pad)
    assert_close(out.view(B, T, C), ref[:, :T], "instnorm", 1e-5)
    if n_pad:
        assert_close(pad, ref[:, T], "instnorm pad row", 1e-5)
    else:
        # InstanceNorm applied twice in one launch (reference :1056 then :468)
        ref2 = F.instance_norm(ref.permute(0, 2, 1), eps=1e-5).permute(0, 2, 1)
        out2, _ = ops.instnorm(x.reshape(B * T, C).to(cuda), B, T, repeat=2)
        assert_close(out2.view(B, T, C), ref2, "instnorm x2", 1e-5)


def test_patch_embed_and_merge(cuda):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 3, 64, 96, generator=g)
    w, b = torch.randn(128, 3, 4, 4, generator=g) / 7, torch.randn(128, generator=g)
    gw, gb = torch.randn(128, generator=g), torch.randn(128, generator=g)
    ref = F.layer_norm(F.conv2d(img, w, b, stride=4).permute(0, 2, 3, 1), (128,), gw, gb, 1e-5)
    out = ops.patch_embed(img.to(cuda), w.reshape(128, 48).to(cuda), b.to(cuda), gw.to(cuda), gb.to(cuda), torch.float32)
    assert_close(out, ref, "patch_embed", 1e-4)
    x = torch.randn(2, 6, 10, 128, generator=g)
    nw, nb = torch.randn(512, generator=g), torch.randn(512, generator=g)
    cat = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    assert_close(ops.patch_merge_ln(x.to(cuda), nw.to(cuda), nb.to(cuda)), F.layer_norm(cat, (512,), nw, nb, 1e-5), "patch_merge", 1e-5)


# ------------------------------------------------------------------ modules vs golden vectors ----
def test_window_attention_golden(cuda, golden):
    from mastermetastyletransfer_amd import ShiftedWindowAttention
    for H, W, shift in G.SWA_CASES:
        tag, sd, (q, k, v) = G.swa_case(H, W, shift)
        m = load(ShiftedWindowAttention(G.DIM, G.NH, [7, 7], list(shift)), sd, cuda)
        assert_close(m(q.to(cuda), k.to(cuda), v.to(cuda)), golden[tag], tag, 1e-4)
        x = q.to(cuda)                                        # fused-QKV path (same tensor object three times)
        assert_close(m(x, x, x), O.shifted_window_attention(q, q, q, sd, "", (7, 7), shift, G.NH), tag + " fused", 1e-4)


def test_blocks_golden(cuda, golden):
    from mastermetastyletransfer_amd import StyleSwinTransformerBlock
    from torch import nn
    for norm, excl in G.BLOCK_CASES:
        tag, sd, (q, k, v) = G.block_case(norm, excl)
        m = load(StyleSwinTransformerBlock(G.DIM, G.NH, [7, 7], [4, 4], norm_layer=nn.LayerNorm if norm else None,
                                           exclude_MLP_after=excl), sd, cuda)
        assert_close(m(q.to(cuda), k.to(cuda), v.to(cuda), calculating_Key_in_encoder=False), golden[tag], tag, 1e-4)


def build_st(dim, nh, sd, dev, flag=True):
    from mastermetastyletransfer_amd import StyleTransformer
    return load(StyleTransformer(dim, dim, nh, nh, [7, 7], [7, 7], [4, 4], [4, 4],
                                 encoder_if_use_processed_Key_in_Scale_and_Shift_calculation=flag), sd, dev)


def test_style_transformer_parts_golden(cuda, golden):
    sd, abcd, fc, fs, cfg = G.st_small()
    a, b, c, d = (t.to(cuda) for t in abcd)
    m = build_st(G.DIM, G.NH, sd, cuda)
    s, mu = m.decoder.decoder_MHA_for_sigma_and_mu(a, b, c, d)
    assert_close(s, golden["sigma_mu_sigma"], "sigma", 1e-4)
    assert_close(mu, golden["sigma_mu_mu"], "mu", 1e-4)
    for flag in (True, False):
        outs = build_st(G.DIM, G.NH, sd, cuda, flag).encoder(a, b, c)
        for n, o in zip(("key", "scale", "shift"), outs):
            assert_close(o, golden[f"style_encoder_flag{int(flag)}_{n}"], f"encoder flag{flag} {n}", 1e-4)
    # Scale is Shift (first-layer situation): the shared-stream shortcut must equal the general path
    k1, s1, sh1 = m.encoder(a, b, b)
    ok, os_, osh = O.style_encoder(abcd[0], abcd[1], abcd[1], sd, cfg)
    assert_close(s1, os_, "encoder shared scale", 1e-4); assert_close(sh1, osh, "encoder shared shift", 1e-4)
    assert_close(m.decoder(a, b, c, d), golden["style_decoder"], "style_decoder", 2e-4)
    for k in (1, 3):
        assert_close(m(fc.to(cuda), fs.to(cuda), k), golden[f"style_transformer_small_k{k}"], f"st small k{k}", 3e-4)


def test_style_transformer_256_and_decoder_golden(cuda, golden):
    from mastermetastyletransfer_amd import Decoder
    sd, fc, fs = G.st_256()
    m = build_st(256, 8, sd, cuda)
    y1 = m(fc.to(cuda), fs.to(cuda), 1)
    assert_close(y1, golden["style_transformer_256_k1"], "st256 k1")
    assert_close(m(fc.to(cuda), fs.to(cuda), 3), golden["style_transformer_256_k3"], "st256 k3")
    for hw in (8, 32):
        dsd, x = G.decoder_case(hw)
        dec = load(Decoder(256), dsd, cuda)
        assert_close(dec(x.to(cuda)), golden[f"decoder_{hw}"], f"decoder {hw}")
    assert_close(dec(y1.permute(0, 3, 1, 2)), golden["st256_k1_then_decoder"], "st256 -> decoder")


# ------------------------------------------------------------------ Swin encoder + full forward ----
def full_model(dev, seed=31):
    from mastermetastyletransfer_amd import MasterStyleTransferModel
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True)
    sds = (seeded_state_dict(O.swin_shapes(), seed), seeded_state_dict(O.style_transformer_shapes(), seed + 1),
           seeded_state_dict(O.decoder_shapes(), seed + 2))
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    return m.to(dev).eval(), sds


def small_swin_model(dev, variant, seed=41):
    """swin_T / swin_S: embed 96 -> 192-channel tokens, 3 / 6 heads; the StyleTransformer and the decoder are built 192 wide with 6 heads"""
    from mastermetastyletransfer_amd import MasterStyleTransferModel
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(swin_variant=variant, style_encoder_dim=192, style_decoder_dim=192, style_encoder_num_heads=6,
                                     style_decoder_num_heads=6, style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7],
                                     allow_random_swin_encoder=True)
    sds = (seeded_state_dict(O.swin_shapes(embed=96, heads=(3, 6)), seed), seeded_state_dict(O.style_transformer_shapes(dim=192, heads=6), seed + 1),
           seeded_state_dict(O.decoder_shapes(192), seed + 2))
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    return m.to(dev).eval(), sds


@pytest.mark.parametrize("variant,B,S,k", [("swin_T", 2, 64, 1), ("swin_S", 1, 256, 1), ("swin_T", 1, 200, 2)])
def test_swin_t_s_full_forward_vs_oracle(cuda, variant, B, S, k):
    """codes/utils.py:59-88 offers swin_T and swin_S next to swin_B: 96-wide embedding, 3 / 6 heads, 192-channel tokens into a 192-wide
    StyleTransformer and decoder (whose 48- and 24-channel layers run zero-padded to 64 / 32).  fp32 mode; the bf16 mode is swin_B's."""
    m, (swin_sd, st_sd, dec_sd) = small_swin_model(cuda, variant)
    cfg = dict(O.DEFAULT_ST_CFG, encoder_num_heads=6, decoder_num_heads=6)
    c = seeded_input(f"ct{S}", (B, 3, S, S), 8, kind="image")
    s = seeded_input(f"st{S}", (B, 3, S, S), 9, kind="image")
    with torch.no_grad():
        got = m(c.to(cuda), s.to(cuda), k)
        ref = O.full_forward(c, s, k, swin_sd, st_sd, dec_sd, cfg=cfg, swin_heads=(3, 6))
        assert_close(m.swin_encoder(c.to(cuda)), O.swin_encoder(c, swin_sd, heads=(3, 6)), f"{variant} encoder", 2e-4)
    assert got.shape == ref.shape and got.dtype == torch.float32
    assert_close(got, ref, f"{variant} full forward B{B} S{S} k{k}")
    # bf16 mode: the 96-wide first stage stays in fp32 (the bf16 GEMMs step K in 64s), everything from the patch merging on runs on
    # the generic bf16 kernels; same tolerance as swin_B's bf16 mode
    m.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        got16 = m(c.to(cuda), s.to(cuda), k).cpu()
    l2 = float((got16 - ref).norm() / ref.norm())
    assert torch.isfinite(got16).all() and l2 < 2.5e-2, l2
    m.set_compute_dtype("f16x3")                                   # the split mode on the 96 / 192-wide variants (generic split GEMMs; r05)
    with torch.no_grad():
        assert_close(m(c.to(cuda), s.to(cuda), k), ref, f"{variant} f16x3")
    m.set_compute_dtype(torch.float32)
    with torch.no_grad():
        assert_close(m(c.to(cuda), s.to(cuda), k), ref, f"{variant} back in fp32")


def test_swin_t_width_mismatch_raises(cuda):
    """a swin_T encoder in front of the default 256-wide StyleTransformer fails like the reference's first nn.Linear would"""
    from mastermetastyletransfer_amd import MasterStyleTransferModel
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(swin_variant="swin_T", allow_random_swin_encoder=True).to(cuda).eval()
    x = torch.rand(1, 3, 64, 64, device=cuda)
    with pytest.raises(RuntimeError, match="shapes cannot be multiplied"), torch.no_grad():
        m(x, x, 1)


@pytest.mark.parametrize("B,S", [(1, 64), (2, 256)])
def test_swin_encoder_vs_oracle(cuda, B, S):
    m, (swin_sd, _, _) = full_model(cuda)
    img = seeded_input(f"enc{S}", (B, 3, S, S), 7, kind="image")
    with torch.no_grad():
        assert_close(m.swin_encoder(img.to(cuda)), O.swin_encoder(img, swin_sd), f"swin encoder {S}", 2e-4)


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-5), ("f16x3", 1e-5), ("bf16", None)])
def test_swin_encoder_vs_transformers_fixture(cuda, mode, tol):
    """the HIP Swin-B features[:4] stage against hidden states of Hugging Face transformers' SwinModel on the same seeded weights
    (tests/golden/golden_swin_hf.npz, made by tests/golden/make_golden_swin_hf.py): an implementation of the published architecture that
    is independent of this repository and of its oracle - the pin for the stage whose own dependency (torchvision) is absent.
    Five geometries: 64^2 (maps smaller than two windows), 128^2 x 2, 200^2 (odd map at the merge), 256^2, 96 x 160."""
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_swin_hf.npz"))
    m, _ = full_model(cuda)
    sd = seeded_state_dict(O.swin_shapes(), int(z["weights_seed"]))
    m.swin_encoder.load_state_dict(sd)
    m.set_compute_dtype({"fp32": torch.float32, "f16x3": "f16x3", "bf16": torch.bfloat16}[mode])
    for c in z["cases"].tolist():
        name, B, H, W, seed = c.split(":")
        img = seeded_input("swin_hf_" + name, (int(B), 3, int(H), int(W)), int(seed), kind="image")
        ref = torch.from_numpy(z["out_" + name])
        with torch.no_grad():
            got = m.swin_encoder(img.to(cuda)).float().cpu()
        if tol is not None:
            err = assert_close(got, ref, f"swin encoder [{mode}] vs transformers ({name})", tol)
            print(f"swin encoder [{mode}] vs transformers {z['transformers_version']} ({name}): max|d| {err:.2e}, max|ref| {float(ref.abs().max()):.2f}")
        else:                                  # bf16 throughput mode: relative L2 (operand rounding through 4 blocks)
            rel = float((got - ref).norm() / ref.norm())
            assert torch.isfinite(got).all() and rel <= 2e-2, f"{name}: rel L2 {rel:.3e}"
            print(f"swin encoder [bf16] vs transformers ({name}): rel L2 {rel:.2e}")


@pytest.mark.parametrize("H,W", [(250, 254), (66, 70), (33, 47)])
def test_full_forward_sides_not_multiples_of_4(cuda, H, W):
    """the 4x4 / stride-4 patch embedding has no padding: it ignores the trailing H % 4 rows and W % 4 columns, the reference accepts such
    images and returns 8 * ceil(floor(side / 4) / 2) pixels per side (250 x 254 -> 248 x 256); so does the HIP path (r05: it used to refuse them)"""
    m, (swin_sd, st_sd, dec_sd) = full_model(cuda)
    c = seeded_input("c_n4", (1, 3, H, W), 8, kind="image")
    s = seeded_input("s_n4", (1, 3, H, W), 9, kind="image")
    with torch.no_grad():
        got = m(c.to(cuda), s.to(cuda), 1)
        ref = O.full_forward(c, s, 1, swin_sd, st_sd, dec_sd)
    assert got.shape == ref.shape == (1, 3, 8 * (((H // 4) + 1) // 2), 8 * (((W // 4) + 1) // 2))
    assert_close(got, ref, f"full forward {H}x{W}", absolute=True)
    with pytest.raises(RuntimeError), torch.no_grad():
        m(c[:, :, :3].to(cuda), s[:, :, :3].to(cuda), 1)


@pytest.mark.parametrize("B,H,W,k", [(3, 200, 200, 1), (5, 72, 104, 2), (7, 40, 56, 1)])
def test_ragged_batches_in_every_mode(cuda, B, H, W, k):
    """token counts per image that are not multiples of the kernels' 128-token tiles (625, 117, 35): tiles cross image boundaries, per-image
    norms and windows must not - all three modes against the oracle, and image 1 of the batch bit-identical to the same image alone"""
    m, (swin_sd, st_sd, dec_sd) = full_model(cuda)
    c = seeded_input("c_rag", (B, 3, H, W), 8, kind="image"); s = seeded_input("s_rag", (B, 3, H, W), 9, kind="image")
    with torch.no_grad():
        ref = O.full_forward(c, s, k, swin_sd, st_sd, dec_sd)
        for mode in (torch.float32, "f16x3", torch.bfloat16):
            m.set_compute_dtype(mode)
            got = m(c.to(cuda), s.to(cuda), k)
            one = m(c[1:2].to(cuda), s[1:2].to(cuda), k)
            assert torch.equal(got[1:2], one), f"{mode}: not batch-independent"
            if mode == torch.bfloat16:
                assert float((got.float().cpu() - ref).norm() / ref.norm()) <= 3e-2
            else:
                assert_close(got, ref, f"ragged batch {B}x{H}x{W} k{k} [{mode}]", absolute=True)


def test_concurrent_streams_and_module_copies(cuda, tmp_path):
    """multi-stream serving: four forwards issued on four streams at once (the FIRST use of the model: the derived-weight caches are built by
    whichever stream comes first and carry an event for the others) equal the sequential results bit for bit; and a used model still
    pickles / deep-copies (caches and their events are dropped from the state)"""
    import copy
    m, (swin_sd, st_sd, dec_sd) = full_model(cuda)
    ins = [(seeded_input(f"cs_c{i}", (2, 3, 96, 96), 8 + i, kind="image").to(cuda), seeded_input(f"cs_s{i}", (2, 3, 96, 96), 90 + i, kind="image").to(cuda)) for i in range(4)]
    streams = [torch.cuda.Stream() for _ in ins]
    outs = [None] * 4
    with torch.no_grad():
        for i, (st, (c, s)) in enumerate(zip(streams, ins)):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs[i] = m(c, s, 1)
        torch.cuda.synchronize()
        seq = [m(c, s, 1) for c, s in ins]
        for i in range(4):
            assert torch.equal(outs[i], seq[i]), i
        m2 = copy.deepcopy(m)
        torch.save(m.decoder, tmp_path / "dec.pt")
        assert torch.equal(m2(*ins[0], 1), seq[0])
    ref = O.full_forward(ins[0][0].cpu(), ins[0][1].cpu(), 1, swin_sd, st_sd, dec_sd)
    assert_close(seq[0], ref, "stream test forward", absolute=True)


@pytest.mark.parametrize("ratio", [1.1, 1.5, 0.25])
def test_mlp_ratios_incl_a_hidden_width_that_is_not_a_multiple_of_32(cuda, ratio):
    """style_encoder_mlp_ratio / style_decoder_mlp_ratio other than 4 (full_model.py:34-35): hidden 384 and 64 run on the kernels as they are,
    281 (ratio 1.1) is zero-padded to 320 on the host side - all three modes against the oracle"""
    from mastermetastyletransfer_amd import MasterStyleTransferModel
    sds = (seeded_state_dict(O.swin_shapes(), 31), seeded_state_dict(O.style_transformer_shapes(mlp_ratio=ratio), 32), seeded_state_dict(O.decoder_shapes(), 33))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True,
                                     style_encoder_mlp_ratio=ratio, style_decoder_mlp_ratio=ratio)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(cuda).eval()
    c = seeded_input("c_mr", (2, 3, 96, 64), 8, kind="image"); s = seeded_input("s_mr", (2, 3, 96, 64), 9, kind="image")
    with torch.no_grad():
        ref = O.full_forward(c, s, 1, *sds)
        for mode in (torch.float32, "f16x3", torch.bfloat16):
            got = m.set_compute_dtype(mode)(c.to(cuda), s.to(cuda), 1)
            if mode == torch.bfloat16:
                assert float((got.float().cpu() - ref).norm() / ref.norm()) <= 3e-2
            else:
                assert_close(got, ref, f"mlp_ratio {ratio} [{mode}]", absolute=True)


@pytest.mark.parametrize("ws,sh", [((4, 4), (2, 2)), ((7, 5), (3, 2)), ((3, 9), (1, 4)), ((8, 8), (0, 0))])
def test_window_and_shift_geometries_in_every_mode(cuda, ws, sh):
    """style_encoder / style_decoder window_size and shift_size other than the benchmark's 7 x 7 / 4 (full_model.py:30-33): small, non-square and
    unshifted windows through the whole model in the three modes (any window of at most 64 tokens runs on the same kernels)"""
    from mastermetastyletransfer_amd import MasterStyleTransferModel
    sds = (seeded_state_dict(O.swin_shapes(), 31), seeded_state_dict(O.style_transformer_shapes(ws=ws), 32, ws=ws), seeded_state_dict(O.decoder_shapes(), 33))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=list(ws), style_decoder_window_size=list(ws), style_encoder_shift_size=list(sh),
                                     style_decoder_shift_size=list(sh), allow_random_swin_encoder=True)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(cuda).eval()
    c = seeded_input("c_ws", (2, 3, 96, 128), 8, kind="image"); s = seeded_input("s_ws", (2, 3, 96, 128), 9, kind="image")
    cfg = dict(O.DEFAULT_ST_CFG, encoder_window_size=ws, decoder_window_size=ws, encoder_shift_size=sh, decoder_shift_size=sh)
    with torch.no_grad():
        ref = O.full_forward(c, s, 1, *sds, cfg=cfg)
        for mode in (torch.float32, "f16x3", torch.bfloat16):
            got = m.set_compute_dtype(mode)(c.to(cuda), s.to(cuda), 1)
            if mode == torch.bfloat16:
                assert float((got.float().cpu() - ref).norm() / ref.norm()) <= 3e-2
            else:
                assert_close(got, ref, f"windows {ws} shift {sh} [{mode}]", absolute=True)


def test_empty_batch_is_an_error_not_a_crash(cuda):
    """B = 0: the CPU path cannot run it either (a reshape with an ambiguous -1 in the window partition); the HIP path must answer with an
    exception - no launch with an empty grid, no device fault - and stay usable afterwards"""
    m, (swin_sd, st_sd, dec_sd) = full_model(cuda)
    c = seeded_input("c64", (1, 3, 64, 64), 8, kind="image")
    with pytest.raises(Exception) as ei, torch.no_grad():
        m(c[0:0].to(cuda), c[0:0].to(cuda), 1)
    assert not isinstance(ei.value, (SystemExit, KeyboardInterrupt))
    with torch.no_grad():
        got = m(c.to(cuda), c.to(cuda), 1)
        ref = O.full_forward(c, c, 1, swin_sd, st_sd, dec_sd)
    assert_close(got, ref, "forward after the refused empty batch", absolute=True)


def test_full_forward_odd_feature_map(cuda):
    """image sides that are multiples of 4 but not of 8 (260 x 132 -> 65 x 33 tokens): torchvision's PatchMerging pads the map to even
    sides with zeros; the output is 8 * ceil(side / 8) like the reference's"""
    m, (swin_sd, st_sd, dec_sd) = full_model(cuda)
    c = seeded_input("c_odd", (1, 3, 260, 132), 8, kind="image")
    s = seeded_input("s_odd", (1, 3, 260, 132), 9, kind="image")
    with torch.no_grad():
        got = m(c.to(cuda), s.to(cuda), 1)
        ref = O.full_forward(c, s, 1, swin_sd, st_sd, dec_sd)
    assert got.shape == ref.shape == (1, 3, 264, 136)
    assert_close(got, ref, "full forward 260x132", absolute=True)


@pytest.mark.parametrize("B,S,k", [(1, 256, 1), (2, 256, 1), (1, 256, 3), (1, 512, 1), (1, 64, 1), (1, 200, 1)])
def test_full_forward_vs_oracle(cuda, B, S, k):
    """BASELINE configs[0] and [4] shapes: same content/style pair, same weights, <= 1e-3 max-abs."""
    m, (swin_sd, st_sd, dec_sd) = full_model(cuda)
    c = seeded_input(f"c{S}", (B, 3, S, S), 8, kind="image")
    s = seeded_input(f"s{S}", (B, 3, S, S), 9, kind="image")
    with torch.no_grad():
        got = m(c.to(cuda), s.to(cuda), k)
        ref = O.full_forward(c, s, k, swin_sd, st_sd, dec_sd)
    assert got.shape == (B, 3, S, S) and got.dtype == torch.float32
    err = assert_close(got, ref, f"full forward B{B} S{S} k{k}", absolute=True)
    print(f"full forward B={B} S={S} k={k}: max|d| = {err:.2e}, ref max {ref.abs().max():.2f}")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_full_batch_32_properties(cuda, mode):
    """BASELINE configs[1] at its own size AND dtype (bs=32, 256x256, bf16; and the fp32 parity mode): per-sample independence and
    determinism - sample i of the batch equals the same pair run alone (no cross-sample statistics anywhere on the path), bitwise
    repeatable - and samples of the batch against the CPU oracle (fp32: 1e-3 max-abs; bf16: the bench's relative-L2 bound)."""
    m, (swin_sd, st_sd, dec_sd) = full_model(cuda)
    m.set_compute_dtype(torch.float32 if mode == "fp32" else torch.bfloat16)
    c = seeded_input("c32", (32, 3, 256, 256), 10, kind="image").to(cuda)
    s = seeded_input("s32", (32, 3, 256, 256), 11, kind="image").to(cuda)
    with torch.no_grad():
        full = m(c, s, 1)
        again = m(c, s, 1)
        assert torch.equal(full, again), "forward is not deterministic"
        for i in (0, 13, 31):
            one = m(c[i:i + 1], s[i:i + 1], 1)
            assert torch.equal(one, full[i:i + 1]), f"sample {i} depends on its batch"
        for i in (5, 31):
            ref = O.full_forward(c[i:i + 1].cpu(), s[i:i + 1].cpu(), 1, swin_sd, st_sd, dec_sd)
            if mode == "fp32":
                assert_close(full[i:i + 1], ref, f"sample {i} of the bs=32 batch", absolute=True)
            else:
                l2 = float((full[i:i + 1].cpu() - ref).norm() / ref.norm())
                mx = float((full[i:i + 1].cpu() - ref).abs().max() / ref.abs().max())
                assert l2 < 2.5e-2 and mx < 0.05, (i, l2, mx)


@pytest.mark.parametrize("mode", ["fp32", "x3"])
def test_config5_full_size(cuda, mode):
    """BASELINE configs[4] at its own size: bs = 16, 512 x 512, fp32 (exact-fp32 mode, and the split-bf16 fast parity mode) - bitwise
    determinism, per-sample batch independence, and two samples of the batch against the CPU oracle at the literal 1e-3 max-abs
    (r03 covered this configuration at B = 1 only)."""
    m, (swin_sd, st_sd, dec_sd) = full_model(cuda)
    m.set_compute_dtype(torch.float32 if mode == "fp32" else "bf16x3")
    c = seeded_input("c5", (16, 3, 512, 512), 12, kind="image").to(cuda)
    s = seeded_input("s5", (16, 3, 512, 512), 13, kind="image").to(cuda)
    with torch.no_grad():
        full = m(c, s, 1)
        assert full.shape == (16, 3, 512, 512) and full.dtype == torch.float32
        assert torch.equal(full, m(c, s, 1)), "forward is not deterministic"
        for i in (0, 15):
            assert torch.equal(m(c[i:i + 1], s[i:i + 1], 1), full[i:i + 1]), f"sample {i} depends on its batch"
        for i in (3, 15):
            ref = O.full_forward(c[i:i + 1].cpu(), s[i:i + 1].cpu(), 1, swin_sd, st_sd, dec_sd)
            err = assert_close(full[i:i + 1], ref, f"sample {i} of the bs=16 512x512 batch ({mode})", absolute=True)
            print(f"config 5 {mode} sample {i}: max|d| = {err:.2e}, ref max {ref.abs().max():.2f}")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_graph_replay_matches_eager(cuda, mode):
    """hipGraph capture of the forward (GraphedStylizer): replays are bit-identical to eager calls, also for new inputs"""
    import warnings
    from mastermetastyletransfer_amd import GraphedStylizer, MasterStyleTransferModel
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True)
    m = m.to(cuda).eval().set_compute_dtype(torch.float32 if mode == "fp32" else torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    stylize = GraphedStylizer(m, 2, 64, 96, transformer_layer_count=2)
    for _ in range(3):
        c, s = torch.rand(2, 3, 64, 96, generator=g).to(cuda), torch.rand(2, 3, 64, 96, generator=g).to(cuda)
        with torch.no_grad():
            want = m(c, s, 2)
        got = stylize(c, s, copy=True)
        assert torch.equal(got, want)
    with pytest.raises(ValueError):
        stylize(torch.zeros(1, 3, 64, 96, device=cuda), torch.zeros(1, 3, 64, 96, device=cuda))
    # two captures with alternating output buffers (the multi-GPU bench: the all-gather of step i reads buffer i % 2 in place)
    two = GraphedStylizer(m, 2, 64, 96, transformer_layer_count=2, copies=2)
    assert len(two.outs) == 2 and two.outs[0].data_ptr() != two.outs[1].data_ptr()
    two.content.copy_(c); two.style.copy_(s)
    a = two.replay(0)
    two.content.copy_(c.flip(0)); two.style.copy_(s.flip(0))
    b = two.replay(1)
    torch.cuda.synchronize()
    assert torch.equal(a, want) and torch.equal(b, m(c.flip(0), s.flip(0), 2)) and two.replay(2) is a


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,oh,ow,norm", [(2, 240, 320, 256, 256, True), (1, 300, 200, 256, 256, False), (3, 64, 48, 256, 256, True),
                                              (1, 256, 256, 256, 256, True), (1, 256, 500, 256, 256, True), (2, 1000, 33, 31, 17, True)])
def test_preprocess_bit_exact(cuda, B, H, W, oh, ow, norm):
    """device resize + ToTensor + Normalize == the Pillow-pinned CPU restatement, bit for bit (integer passes, IEEE fp32 epilogue)"""
    from oracle import imageio_oracle as IO
    from mastermetastyletransfer_amd import Preprocess
    rng = np.random.default_rng(H + W)
    imgs = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    got = Preprocess((oh, ow), imagenet_normalization=norm)(torch.from_numpy(imgs).to(cuda)).cpu().numpy()
    want = np.stack([IO.preprocess(imgs[b], oh, ow, normalize=norm) for b in range(B)])
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
def test_postprocess_bit_exact(cuda):
    from oracle import imageio_oracle as IO
    from mastermetastyletransfer_amd import postprocess
    x = torch.randn(3, 3, 37, 53, generator=torch.Generator().manual_seed(2)) * 0.8 + 0.5
    x[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 1.0 - 1e-7, 254.999 / 255])
    got = postprocess(x.to(cuda)).cpu().numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, IO.postprocess(x.numpy()))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [0, 1, 2])
def test_perceptual_loss_forward(cuda, mode, case):
    """custom_loss forward (VGG-19 features + content/style terms) against the reference-pinned CPU restatement and the committed
    outputs of the imported reference loss.py"""
    from oracle import loss_oracle as L
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden_loss import CASES, seeded_vgg_and_images
    from mastermetastyletransfer_amd import custom_loss
    dc, ds, B, H, W = CASES[case]
    sd, (c, s, o) = seeded_vgg_and_images(case, B, H, W)
    crit = custom_loss(vgg_state_dict=sd, distance_content=dc, distance_style=ds,
                       compute_dtype=torch.float32 if mode == "fp32" else torch.bfloat16).to(cuda).eval()
    with torch.no_grad():
        total, cl, sl = crit(c.to(cuda), s.to(cuda), o.to(cuda), output_content_and_style_loss=True)
        want = L.overall_loss(c, s, o, sd, 10.0, dc, ds)
        taps = crit.feature_extractor_model(o.to(cuda))
        ref_taps = L.vgg19_features(o, sd)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_loss.npz"))[f"case{case}"]
    got = np.array([float(total), float(cl), float(sl)])
    tol = 2e-4 if mode == "fp32" else 3e-2
    assert np.allclose(got, [float(v) for v in want], rtol=tol), (got, want)
    assert np.allclose(got, gold, rtol=tol), (got, gold)
    for t, r in zip(taps, ref_taps):
        d = (t.float().permute(0, 3, 1, 2).cpu() - r).abs().max() / r.abs().max()
        assert d < (1e-4 if mode == "fp32" else 3e-2), d
    # the quirks of the reference's signature: the similarity term is identically zero, an explicit lambda is replaced by the default
    with torch.no_grad():
        out4 = crit(c.to(cuda), s.to(cuda), o.to(cuda), lambda_value=3.0, output_content_and_style_loss=True, output_similarity_loss=True)
    assert len(out4) == 4 and float(out4[3]) == 0.0 and abs(float(out4[0]) - float(total)) <= 1e-6 * abs(float(total))


@pytest.mark.gpu
def test_harness_chain_end_to_end(cuda):
    """What test_model.py does around one content/style pair, on the device, against the chained CPU oracles:
    uint8 HWC -> Resize/ToTensor/Normalize -> forward -> custom_loss -> clip*255 -> uint8 (fp32 parity mode)."""
    from oracle import imageio_oracle as IO
    from oracle import loss_oracle as L
    from mastermetastyletransfer_amd import MasterStyleTransferModel, Preprocess, custom_loss, postprocess
    rng = np.random.default_rng(11)
    c8, s8 = rng.integers(0, 256, (1, 150, 200, 3), dtype=np.uint8), rng.integers(0, 256, (1, 90, 120, 3), dtype=np.uint8)
    sds = (seeded_state_dict(O.swin_shapes(), 31), seeded_state_dict(O.style_transformer_shapes(), 32), seeded_state_dict(O.decoder_shapes(), 33))
    vgg = seeded_state_dict(L.vgg19_shapes(), 34)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(cuda).eval()
    crit = custom_loss(vgg_state_dict=vgg).to(cuda).eval()
    pre = Preprocess((64, 64), imagenet_normalization=True)
    with torch.no_grad():
        c, s = pre(torch.from_numpy(c8).to(cuda)), pre(torch.from_numpy(s8).to(cuda))
        out = m(c, s, 1)
        total, cl, sl = crit(c, s, out, output_content_and_style_loss=True)
        img8 = postprocess(out).cpu().numpy()
        # the same chain on the CPU
        rc = torch.from_numpy(IO.preprocess(c8[0], 64, 64))[None]; rs = torch.from_numpy(IO.preprocess(s8[0], 64, 64))[None]
        assert torch.equal(c.cpu(), rc) and torch.equal(s.cpu(), rs)                    # pre-processing is bit-exact
        ro = O.full_forward(rc, rs, 1, *sds)
        rt, rcl, rsl = L.overall_loss(rc, rs, ro, vgg, 10.0)
    assert (out.cpu() - ro).abs().max() < 1e-3
    assert np.allclose([float(total), float(cl), float(sl)], [float(rt), float(rcl), float(rsl)], rtol=5e-4)
    ref8 = IO.postprocess(ro.numpy())
    assert np.abs(img8.astype(int) - ref8.astype(int)).max() <= 1                       # a 1e-3 fp difference can cross a truncation boundary


# ---------------------------------------------------------------------------------------------------------------------------------------
# torchvision pin (Swin-B features[:4] and VGG-19 cfg "E" live in torchvision, which the reference does not vendor or
# pin): where the GPU box has torchvision, the oracle's restatement AND the HIP path are checked against torchvision's own modules
# with the same seeded weights.  torchvision is a third-party library, not the reference; nothing is vendored.  Skips where absent
# (this image has none; the Swin stage is pinned against transformers' independent SwinModel instead: test_swin_encoder_vs_transformers_fixture).
# ---------------------------------------------------------------------------------------------------------------------------------------
def test_swin_encoder_vs_torchvision(cuda):
    tv = pytest.importorskip("torchvision")
    from torchvision.models import swin_b
    net = swin_b(weights=None).features[:4].eval()
    sd = seeded_state_dict(O.swin_shapes(), 1)
    missing = net.load_state_dict(sd, strict=True)
    img = seeded_input("tv_swin", (2, 3, 128, 128), 7, kind="image")
    with torch.no_grad():
        ref = net(img)
        assert_close(O.swin_encoder(img, sd), ref, "oracle swin_encoder vs torchvision", 2e-5)
        m, _ = full_model(cuda)
        m.swin_encoder.load_state_dict(sd)
        assert_close(m.swin_encoder(img.to(cuda)), ref, "HIP swin encoder vs torchvision", 2e-4)


def test_vgg19_vs_torchvision(cuda):
    tv = pytest.importorskip("torchvision")
    from torchvision.models import vgg19
    from oracle import loss_oracle as L
    from mastermetastyletransfer_amd import VGG19_custom
    net = vgg19(weights=None).features[:30].eval()
    sd = seeded_state_dict(L.vgg19_shapes(), 4)
    net.load_state_dict(sd, strict=True)
    img = seeded_input("tv_vgg", (2, 3, 64, 64), 8, kind="image")
    taps_at = (6, 11, 20, 29)
    with torch.no_grad():
        refs, x = [], img
        for i, layer in enumerate(net):
            x = layer(x)
            if i in taps_at:
                refs.append(x.clone())
        for a, b in zip(L.vgg19_features(img, sd), refs):
            assert_close(a, b, "oracle VGG-19 vs torchvision", 2e-5)
        v = VGG19_custom()
        v.features.load_state_dict(sd)
        v = v.to(cuda)
        for a, b in zip(v(img.to(cuda)), refs):
            assert_close(a.permute(0, 3, 1, 2), b, "HIP VGG-19 vs torchvision", 2e-4)


@pytest.mark.parametrize("i", range(5))
@pytest.mark.parametrize("mode", ["fp32", "bf16", "f16x3"])
def test_non_default_decoder_branches(cuda, i, mode):
    """decoder_use_instance_norm_with_affine, decoder_use_regular_MHA_instead_of_Swin_at_the_end and
    decoder_use_Key_instance_norm_after_linear_transformation=False on the HIP path against the reference's own outputs
    (tests/golden/golden_branches.npz): fp32 <= 1e-3 of max|ref| ... bf16 relative L2"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_branches as MB
    from mastermetastyletransfer_amd import StyleTransformer
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_branches.npz"))
    tag, affine, regular, after = MB.CASES[i]
    sd, fc, fs, cfg = MB.case(i)
    m = StyleTransformer(MB.DIM, MB.DIM, MB.NH, MB.NH, [7, 7], [7, 7], [4, 4], [4, 4], decoder_use_instance_norm_with_affine=affine,
                         decoder_use_regular_MHA_instead_of_Swin_at_the_end=regular,
                         decoder_use_Key_instance_norm_after_linear_transformation=after)
    m.load_state_dict(sd)
    m = m.to(cuda).eval()
    for mod in m.modules():
        if hasattr(mod, "compute_dtype"):
            mod.compute_dtype = torch.bfloat16 if mode == "bf16" else torch.float32
            mod.split_weights = "f16" if mode == "f16x3" else False
    for k in (1, 2):
        with torch.no_grad():
            got = m(fc.to(cuda), fs.to(cuda), k).float().cpu()
        ref = torch.from_numpy(gold[f"{tag}_k{k}"])
        if mode in ("fp32", "f16x3"):
            assert_close(got, ref, f"{tag} k={k} [{mode}]", 1e-4)
        else:
            l2 = float((got - ref).norm() / ref.norm())
            assert l2 < 3e-2, (tag, k, l2)


def test_regular_mha_with_affine_norm_raises_like_the_reference(cuda):
    from mastermetastyletransfer_amd import StyleTransformer
    m = StyleTransformer(64, 64, 2, 2, [7, 7], [7, 7], [4, 4], [4, 4], decoder_use_instance_norm_with_affine=True,
                         decoder_use_regular_MHA_instead_of_Swin_at_the_end=True).to(cuda).eval()
    with pytest.raises(ValueError, match="to match num_features"):
        with torch.no_grad():
            m(torch.zeros(2, 8, 8, 64, device=cuda), torch.zeros(2, 8, 8, 64, device=cuda), 1)


@pytest.mark.parametrize("mode", ["fp32", "bf16", "x3"])
def test_bias_free_projections_equal_zero_biases(cuda, mode):
    """qkv_bias = proj_bias = False (full_model.py:39-42): F.linear without a bias adds nothing and a zero-padded window slot's key /
    value row is then zero - the same kernels with a zero bias row: the bias-free model equals, bit for bit, the default model whose
    projection biases are zero and whose other weights are the same."""
    from mastermetastyletransfer_amd import StyleTransformer
    torch.manual_seed(2)
    nb = StyleTransformer(256, 256, 8, 8, [7, 7], [7, 7], [4, 4], [4, 4], encoder_qkv_bias=False, decoder_qkv_bias=False,
                          encoder_proj_bias=False, decoder_proj_bias=False)
    wb = StyleTransformer(256, 256, 8, 8, [7, 7], [7, 7], [4, 4], [4, 4])
    sd = wb.state_dict()
    for k, v in nb.state_dict().items():
        sd[k] = v.clone()
    for k in sd:
        if k.endswith(".bias") and k.split(".")[-2] in ("Wq", "Wk", "Wv", "Wv_scale", "Wv_shift", "proj"):
            sd[k] = torch.zeros_like(sd[k])
    wb.load_state_dict(sd)
    nb, wb = nb.to(cuda).eval(), wb.to(cuda).eval()
    for m in (nb, wb):
        for sub in m.modules():
            if hasattr(sub, "compute_dtype"):
                sub.compute_dtype = torch.bfloat16 if mode == "bf16" else torch.float32
                sub.split_weights = mode == "x3"
    g = torch.Generator().manual_seed(3)
    fc, fs = torch.randn(2, 19, 23, 256, generator=g).to(cuda), torch.randn(2, 19, 23, 256, generator=g).to(cuda)       # padded map: pad rows matter
    with torch.no_grad():
        a, b = nb(fc, fs, 2), wb(fc, fs, 2)
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_relu_mlp_activation(cuda):
    """MLP_activation_layer = nn.ReLU (full_model.py:47-50; torchvision MLP(activation_layer=...)): block output = attention half + W2 relu(W1
    LN2(.) + b1) + b2, composed here from the same block without its MLP and torch arithmetic"""
    from mastermetastyletransfer_amd import StyleSwinTransformerBlock
    from torch import nn
    torch.manual_seed(4)
    blk = StyleSwinTransformerBlock(256, 8, [7, 7], [4, 4], MLP_activation_layer=nn.ReLU)
    half = StyleSwinTransformerBlock(256, 8, [7, 7], [4, 4], exclude_MLP_after=True)
    half.load_state_dict({k: v for k, v in blk.state_dict().items() if k in half.state_dict()})
    blk, half = blk.to(cuda).eval(), half.to(cuda).eval()
    x = torch.randn(2, 14, 14, 256, generator=torch.Generator().manual_seed(5)).to(cuda)
    with torch.no_grad():
        got = blk(x, x, x)
        a = half(x, x, x).cpu().double()
        w = {k: v.cpu().double() for k, v in blk.state_dict().items()}
        h = F.relu(F.linear(F.layer_norm(a, (256,), w["norm2.weight"], w["norm2.bias"], 1e-5), w["mlp.0.weight"], w["mlp.0.bias"]))
        ref = a + F.linear(h, w["mlp.3.weight"], w["mlp.3.bias"])
    assert_close(got, ref.float(), "block with a ReLU MLP", 1e-4)
    for mode in ("bf16", "x3"):
        for sub in blk.modules():
            if hasattr(sub, "compute_dtype"):
                sub.compute_dtype = torch.bfloat16 if mode == "bf16" else torch.float32
                sub.split_weights = mode == "x3"
        with torch.no_grad():
            got = blk(x, x, x)
        assert_close(got, ref.float(), f"block with a ReLU MLP ({mode})", 3e-2 if mode == "bf16" else 1e-4)
