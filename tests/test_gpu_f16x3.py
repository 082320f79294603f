"""GPU parity of the split fast parity mode with IEEE-HALF halves ("f16x3", MASTER_F16X3; r05): the kernels of tests/test_gpu_x3.py with
hi = half(x), lo = half(x - hi) instead of bf16 halves.  hi + lo carries 22 significant bits instead of 17, so one contraction sits at
fp32-accumulation level from float64 (bound here: 4e-6 of max|ref|, the bf16 halves' bound is 3e-5) and the stylized image at the
exact-fp32 mode's distance from the CPU oracle: the full-forward tests hold the north star's literal 1e-3 AND a 3e-4 bound that the bf16
halves do not meet on these pairs (profiles/r05_x3_split_sim.txt).  The multi-seed / multi-size sweep ADVICE r04 asked for is here, for
both formats."""
import warnings

import pytest
import torch
import torch.nn.functional as F

from oracle import master_oracle as O
from oracle.seeded import seeded_input, seeded_state_dict

pytestmark = pytest.mark.gpu

F16_KERNEL_TOL = 4e-6          # one contraction, relative to max|ref| (fp32 accumulation of K <= 2304 products + 2^-22 operand residuals)
ABS_TOL = 1e-3                 # the north star's bar on the stylized image, absolute
F16_IMAGE_TOL = 3e-4           # what the half halves actually hold on outputs of max-abs <= 20 (measured <= 1.3e-4; exact-fp32 mode: <= 1.2e-4)


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.double()
    assert got.shape == ref.shape, (tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got).all()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def sw(w, layout=0):
    from mastermetastyletransfer_amd import ops
    return ops.split_weights(w, layout=layout, fmt="f16")


def test_split_weights_half_layout_bit_exact(cuda):
    """hi = half(w), lo = half(w - hi) (round to nearest even, subnormal results kept), per 32-wide k chunk 32 hi then 32 lo; layout 1"""
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn(37, 96, generator=g) * torch.logspace(-7, 4, 96)[None, :]
    w[0, :6] = torch.tensor([0.0, -0.0, 1e-30, 65504.0, 6.0e-8, -3.1e-5])
    got = sw(w.to(cuda)).cpu()
    hi = w.half()
    lo = (w - hi.float()).half()
    ref = torch.stack([hi.view(37, 3, 32), lo.view(37, 3, 32)], dim=2)
    assert got.shape == (37, 3, 2, 32) and got.dtype == torch.float16
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    assert ops.is_split(got) and ops.x3_code(got) == ops.F16X3 and ops.x3_code(got.view(torch.bfloat16)) == ops.F32X3
    # hi + lo reproduces w to 2^-21 of its magnitude or to the subnormal floor 2^-24
    rec = (got[:, :, 0].float() + got[:, :, 1].float()).view(37, 96)
    assert bool(((rec - w).abs() <= torch.maximum(w.abs() * 2.0 ** -21, torch.tensor(2.0 ** -24))).all())
    w1 = torch.randn(5, 64, generator=g)
    perm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
    assert torch.equal(sw(w1.to(cuda), layout=1).cpu().view(torch.int16), sw(w1.view(5, 4, 16)[:, :, perm].reshape(5, 64).to(cuda)).cpu().view(torch.int16))


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 256, 256), (1024, 768, 256), (512, 256, 1024), (200, 64, 64), (130, 32, 96),
                                   (70000, 128, 128), (66000, 256, 512), (33, 20, 32), (64, 3, 64)])
def test_linear_f16x3_shapes(cuda, M, N, K):
    """every tile configuration of the dispatch, ragged M and N, one chunk"""
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    out = ops.linear(a.to(cuda), sw(w.to(cuda)), b.to(cuda))
    e = rel_err(out, F.linear(a.double(), w.double(), b.double()))
    assert e <= F16_KERNEL_TOL, f"linear f16x3 {M}x{N}x{K}: {e:.3e}"


def test_linear_f16x3_matches_its_own_model_and_beats_the_bf16_halves(cuda):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(11)
    M, N, K = 512, 256, 256
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / 16
    out = ops.linear(a.to(cuda), sw(w.to(cuda)), None)
    out_b = ops.linear(a.to(cuda), ops.split_weights(w.to(cuda)), None)
    sp = lambda t: (t.half().double(), (t - t.half().float()).half().double())
    (ah, al), (wh, wl) = sp(a), sp(w)
    model = al @ wh.T + ah @ wl.T + ah @ wh.T
    exact = a.double() @ w.double().T
    assert rel_err(out, model) < 2e-6                                  # fp32 accumulation only
    assert rel_err(out, exact) < F16_KERNEL_TOL
    assert rel_err(out_b, exact) > 4 * rel_err(out, exact)             # bf16 halves: ~3e-6 here


def test_linear_f16x3_epilogues_strides_and_small_magnitudes(cuda):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 333, 256, 256
    big = torch.randn(M, 3 * K, generator=g)
    a = big[:, K:2 * K]
    w, b = torch.randn(N, K, generator=g) / 16, torch.randn(N, generator=g)
    r1, r2 = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    ad = big.to(cuda)[:, K:2 * K]
    wd, bd = sw(w.to(cuda)), b.to(cuda)
    lin = F.linear(a.double(), w.double(), b.double())
    assert rel_err(ops.linear(ad, wd, bd, act=ops.ACT_GELU), F.gelu(lin)) < F16_KERNEL_TOL
    assert rel_err(ops.linear(ad, wd, bd, r1=r1.to(cuda), r2=r2.to(cuda)), r1.double() * lin + r2.double()) < F16_KERNEL_TOL
    assert rel_err(ops.linear(ad, wd, bd, act=ops.ACT_RELU), F.relu(lin)) < F16_KERNEL_TOL
    w3 = torch.randn(6, K, generator=g) / 16
    assert rel_err(ops.linear(ad, sw(w3.to(cuda)), None), F.linear(a.double(), w3.double())) < F16_KERNEL_TOL
    # operands of small magnitude lose their low halves gradually below 2^-14 (subnormal halves are kept, not flushed): at 1e-2 x 3e-2 the
    # product is still ~1e-6 from float64; a flushed low half would be ~2e-3
    a2, w2 = a * 1e-2, w * 0.5
    e = rel_err(ops.linear((big * 1e-2).to(cuda)[:, K:2 * K], sw(w2.to(cuda)), None), F.linear(a2.double(), w2.double()))
    assert e < 1e-5, e


@pytest.mark.parametrize("B,H,W,Cin,Cout,up,relu", [(2, 8, 8, 256, 128, False, True), (1, 8, 8, 128, 128, True, True), (2, 5, 7, 64, 64, True, True),
                                                     (1, 16, 16, 64, 32, False, True), (3, 33, 17, 128, 64, False, True), (1, 64, 64, 32, 32, True, True)])
def test_conv3x3_reflect_f16x3(cuda, B, H, W, Cin, Cout, up, relu):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(H * 31 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    xi = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up else x.double()
    ref = F.conv2d(F.pad(xi, (1, 1, 1, 1), mode="reflect"), w.double(), b.double())
    ref = F.relu(ref) if relu else ref
    out = ops.conv3x3_reflect(x.permute(0, 2, 3, 1).contiguous().to(cuda), sw(w.permute(0, 2, 3, 1).contiguous().to(cuda)), b.to(cuda), up, relu)
    e = rel_err(out.permute(0, 3, 1, 2), ref)
    assert e <= F16_KERNEL_TOL, f"conv f16x3: {e:.3e}"


@pytest.mark.parametrize("M,C,ln,res", [(128, 256, False, True), (300, 256, True, True), (1000, 128, True, True), (4099, 256, True, False), (33, 128, True, True)])
def test_fused_mlp_f16x3(cuda, M, C, ln, res):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(M + C)
    HID = 4 * C
    x = torch.randn(M, C, generator=g) * 1.5 + 0.3
    w1, b1 = torch.randn(HID, C, generator=g) / C ** 0.5, torch.randn(HID, generator=g) * 0.1
    w2, b2 = torch.randn(C, HID, generator=g) / HID ** 0.5, torch.randn(C, generator=g) * 0.1
    lg, lb = torch.randn(C, generator=g) * 0.2 + 1.0, torch.randn(C, generator=g) * 0.1
    xd = x.to(cuda)
    out = ops.mlp(xd, sw(w1.to(cuda)), b1.to(cuda), sw(w2.to(cuda), layout=1), b2.to(cuda),
                  res=xd if res else None, ln_gamma=lg.to(cuda) if ln else None, ln_beta=lb.to(cuda) if ln else None)
    xin = F.layer_norm(x.double(), (C,), lg.double(), lb.double(), 1e-5) if ln else x.double()
    ref = F.linear(F.gelu(F.linear(xin, w1.double(), b1.double())), w2.double(), b2.double())
    if res:
        ref = ref + x.double()
    e = rel_err(out, ref)
    assert e <= 2 * F16_KERNEL_TOL, f"fused mlp f16x3 M={M} C={C}: {e:.3e}"          # + the fp32 LayerNorm and the 1.5e-7 erf


@pytest.mark.parametrize("dual", [False, True])
@pytest.mark.parametrize("H,W,shift", [(14, 14, (4, 4)), (9, 11, (4, 4)), (32, 32, (3, 3))])
def test_window_attention_f16x3_vs_exact_fp32(cuda, H, W, shift, dual):
    from mastermetastyletransfer_amd import ops
    from mastermetastyletransfer_amd.window_plan import window_plan
    heads, B = 8, 3
    C = heads * 32
    g = torch.Generator().manual_seed(H * 100 + W)
    T = H * W
    qkv = (torch.randn(B * T, 4 * C, generator=g) * 1.5).to(cuda)
    pads = (torch.randn(3, C, generator=g)).to(cuda)
    bias64 = torch.zeros(heads, 64, 64); bias64[:, :, 49:] = -1e30; bias64[:, :49, :49] = torch.randn(heads, 49, 49, generator=g) * 0.5
    bias64 = bias64.to(cuda)
    plan = window_plan(H, W, [7, 7], list(shift), cuda)
    args = (qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:3 * C], qkv[:, 3 * C:] if dual else None, pads[0], pads[1], pads[2] if dual else None,
            bias64, plan.src, plan.region, B, T, heads, plan.nW, plan.N, 32 ** -0.5)
    ref = ops.window_attn(*args)
    got = ops.window_attn(*args, x3="f16")
    got_b = ops.window_attn(*args, x3="bf16")
    for a, b, c in zip(got, ref, got_b):
        if b is not None:
            e = rel_err(a, b.double().cpu())
            assert e < 6e-6 and e < rel_err(c, b.double().cpu()), e        # both fp32-accumulated; the logits' error enters through exp


def test_conv_chain_in_split_half_layout(cuda):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(9)
    B, H, W = 2, 12, 20
    x = torch.randn(B, H, W, 64, generator=g).to(cuda)
    w1 = torch.randn(96, 3, 3, 64, generator=g) / 24; w2 = torch.randn(32, 3, 3, 96, generator=g) / 30
    b1, b2 = torch.randn(96, generator=g).to(cuda), torch.randn(32, generator=g).to(cuda)
    s1, s2 = sw(w1.to(cuda)), sw(w2.to(cuda))
    mid32 = ops.conv3x3_reflect(x, s1, b1, True, True)
    mid = ops.conv3x3_reflect(x, s1, b1, True, True, out_split=True)
    assert ops.is_split_act(mid) and mid.dtype == torch.float16 and mid.shape == (B, 2 * H, 2 * W, 3, 2, 32)
    hi = mid32.half(); lo = (mid32 - hi.float()).half()
    ref_mid = torch.stack([hi.view(B, 2 * H, 2 * W, 3, 32), lo.view(B, 2 * H, 2 * W, 3, 32)], dim=4)
    assert torch.equal(mid.view(torch.int16), ref_mid.view(torch.int16))
    out_a = ops.conv3x3_reflect(mid, s2, b2, False, True)
    out_b = ops.conv3x3_reflect(mid32, s2, b2, False, True)
    assert out_a.dtype == torch.float32 and rel_err(out_a, out_b.double().cpu()) < F16_KERNEL_TOL
    with pytest.raises(AssertionError):                                 # a bf16-split map is not an operand of half-split weights
        ops.conv3x3_reflect(mid.view(torch.bfloat16), s2, b2, False, True)


@pytest.mark.parametrize("B,Hs,Ws,split_in", [(2, 8, 16, False), (1, 12, 48, True), (1, 64, 64, False)])
def test_decoder_tail_f16x3(cuda, B, Hs, Ws, split_in):
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(Hs * 7 + Ws)
    x = torch.randn(B, 32, Hs, Ws, generator=g)
    w8, b8 = torch.randn(32, 32, 3, 3, generator=g) / 17, torch.randn(32, generator=g) * 0.1
    w9, b9 = torch.randn(3, 32, 3, 3, generator=g) / 17, torch.randn(3, generator=g) * 0.1
    up = F.interpolate(x.double(), scale_factor=2, mode="nearest")
    mid = F.relu(F.conv2d(F.pad(up, (1, 1, 1, 1), mode="reflect"), w8.double(), b8.double()))
    ref = F.conv2d(F.pad(mid, (1, 1, 1, 1), mode="reflect"), w9.double(), b9.double())
    xd = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    if split_in:
        hi = xd.half(); lo = (xd - hi.float()).half()
        xd = torch.stack([hi, lo], dim=3).view(B, Hs, Ws, 1, 2, 32).contiguous()
    out = ops.decoder_tail(xd, sw(w8.permute(0, 2, 3, 1).contiguous().to(cuda)), b8.to(cuda), sw(w9.permute(0, 2, 3, 1).contiguous().to(cuda)), b9.to(cuda))
    assert out.shape == (B, 3, 2 * Hs, 2 * Ws) and out.dtype == torch.float32
    e = rel_err(out, ref)
    assert e <= 2 * F16_KERNEL_TOL, f"decoder tail f16x3: {e:.3e}"


def full_model(dev, mode="f16x3", seed=31):
    from mastermetastyletransfer_amd import MasterStyleTransferModel
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True)
    sds = (seeded_state_dict(O.swin_shapes(), seed), seeded_state_dict(O.style_transformer_shapes(), seed + 1),
           seeded_state_dict(O.decoder_shapes(), seed + 2))
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    return m.to(dev).eval().set_compute_dtype(mode), sds


@pytest.mark.parametrize("B,S,k", [(1, 256, 1), (2, 256, 1), (1, 256, 3), (1, 512, 1), (1, 64, 1), (1, 200, 1)])
def test_full_forward_f16x3_vs_oracle(cuda, B, S, k):
    """BASELINE configs[0] and [4] shapes: the literal 1e-3 and the 3e-4 the half halves hold, ABSOLUTE, against the CPU oracle"""
    m, (swin_sd, st_sd, dec_sd) = full_model(cuda)
    c = seeded_input(f"c{S}", (B, 3, S, S), 8, kind="image")
    s = seeded_input(f"s{S}", (B, 3, S, S), 9, kind="image")
    with torch.no_grad():
        got = m(c.to(cuda), s.to(cuda), k).cpu()
        ref = O.full_forward(c, s, k, swin_sd, st_sd, dec_sd)
    assert got.shape == (B, 3, S, S) and got.dtype == torch.float32 and torch.isfinite(got).all()
    err, top = float((got - ref).abs().max()), float(ref.abs().max())
    print(f"f16x3 full forward B={B} S={S} k={k}: max|d| = {err:.2e}, ref max {top:.2f}")
    # with these seeded (untrained) weights k = 3 drives the output to ~5e3, where an absolute bar is meaningless: there 2e-5 of max|ref|
    assert err <= (F16_IMAGE_TOL if top <= 20.0 else 2e-5 * top), (err, top)


def test_parity_sweep_over_seeds_and_sizes_both_formats(cuda):
    """the worst case over pairs, not one seeded pair.  12 content / style pairs (64^2, 128^2, 256^2 x 4 seeds, incl. the
    smoke() pair): the half halves stay under 3e-4 on every pair (the exact-fp32 kernels: ~1.2e-4); the bf16 halves range over
    5e-4 ... 1.1e-3 on the same pairs, i.e. they meet the 1e-3 bar on the benchmark pair but not with margin everywhere - which is why
    "f16x3" is the parity mode the benchmark quotes and "bf16x3" the range-safe fallback.  The bound for the bf16 halves here is 2e-3."""
    mh, sds = full_model(cuda, "f16x3", seed=1)
    worst = {"f16x3": 0.0, "bf16x3": 0.0, "fp32": 0.0}
    for size in (64, 128, 256):
        for seed in (4, 14, 24, 34):
            c = seeded_input("smoke_c" if seed == 4 else f"sim_c{seed}", (1, 3, size, size), seed, kind="image")
            s = seeded_input("smoke_s" if seed == 4 else f"sim_s{seed}", (1, 3, size, size), seed + 1, kind="image")
            with torch.no_grad():
                ref = O.full_forward(c, s, 1, *sds)
                for mode in ("f16x3", "bf16x3", "fp32"):
                    mh.set_compute_dtype(torch.float32 if mode == "fp32" else mode)
                    e = float((mh(c.to(cuda), s.to(cuda), 1).cpu() - ref).abs().max())
                    worst[mode] = max(worst[mode], e)
    print("parity sweep, worst max|d| over 12 pairs:", {k: f"{v:.2e}" for k, v in worst.items()})
    assert worst["fp32"] <= F16_IMAGE_TOL and worst["f16x3"] <= F16_IMAGE_TOL and worst["bf16x3"] <= 2e-3, worst
    assert worst["f16x3"] < 0.5 * worst["bf16x3"]


def test_full_batch_32_f16x3_properties(cuda):
    """bs = 32 at 256 x 256: bitwise determinism, per-sample batch independence, samples against the CPU oracle"""
    m, (swin_sd, st_sd, dec_sd) = full_model(cuda)
    c = seeded_input("c32", (32, 3, 256, 256), 10, kind="image").to(cuda)
    s = seeded_input("s32", (32, 3, 256, 256), 11, kind="image").to(cuda)
    with torch.no_grad():
        full = m(c, s, 1)
        assert torch.equal(full, m(c, s, 1)), "forward is not deterministic"
        for i in (0, 13, 31):
            assert torch.equal(m(c[i:i + 1], s[i:i + 1], 1), full[i:i + 1]), f"sample {i} depends on its batch"
        for i in (5, 31):
            ref = O.full_forward(c[i:i + 1].cpu(), s[i:i + 1].cpu(), 1, swin_sd, st_sd, dec_sd)
            err = float((full[i:i + 1].cpu() - ref).abs().max())
            assert err <= F16_IMAGE_TOL, (i, err)


def test_f16x3_mode_switches_graph_replay_and_refuses_training(cuda):
    from mastermetastyletransfer_amd import GraphedStylizer
    m, _ = full_model(cuda)
    c = seeded_input("cg", (2, 3, 256, 256), 8, kind="image").to(cuda)
    s = seeded_input("sg", (2, 3, 256, 256), 9, kind="image").to(cuda)
    with torch.no_grad():
        yh = m(c, s, 1)
        yb = m.set_compute_dtype("bf16x3")(c, s, 1)
        y32 = m.set_compute_dtype(torch.float32)(c, s, 1)
        assert not m.split_weights and float((yh - y32).abs().max()) <= F16_IMAGE_TOL and not torch.equal(yh, yb)
        assert torch.equal(m.set_compute_dtype(torch.float32, split="f16")(c, s, 1), yh)      # the derived-weight cache follows the mode
        assert m.split_weights == "f16"
        g = GraphedStylizer(m, 2, 256, 256, 1)
        g.content.copy_(c); g.style.copy_(s)
        assert torch.equal(g.replay(0), yh)
    with pytest.raises(ValueError):
        m.set_compute_dtype(torch.bfloat16, split="f16")
    with pytest.raises(ValueError):
        m.set_compute_dtype(torch.float32, split="fp8")
    m.set_compute_dtype("f16x3").train()
    with pytest.raises(NotImplementedError, match="exact-fp32"):
        m(c, s, 1)


def test_f16x3_kernels_repeatable_at_scale(cuda):
    """every split kernel with half halves at the sizes of the bs = 32 forward, several launches on the same inputs: bitwise identical"""
    import os
    import runpy
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = sys.argv
    sys.argv = ["x3_determinism.py", "--fmt", "f16"]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(os.path.join(root, "profiles", "tools", "x3_determinism.py"), run_name="__main__")
    finally:
        sys.argv = argv
    assert e.value.code == 0


@pytest.mark.parametrize("B,H,W,dc,ds", [(2, 64, 64, "euclidian", "euclidian"), (2, 128, 128, "euclidian_squared", "euclidian_squared"),
                                         (1, 256, 256, "euclidian", "euclidian"), (2, 88, 136, "euclidian_squared", "euclidian_squared")])
def test_perceptual_loss_backward_f16x3_network(cuda, B, H, W, dc, ds):
    """custom_loss.set_compute_dtype("f16x3"): IEEE-half halves in the frozen VGG-19's FORWARD convolutions (the pass whose rounding this
    loss amplifies ~1e4 times), bf16 halves in its data-gradient chain (gradients span fp32's exponent range).  The per-pixel image
    gradient, which the all-bf16-halves network leaves 0.9 - 1.7e-2 (relative L2) from float64, comes back to the exact fp32 network's
    distance; loss values to 2e-4."""
    from mastermetastyletransfer_amd import custom_loss
    from oracle import loss_oracle as L
    sd = seeded_state_dict(L.vgg19_shapes(), 51)
    c, s = (seeded_input(f"plb_{n}", (B, 3, H, W), 52 + i, kind="image") for i, n in enumerate("cs"))
    o = seeded_input("plb_o", (B, 3, H, W), 55, kind="image")
    o64 = o.double().requires_grad_(True)
    total, cl, sl = L.overall_loss(c.double(), s.double(), o64, {k_: v.double() for k_, v in sd.items()}, 7.0, dc, ds)
    total.backward()
    ref = o64.grad
    loss = custom_loss(default_lambda_value=7.0, distance_content=dc, distance_style=ds, vgg_state_dict=sd).to(cuda)
    l2 = {}
    for mode in ("f16x3", "bf16x3", torch.float32):
        loss.set_compute_dtype(mode)
        od = o.to(cuda).requires_grad_(True)
        t2, c2, s2 = loss(c.to(cuda), s.to(cuda), od, output_content_and_style_loss=True)
        assert abs(float(t2.detach()) - float(total)) <= 2e-4 * abs(float(total)) and abs(float(c2.detach()) - float(cl)) <= 2e-4 * abs(float(cl))
        t2.backward()
        assert torch.isfinite(od.grad).all()
        l2[str(mode)] = float((od.grad.cpu().double() - ref).norm() / ref.norm())
    print(f"loss network {B}x{H}x{W} {dc}: image gradient relative L2 from float64:", {k_: f"{v:.2e}" for k_, v in l2.items()})
    assert l2["f16x3"] <= max(2.5 * l2["torch.float32"], 2e-3), l2          # the exact network's level (its own distance is 0.5 - 3.5e-3)
    assert l2["f16x3"] < 0.5 * l2["bf16x3"], l2


def test_loss_network_modes_and_the_data_gradient_format(cuda):
    """the data gradient never takes half halves: the C ABI rejects MASTER_F16X3 there, the module hands it bf16-split flipped weights"""
    from mastermetastyletransfer_amd import custom_loss, ops
    from mastermetastyletransfer_amd.ops import _p, _stream, load_library
    from oracle import loss_oracle as L
    loss = custom_loss(default_lambda_value=10.0, vgg_state_dict=seeded_state_dict(L.vgg19_shapes(), 6)).to(cuda).set_compute_dtype("f16x3")
    vgg = loss.feature_extractor_model
    assert vgg.split_weights == "f16" and vgg._weights(2)[0].dtype == torch.float16 and vgg._flipped(2).dtype == torch.bfloat16
    with pytest.raises(ValueError):
        loss.set_compute_dtype("fp8x3")
    with pytest.raises(ValueError):
        loss.set_compute_dtype(torch.bfloat16, split="f16")
    g = torch.randn(1, 8, 8, 64, device=cuda); out = torch.empty(1, 8, 8, 64, device=cuda)
    wf = ops.split_weights(torch.randn(64, 3, 3, 64, device=cuda) / 24, fmt="f16")
    rc = load_library().master_conv3x3_dgrad(_p(g), _p(wf), _p(out), None, None, 1, 8, 8, 64, 64, 0, ops.F16X3, _stream())
    assert rc != 0


def test_train_step_with_f16x3_loss_network_and_encoder(cuda):
    """whole train step with the two FROZEN networks in the split mode - the VGG-19 of the loss (half halves forward, bf16 halves in the
    data gradients) and the Swin encoder (half halves; no graph) - and the trainable StyleTransformer / decoder, their backward and Adam in
    exact fp32, against the float64 oracle: every parameter gradient within the bound the all-fp32 step is held to (5e-3 of the tensor's
    max; tests/test_gpu_train.py::test_full_train_step), loss to 2e-4"""
    from mastermetastyletransfer_amd import MasterStyleTransferModel, custom_loss
    from mastermetastyletransfer_amd import train_forward as TF
    from oracle import loss_oracle as L, train_oracle as T
    B, S, k = 4, 128, 1
    sds = (seeded_state_dict(O.swin_shapes(), 1), seeded_state_dict(O.style_transformer_shapes(), 2), seeded_state_dict(O.decoder_shapes(), 3))
    vgg = seeded_state_dict(L.vgg19_shapes(), 4)
    c = seeded_input("bfl_c", (B, 3, S, S), 5, kind="image"); s = seeded_input("bfl_s", (B, 3, S, S), 6, kind="image")
    g = torch.Generator().manual_seed(7)
    probs = [p for p in O.SWIN_B_SD_PROBS if p > 0]
    swin_masks = [torch.empty(B).bernoulli_(1 - p, generator=g).div_(1 - p) for _ in range(2) for p in probs for _ in range(2)]
    masks = O.SDMasks.draw(O.SD_CALLS_PER_LAYER * k, B, 0.1, g)
    losses64, ref64 = T.train_step_grads(c.double(), s.double(), k, *sds, vgg, masks, 10.0, swin_masks=swin_masks)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(cuda).train().set_encoder_compute_dtype("f16x3")
    for p in m.swin_encoder.parameters():
        p.requires_grad = False
    loss_fn = custom_loss(default_lambda_value=10.0, vgg_state_dict=vgg).to(cuda).set_compute_dtype("f16x3")
    named = [("style_transformer." + n, p) for n, p in m.style_transformer.named_parameters()] + [("decoder." + n, p) for n, p in m.decoder.named_parameters()]
    gscale = max(float(v.abs().max()) for v in ref64.values())
    worst = {}
    for enc in ("f16x3", torch.float32):
        m.set_encoder_compute_dtype(enc)
        for _, p in named:
            p.grad = None
        TF.sd_masks.set_masks(swin_masks + masks)
        try:
            out = m(c.to(cuda), s.to(cuda), k)
        finally:
            TF.sd_masks.set_masks(None)
        total = loss_fn(c.to(cuda), s.to(cuda), out)
        total.backward()
        assert abs(float(total.detach()) - float(losses64["total"])) <= 2e-4 * abs(float(losses64["total"]))
        errs = sorted((float((p.grad.cpu().double() - ref64[n]).abs().max()) / float(ref64[n].abs().max()), n) for n, p in named
                      if n in ref64 and p.grad is not None and float(ref64[n].abs().max()) > 1e-6 * gscale)
        worst[str(enc)] = errs[-1]
        print(f"f16x3 loss network, encoder {enc}: worst parameter gradient {errs[-1][0]:.2e} of its tensor's max ({errs[-1][1]}), median {errs[len(errs) // 2][0]:.2e}")
        assert errs[-1][0] <= 5e-3, errs[-3:]
    with pytest.raises(ValueError):
        m.set_encoder_compute_dtype(torch.bfloat16)


def test_encoder_only_switch_in_eval_mode(cuda):
    """set_encoder_compute_dtype() next to any model mode: the encoder's fp32 tokens are converted where the rest of the model is bf16;
    a later set_compute_dtype() overrides the encoder switch"""
    m, (swin_sd, st_sd, dec_sd) = full_model(cuda, torch.float32)
    c = seeded_input("c64", (1, 3, 64, 64), 8, kind="image").to(cuda)
    s = seeded_input("s64", (1, 3, 64, 64), 9, kind="image").to(cuda)
    with torch.no_grad():
        y32 = m(c, s, 1)
        yenc = m.set_encoder_compute_dtype("f16x3")(c, s, 1)                      # fp32 model, split encoder
        assert float((yenc - y32).abs().max()) <= F16_IMAGE_TOL and not torch.equal(yenc, y32)
        ybf = m.set_compute_dtype(torch.bfloat16)(c, s, 1).float()
        ymix = m.set_encoder_compute_dtype("f16x3")(c, s, 1).float()              # bf16 model, split encoder
        assert torch.isfinite(ymix).all() and float((ymix - ybf).norm() / ybf.norm()) < 3e-2
        assert torch.equal(m.set_compute_dtype(torch.float32)(c, s, 1), y32)       # back: the encoder follows the model again


def test_overflow_inside_the_model_reaches_the_image(cuda):
    """whole-model form of the trade-off below: a forward whose activations leave half's range (seeded random weights at k = 5: the exact
    modes produce values up to 4.5e7) must come out of the f16x3 mode with inf / NaN in the image - the decoder's ReLUs propagate NaN like
    torch.relu (r05: they used v_max_f32, which answers 0 for a NaN, and this forward came out finite and wrong).  The exact-fp32 kernels and
    the bf16 halves, which keep fp32's exponent, stay finite on the same input."""
    import warnings
    from mastermetastyletransfer_amd import MasterStyleTransferModel
    sds = (seeded_state_dict(O.swin_shapes(), 31), seeded_state_dict(O.style_transformer_shapes(), 32), seeded_state_dict(O.decoder_shapes(), 33))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(cuda).eval()
    c = seeded_input("c", (2, 3, 96, 64), 8, kind="image").to(cuda); s = seeded_input("s", (2, 3, 96, 64), 9, kind="image").to(cuda)
    with torch.no_grad():
        y32 = m.set_compute_dtype(torch.float32)(c, s, 5)
        assert torch.isfinite(y32).all() and float(y32.abs().max()) > 1e6
        yh = m.set_compute_dtype("f16x3")(c, s, 5)
        assert not torch.isfinite(yh).all(), "an overflow of half's range inside the model must not produce a finite image"
        yb = m.set_compute_dtype("bf16x3")(c, s, 5)
        assert torch.isfinite(yb).all() and float((yb - y32).abs().max()) <= 2e-3 * float(y32.abs().max())


def test_half_halves_overflow_is_loud_and_bf16_halves_cover_the_range(cuda):
    """the documented trade-off: an operand beyond half's range (|x| > 65 504) makes the f16x3 product inf / NaN - never a silently wrong
    finite number - while the bf16 halves, which keep fp32's exponent, stay accurate; tiny operands (1e-6) underflow the half split's low
    half gracefully (error bounded by the subnormal floor), the bf16 split keeps its relative accuracy"""
    from mastermetastyletransfer_amd import ops
    g = torch.Generator().manual_seed(21)
    M, N, K = 256, 128, 128
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / 11
    big = a * 1e5                                                        # |x| up to ~4e5
    ref = F.linear(big.double(), w.double())
    out_h = ops.linear(big.to(cuda), sw(w.to(cuda)), None).cpu()
    out_b = ops.linear(big.to(cuda), ops.split_weights(w.to(cuda)), None)
    assert not torch.isfinite(out_h).all()
    assert rel_err(out_b, ref) < 3e-5
    tiny = a * 1e-6
    ref_t = F.linear(tiny.double(), w.double())
    e_h = rel_err(ops.linear(tiny.to(cuda), sw(w.to(cuda)), None), ref_t)
    e_b = rel_err(ops.linear(tiny.to(cuda), ops.split_weights(w.to(cuda)), None), ref_t)
    assert e_b < 3e-5 and e_h < 2e-2 and e_h > e_b                       # half: hi alone (11 bits, partly subnormal) is what is left of a 1e-6 operand
    from mastermetastyletransfer_amd.ops import MasterHipError
    with pytest.raises(MasterHipError, match="outside IEEE half"):       # static operands are range-checked where they are split
        sw((w * 1e7).to(cuda))
