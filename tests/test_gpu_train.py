"""GPU: the train step (reference train_only_inner_loop.py:519-575) on the HIP forward + backward kernels against torch autograd on the
CPU oracle: every backward kernel alone, then the reference-pinned golden train cases (StyleTransformer + Decoder + loss), then the whole
model step (frozen Swin encoder with its StochasticDepth, random k, Adam).  fp32 mode; bar: <= 1e-3 of each tensor's max-abs."""
import os
import random
import sys
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import loss_oracle as L
from oracle import master_oracle as O
from oracle import train_oracle as T
from oracle.seeded import seeded_input, seeded_state_dict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden_train as M                                    # noqa: E402  (case definitions only; the reference is not imported)

pytestmark = pytest.mark.gpu
TOL = 1e-3


def rel(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-20)


def check(name, got, ref, tol=TOL, floor=0.0):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = float((got - ref).abs().max())
    assert err <= tol * max(float(ref.abs().max()), floor), f"{name}: max|d| {err:.3e} vs max|ref| {float(ref.abs().max()):.3e}"


# ---------------------------------------------------------------------------------------------------------------------------------------
# single kernels
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(8192, 256, 256), (1000, 768, 256), (4100, 1024, 256), (333, 256, 1024), (64, 32, 64), (50, 512, 128),
                                   (1000, 576, 192), (300, 192, 768), (130, 288, 96)])      # the last three: swin_T / swin_S widths
def test_linear_backward(cuda, M, N, K):
    from mastermetastyletransfer_amd.autograd import LinearFn
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g, requires_grad=True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, generator=g).requires_grad_(True)
    go = torch.randn(M, N, generator=g)
    F.linear(a, w, b).backward(go)
    ad, wd, bd = (t.detach().to(cuda).requires_grad_(True) for t in (a, w, b))
    out = LinearFn.apply(ad, wd, bd)
    check("out", out, F.linear(a, w, b), 1e-4)
    out.backward(go.to(cuda))
    check("da", ad.grad, a.grad); check("dw", wd.grad, w.grad); check("db", bd.grad, b.grad)


@pytest.mark.parametrize("M,C,HID,ln", [(2048, 256, 1024, True), (700, 256, 1024, False), (130, 128, 512, True), (500, 192, 768, True), (561, 256, 1024, True)])
def test_mlp_backward(cuda, M, C, HID, ln):
    from mastermetastyletransfer_amd.autograd import MLPFn
    g = torch.Generator().manual_seed(M)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).requires_grad_(True)
    x, w1, b1, w2, b2 = mk(M, C), mk(HID, C, sc=C ** -0.5), mk(HID, sc=0.3), mk(C, HID, sc=HID ** -0.5), mk(C, sc=0.3)
    gw, gb = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True), mk(C, sc=0.2)
    go = torch.randn(M, C, generator=g)
    xin = F.layer_norm(x, (C,), gw, gb, 1e-5) if ln else x
    ref = F.linear(F.gelu(F.linear(xin, w1, b1)), w2, b2)
    ref.backward(go)
    leaves = [x, w1, b1, w2, b2, gw, gb]
    dl = [t.detach().to(cuda).requires_grad_(True) for t in leaves]
    out = MLPFn.apply(dl[0], dl[1], dl[2], dl[3], dl[4], dl[5] if ln else None, dl[6] if ln else None, 1e-5)
    check("out", out, ref, 1e-4)
    out.backward(go.to(cuda))
    for n, d, r in zip("x w1 b1 w2 b2 gamma beta".split(), dl, leaves):
        if r.grad is not None:
            check(n, d.grad, r.grad)


@pytest.mark.parametrize("M,C", [(1024, 256), (77, 128), (300, 512), (64, 1024), (200, 192), (100, 96), (65, 384)])
def test_layernorm_backward(cuda, M, C):
    from mastermetastyletransfer_amd.autograd import LayerNormFn
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).requires_grad_(True)
    gw = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    gb = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    go = torch.randn(M, C, generator=g)
    F.layer_norm(x, (C,), gw, gb, 1e-5).backward(go)
    xd, wd, bd = (t.detach().to(cuda).requires_grad_(True) for t in (x, gw, gb))
    LayerNormFn.apply(xd, wd, bd, 1e-5).backward(go.to(cuda))
    check("dx", xd.grad, x.grad); check("dgamma", wd.grad, gw.grad); check("dbeta", bd.grad, gb.grad)


@pytest.mark.parametrize("M,C,HID,ln", [(2048, 256, 1024, True), (700, 256, 1024, False), (130, 128, 512, False)])
def test_forked_skip_connections_backward(cuda, M, C, HID, ln):
    """x + 0.7 * mlp(norm?(x)) and x + LayerNorm(x) * 2 with the skip connection taken from the alias the Function hands back: x has a
    single consumer and the skip gradient is added inside the backward kernels (LayerNorm backward / fc1 data-gradient epilogue)"""
    from mastermetastyletransfer_amd.autograd import LayerNormFn, MLPFn
    g = torch.Generator().manual_seed(M + 1)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).requires_grad_(True)
    x, w1, b1, w2, b2 = mk(M, C), mk(HID, C, sc=C ** -0.5), mk(HID, sc=0.3), mk(C, HID, sc=HID ** -0.5), mk(C, sc=0.3)
    gw, gb = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True), mk(C, sc=0.2)
    go = torch.randn(M, C, generator=g)
    xin = F.layer_norm(x, (C,), gw, gb, 1e-5) if ln else x
    ref = x + 0.7 * F.linear(F.gelu(F.linear(xin, w1, b1)), w2, b2)
    ref = ref + 2.0 * F.layer_norm(ref, (C,), gw, gb, 1e-5)
    ref.backward(go)
    leaves = [x, w1, b1, w2, b2, gw, gb]
    dl = [t.detach().to(cuda).requires_grad_(True) for t in leaves]
    y, xs = MLPFn.apply(dl[0], dl[1], dl[2], dl[3], dl[4], dl[5] if ln else None, dl[6] if ln else None, 1e-5, True)
    assert xs.data_ptr() == dl[0].data_ptr()
    h = xs + 0.7 * y
    n, hs = LayerNormFn.apply(h, dl[5], dl[6], 1e-5, True)
    out = hs + 2.0 * n
    check("out", out, ref, 1e-4)
    out.backward(go.to(cuda))
    for nm, d, r in zip("x w1 b1 w2 b2 gamma beta".split(), dl, leaves):
        check(nm, d.grad, r.grad)


@pytest.mark.parametrize("B,T,C,HID", [(4, 1024, 256, 1024), (3, 100, 128, 512), (2, 49, 256, 1024)])
def test_residual_blocks_with_row_mask_backward(cuda, B, T, C, HID):
    """res + mask[image] * (a W^T + b)  and  x + mask[image] * mlp(x): the StochasticDepth row mask and the residual add inside the forward
    GEMM's epilogue, the mask inside the data- and weight-gradient GEMMs (style_transformer.py:390-397); T = 100 / 49 make tiles straddle
    images.  Last: a bias that trains while its weight is frozen."""
    from mastermetastyletransfer_amd.autograd import LinearFn, MLPFn
    g = torch.Generator().manual_seed(B * T + C)
    M = B * T
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).requires_grad_(True)
    a, res, w, b = mk(M, C), mk(M, C), mk(C, C, sc=C ** -0.5), mk(C, sc=0.3)
    w1, b1, w2, b2 = mk(HID, C, sc=C ** -0.5), mk(HID, sc=0.3), mk(C, HID, sc=HID ** -0.5), mk(C, sc=0.3)
    mask = torch.tensor([0.0 if i == 1 else 1.0 / 0.9 for i in range(B)])
    rows = mask.repeat_interleave(T)[:, None]
    go = torch.randn(M, C, generator=g)
    y = res + rows * F.linear(a, w, b)
    ref = y + rows * F.linear(F.gelu(F.linear(y, w1, b1)), w2, b2)
    ref.backward(go)
    leaves = [a, res, w, b, w1, b1, w2, b2]
    dl = [t.detach().to(cuda).requires_grad_(True) for t in leaves]
    md = mask.to(cuda)
    yd = LinearFn.apply(dl[0], dl[2], dl[3], dl[1], md, T)
    out = MLPFn.apply(yd, dl[4], dl[5], dl[6], dl[7], None, None, 1e-5, False, md, T)
    check("out", out, ref, 1e-4)
    out.backward(go.to(cuda))
    for nm, d, r_ in zip("a res w b w1 b1 w2 b2".split(), dl, leaves):
        check(nm, d.grad, r_.grad)
    # frozen weight, trainable bias
    wf = w.detach().to(cuda)
    bd = b.detach().to(cuda).requires_grad_(True)
    ad = a.detach().to(cuda).requires_grad_(True)
    LinearFn.apply(ad, wf, bd, res.detach().to(cuda), md, T).backward(go.to(cuda))
    b.grad = None; a.grad = None
    (res.detach() + rows * F.linear(a, w.detach(), b)).backward(go)
    check("db alone", bd.grad, b.grad); check("da", ad.grad, a.grad)


def test_bias_tile_and_conv_weight_pack(cuda):
    """the relative-position bias tile (style_transformer.py:21-28) and its table gradient; the three layouts of a trainable conv weight"""
    from mastermetastyletransfer_amd.autograd import BiasTileFn, conv_weight_pack
    g = torch.Generator().manual_seed(3)
    for ws, heads in ((7, 8), (8, 4), (4, 3)):
        N = ws * ws
        table = torch.randn((2 * ws - 1) ** 2, heads, generator=g).requires_grad_(True)
        coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0) + (ws - 1)
        index = (rel[..., 0] * (2 * ws - 1) + rel[..., 1]).reshape(-1)
        ref = torch.zeros(heads, 64, 64)
        ref[:, :N, :N] = table[index].view(N, N, heads).permute(2, 0, 1)
        ref[:, :, N:] = -1e30
        go = torch.randn(heads, 64, 64, generator=g)
        (ref * go).sum().backward()
        td = table.detach().to(cuda).requires_grad_(True)
        tile = BiasTileFn.apply(td, index.to(cuda), N)
        assert torch.equal(tile.cpu(), ref.detach())
        (tile * go.to(cuda)).sum().backward()
        check("dtable", td.grad, table.grad, 1e-5)
    for Cout, Cin in ((128, 256), (3, 32), (64, 64)):
        w = torch.randn(Cout, Cin, 3, 3, generator=g)
        wp, wf = conv_weight_pack(w.to(cuda), True)
        Cp = (Cout + 31) // 32 * 32
        assert torch.equal(wp.cpu(), w.permute(0, 2, 3, 1).contiguous())
        ref_f = torch.zeros(Cin, 3, 3, Cp)
        ref_f[..., :Cout] = w.flip(2, 3).permute(1, 2, 3, 0)
        assert torch.equal(wf.cpu(), ref_f)


def test_reduce_segments(cuda):
    """up to four slab reductions with their own split counts in one launch; accumulate adds to the output"""
    from mastermetastyletransfer_amd.autograd import reduce_segments
    g = torch.Generator().manual_seed(5)
    shapes = [(37, 8 * 4096), (12, 256), (600, 256), (1, 100)]
    parts = [torch.randn(k, n, generator=g) for k, n in shapes]
    outs0 = [torch.randn(n, generator=g) for _, n in shapes]
    accs = [False, True, False, True]
    dp = [p_.to(cuda) for p_ in parts]
    do = [o.clone().to(cuda) for o in outs0]
    reduce_segments([(p_, o, p_.shape[1], p_.shape[1], p_.shape[0], a) for p_, o, a in zip(dp, do, accs)])
    for p_, o0, o, a in zip(parts, outs0, do, accs):
        ref = p_.double().sum(0) + (o0.double() if a else 0)
        check("segment", o, ref.float(), 1e-5)


@pytest.mark.parametrize("B,T,C,n_pad", [(2, 1024, 256, 201), (3, 64, 256, 132), (2, 100, 64, 0), (1, 4096, 128, 0), (2, 100, 192, 30), (3, 187, 256, 107)])
def test_instnorm_backward(cuda, B, T, C, n_pad):
    """InstanceNorm over T rows + n_pad copies of a parameter row; gradients w.r.t. x and the row (style_transformer.py:520-530)"""
    from mastermetastyletransfer_amd.autograd import InstNormFn
    g = torch.Generator().manual_seed(T)
    x = (torch.randn(B * T, C, generator=g) * 1.5 + 0.3).requires_grad_(True)
    row = (0.5 * torch.randn(C, generator=g)).requires_grad_(True)
    go = torch.randn(B * T, C, generator=g)
    gpad = torch.randn(B, C, generator=g)
    ext = torch.cat([x.view(B, T, C), row.view(1, 1, C).expand(B, n_pad, C)], dim=1) if n_pad else x.view(B, T, C)
    y = F.instance_norm(ext.permute(0, 2, 1), eps=1e-5).permute(0, 2, 1)
    loss = (y[:, :T].reshape(B * T, C) * go).sum()
    if n_pad:
        loss = loss + (y[:, T] * gpad).sum() * 1.0       # every pad copy is the same row; upstream gradient = the sum over its copies
    loss.backward()
    xd, rd = x.detach().to(cuda).requires_grad_(True), row.detach().to(cuda).requires_grad_(True)
    yd, pad_out = InstNormFn.apply(xd, rd if n_pad else None, B, T, n_pad, 1e-5)
    check("y", yd, y[:, :T].reshape(B * T, C), 1e-4)
    l2 = (yd * go.to(cuda)).sum()
    if n_pad:
        check("pad_out", pad_out, y[:, T], 1e-4)
        l2 = l2 + (pad_out * gpad.to(cuda)).sum()
    l2.backward()
    check("dx", xd.grad, x.grad)
    if n_pad:
        check("drow", rd.grad, row.grad)


def _attn_module(cuda, dim, nh, shift, seed):
    from mastermetastyletransfer_amd import ShiftedWindowAttention
    m = ShiftedWindowAttention(dim, nh, [7, 7], list(shift))
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed)
    m.load_state_dict(sd)
    return m.to(cuda), sd


@pytest.mark.parametrize("H,W,shift", [(7, 7, (0, 0)), (14, 14, (3, 3)), (9, 9, (4, 4)), (9, 12, (4, 4)), (14, 9, (4, 0)), (32, 32, (4, 4)), (11, 17, (4, 4))])
@pytest.mark.parametrize("mode", ["qkv", "qk_v", "q_k_v", "dual"])
def test_window_attention_backward(cuda, H, W, shift, mode):
    """train_forward.attention (projections + master_window_attn_bwd + proj) vs autograd through the oracle's attention: gradients of
    the inputs, of Wq/Wk/Wv/proj (incl. the bias rows of zero-padded slots) and of the relative-position table"""
    from mastermetastyletransfer_amd import train_forward as TF
    dim, nh, B = 64, 2, 2
    m, sd = _attn_module(cuda, dim, nh, shift, 31)
    xs = [seeded_input(f"wab_{mode}_{n}", (B, H, W, dim), 31).requires_grad_(True) for n in "abcd"]
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    if mode == "qkv":
        ins = (xs[0], xs[0], [xs[0]])
    elif mode == "qk_v":
        ins = (xs[0], xs[0], [xs[1]])
    elif mode == "q_k_v":
        ins = (xs[0], xs[1], [xs[2]])
    else:
        ins = (xs[0], xs[0], [xs[1], xs[2]])
    refs = [O.shifted_window_attention(ins[0], ins[1], v, p, "", (7, 7), shift, nh) for v in ins[2]]
    gos = [seeded_input(f"wab_go{i}", (B, H, W, dim), 32) for i in range(len(refs))]
    sum((r * g).sum() for r, g in zip(refs, gos)).backward()
    tok = {}

    def dev(t):
        if id(t) not in tok:
            tok[id(t)] = t.detach().reshape(B * H * W, dim).to(cuda).requires_grad_(True)
        return tok[id(t)]
    outs = TF.attention(m, dev(ins[0]), dev(ins[1]), [dev(v) for v in ins[2]], B, H, W)
    for o, r in zip(outs, refs):
        check("out", o, r.reshape(B * H * W, dim), 1e-4)
    sum((o * g.reshape(B * H * W, dim).to(cuda)).sum() for o, g in zip(outs, gos)).backward()
    for t in xs:
        if t.grad is not None:
            check("dx", tok[id(t)].grad, t.grad.reshape(B * H * W, dim))
    gscale = max(float(v.grad.abs().max()) for k, v in p.items() if v.is_floating_point() and v.grad is not None)
    for n, prm in m.named_parameters():
        check(n, prm.grad, p[n].grad, floor=1e-3 * gscale)        # Wk.bias: zero in exact arithmetic (softmax shift invariance)


@pytest.mark.parametrize("B,H,W", [(2, 9, 9), (2, 32, 32), (1, 14, 12), (3, 11, 17)])
def test_sigma_mu_attention_backward(cuda, B, H, W):
    from mastermetastyletransfer_amd import StyleTransformer
    from mastermetastyletransfer_amd import train_forward as TF
    dim, nh = 64, 2
    st = StyleTransformer(dim, dim, nh, nh, [7, 7], [7, 7], [4, 4], [4, 4])
    sd = seeded_state_dict(O.style_transformer_shapes(dim, nh), 41)
    st.load_state_dict(sd)
    att = st.decoder.decoder_MHA_for_sigma_and_mu.to(cuda)
    pfx = "decoder.decoder_MHA_for_sigma_and_mu."
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    xs = [seeded_input(f"smb_{n}", (B, H, W, dim), 41).requires_grad_(True) for n in "abcd"]
    rs, rm = O.sigma_mu_attention(*xs, p, pfx, (7, 7), (4, 4), nh)
    gs, gm = seeded_input("smb_gs", (B, H, W, dim), 42), seeded_input("smb_gm", (B, H, W, dim), 43)
    ((rs * gs).sum() + (rm * gm).sum()).backward()
    d = [t.detach().reshape(B * H * W, dim).to(cuda).requires_grad_(True) for t in xs]
    s, mu = TF.sigma_mu(att, *d, B, H, W)
    check("sigma", s, rs.reshape(-1, dim), 1e-4); check("mu", mu, rm.reshape(-1, dim), 1e-4)
    ((s * gs.reshape(-1, dim).to(cuda)).sum() + (mu * gm.reshape(-1, dim).to(cuda)).sum()).backward()
    for t, dt in zip(xs, d):
        check("dx", dt.grad, t.grad.reshape(-1, dim))
    gscale = max(float(p[pfx + n].grad.abs().max()) for n, _ in att.named_parameters())
    for n, prm in att.named_parameters():
        check(n, prm.grad, p[pfx + n].grad, floor=1e-3 * gscale)   # Wk.bias: zero in exact arithmetic (InstanceNorm removes it)


@pytest.mark.parametrize("B,H,W,Cin,Cout,up,relu,nchw", [(2, 8, 8, 256, 128, False, True, False), (1, 16, 16, 128, 128, True, True, False),
                                                         (2, 12, 20, 64, 32, True, True, False), (2, 16, 16, 32, 3, False, False, True),
                                                         (1, 2, 2, 32, 32, True, True, False), (1, 6, 10, 128, 64, False, True, False),
                                                         (1, 8, 8, 192, 96, False, True, False), (2, 6, 6, 96, 64, True, True, False),
                                                         (3, 11, 17, 256, 128, False, True, False), (3, 11, 17, 128, 128, True, True, False),
                                                         (1, 5, 7, 64, 64, True, True, False), (2, 9, 13, 128, 64, False, True, False)])
def test_conv_reflect_backward(cuda, B, H, W, Cin, Cout, up, relu, nchw):
    """Upsample? -> ReflectionPad -> Conv3x3 -> ReLU? (decoder.py:24-54): data, weight and bias gradients"""
    from mastermetastyletransfer_amd.autograd import ConvReflectFn
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).requires_grad_(True)
    b = (0.1 * torch.randn(Cout, generator=g)).requires_grad_(True)
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(F.pad(xi, (1, 1, 1, 1), mode="reflect"), w, b)
    if relu:
        ref = F.relu(ref)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(cuda).requires_grad_(True)
    wd, bd = w.detach().to(cuda).requires_grad_(True), b.detach().to(cuda).requires_grad_(True)
    out = ConvReflectFn.apply(xd, wd, bd, up, relu, nchw)
    check("out", out if nchw else out.permute(0, 3, 1, 2), ref, 1e-4)
    out.backward((go if nchw else go.permute(0, 2, 3, 1).contiguous()).to(cuda))
    check("dx", xd.grad.permute(0, 3, 1, 2), x.grad); check("dw", wd.grad, w.grad); check("db", bd.grad, b.grad)


@pytest.mark.parametrize("B,H,W,dc,ds", [(2, 64, 64, "euclidian", "euclidian"), (1, 48, 80, "euclidian_squared", "euclidian_squared"),
                                         (2, 32, 32, "euclidian", "euclidian_squared"), (2, 88, 136, "euclidian_squared", "euclidian_squared"),
                                         (1, 52, 44, "euclidian_squared", "euclidian")])
def test_perceptual_loss_backward(cuda, B, H, W, dc, ds):
    """custom_loss with an output image that requires grad: loss values and d(total)/d(output image) vs autograd through the oracle
    (VGG-19 data gradients, max-pool + ReLU backward, InstanceNorm'd L1/L2 and mean/std terms)"""
    from mastermetastyletransfer_amd import custom_loss
    sd = seeded_state_dict(L.vgg19_shapes(), 51)
    c, s = (seeded_input(f"plb_{n}", (B, 3, H, W), 52 + i, kind="image") for i, n in enumerate("cs"))
    o = seeded_input("plb_o", (B, 3, H, W), 55, kind="image").requires_grad_(True)
    total, cl, sl = L.overall_loss(c, s, o, sd, 7.0, dc, ds)
    total.backward()
    o64 = o.detach().double().requires_grad_(True)           # float64 evaluation: the reference for the image gradient
    L.overall_loss(c.double(), s.double(), o64, {k_: v.double() for k_, v in sd.items()}, 7.0, dc, ds)[0].backward()
    loss = custom_loss(default_lambda_value=7.0, distance_content=dc, distance_style=ds, vgg_state_dict=sd).to(cuda)
    od = o.detach().to(cuda).requires_grad_(True)
    t2, c2, s2 = loss(c.to(cuda), s.to(cuda), od, output_content_and_style_loss=True)
    assert abs(float(t2) - float(total)) <= 2e-4 * abs(float(total)) and abs(float(c2) - float(cl)) <= 2e-4 * abs(float(cl))
    t2.backward()
    # Reference = the float64 oracle: with small maps the fp32 CPU oracle itself is up to 1.7e-2 off (InstanceNorm of nearly constant
    # channels; case 1x48x80, profiles/r02_grad_conditioning.txt).  |x| has a discontinuous derivative: where IN(Fc) - IN(Fcs) is at
    # rounding level an element's gradient flips by 2/(B*C*HW), so the L1 distances carry isolated differences of ~1e-3 of the
    # maximum (measured 1.02e-3) and are held to 1e-3 in L2 and 4e-3 on the maximum; the smooth squared distances to 1e-3 on both.
    got, ref = od.grad.cpu().double(), o64.grad
    l2 = float((got - ref).norm() / ref.norm())
    # r05: sizes whose maps turn odd (88 x 136 -> 11 x 17 -> 5 x 8, floor-mode pools) joined the list.  With the seeded random VGG-19 a ReLU / max-pool
    # switch that falls differently in fp32 than in float64 moves the whole gradient: at 2 x 88 x 136 the CPU fp32 oracle is 7.87e-3 in L2 from its
    # own float64 evaluation - and the HIP path sits at the same 7.87e-3 (profiles/r05_edge_sweep_mi355x.txt) - so the bounds scale with the CPU
    # fp32 oracle's own distance where that is larger
    cpu32 = o.grad.double()
    l2_cpu = float((cpu32 - ref).norm() / ref.norm()); mx_cpu = float((cpu32 - ref).abs().max() / ref.abs().max())
    assert l2 <= max(1e-3, 2 * l2_cpu), (l2, l2_cpu)
    check("d_output", got, ref, max(1e-3 if dc == "euclidian_squared" and ds == "euclidian_squared" else 4e-3, 2 * mx_cpu))


def test_rowscale_and_muladd(cuda):
    from mastermetastyletransfer_amd.autograd import MulAddFn, RowScaleAddFn
    g = torch.Generator().manual_seed(3)
    B, T, C = 3, 50, 64
    x, r, q, sg, mu = (torch.randn(B * T, C, generator=g).requires_grad_(True) for _ in range(5))
    sc = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9])
    go = torch.randn(B * T, C, generator=g)
    (r + x * sc.repeat_interleave(T)[:, None]).backward(go)
    (q * sg + mu).backward(go)
    d = [t.detach().to(cuda).requires_grad_(True) for t in (x, r, q, sg, mu)]
    o1 = RowScaleAddFn.apply(d[0], d[1], sc.to(cuda), B, T)
    o2 = MulAddFn.apply(d[2], d[3], d[4])
    check("o1", o1, r + x * sc.repeat_interleave(T)[:, None], 1e-6); check("o2", o2, q * sg + mu, 1e-6)
    o1.backward(go.to(cuda)); o2.backward(go.to(cuda))
    for n, dt, t in zip("x r q sigma mu".split(), d, (x, r, q, sg, mu)):
        check(n, dt.grad, t.grad, 1e-6)


def test_fused_adam_and_reptile(cuda):
    from mastermetastyletransfer_amd import FusedAdam, reptile_update
    g = torch.Generator().manual_seed(9)
    shapes = [(300,), (7, 5), (1024, 256), (3, 32, 3, 3), (1,)]
    ref = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    dev = [torch.nn.Parameter(p.detach().to(cuda)) for p in ref]
    o1, o2 = torch.optim.Adam(ref, lr=3e-3), FusedAdam(dev, lr=3e-3)
    for step in range(4):
        for p, d in zip(ref, dev):
            p.grad = torch.randn(p.shape, generator=g) * 10 ** float(torch.randint(-5, 2, (1,), generator=g))
            d.grad = p.grad.to(cuda)
        o1.step(); o2.step()
    for p, d in zip(ref, dev):
        assert torch.allclose(d.detach().cpu(), p.detach(), rtol=2e-6, atol=1e-7)
    theta = [torch.randn(*s, generator=g) for s in shapes]
    omega = [torch.randn(*s, generator=g) for s in shapes]
    td = [t.to(cuda) for t in theta]
    reptile_update(td, [o.to(cuda) for o in omega], 0.3)
    for t, o, d in zip(theta, omega, td):
        assert torch.allclose(d.cpu(), t + 0.3 * (o - t), rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------------------------------------------------------------------------------
# reference-pinned golden cases: StyleTransformer (train mode, explicit masks) -> Decoder -> custom_loss -> backward
# ---------------------------------------------------------------------------------------------------------------------------------------
def _model(cuda):
    from mastermetastyletransfer_amd import MasterStyleTransferModel
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True)
    return m


def _compare_grads(named_params, ref64, ref32, worst=3e-3):
    """The gradient of this loss is ill-conditioned in fp32: |.| distances and ReLU/max-pool switches are discontinuous, and
    InstanceNorm (eps 1e-5) amplifies rounding on nearly constant channels - the fp32 CPU oracle itself sits ~1e-3 (relative to each
    tensor's max-abs) from its float64 evaluation (profiles/r02_grad_conditioning.txt).  So both fp32 implementations are measured
    against the float64 oracle: every HIP gradient within max(3e-3, 4x the CPU-fp32 error) and the median over the tensors within
    max(1e-3, 2x the CPU-fp32 median).  (Each backward kernel alone meets 1e-3 against torch autograd: the tests above.)"""
    gscale = max(float(g.abs().max()) for g in ref64.values())
    eh, ec = [], []
    for n, p in named_params:
        if n not in ref64:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0 or not p.requires_grad, n
            continue
        assert p.grad is not None, n
        if n in M.ZERO_GRAD:
            assert float(p.grad.abs().max()) <= 1e-4 * gscale, n
            continue
        den = max(float(ref64[n].abs().max()), 1e-6 * gscale)
        h = float((p.grad.cpu().double() - ref64[n]).abs().max()) / den
        c = float((ref32[n].double() - ref64[n]).abs().max()) / den
        assert h <= max(worst, 4 * c), f"{n}: HIP gradient error vs float64 {h:.3e} (CPU fp32: {c:.3e})"
        eh.append(h); ec.append(c)
    mh, mc = float(np.median(eh)), float(np.median(ec))
    assert mh <= max(1e-3, 2 * mc), (mh, mc)
    return max(eh), mh, max(ec), mc


@pytest.mark.parametrize("i", range(len(M.CASES)))
def test_golden_train_cases(cuda, i):
    from mastermetastyletransfer_amd import custom_loss
    from mastermetastyletransfer_amd import train_forward as TF
    from tests.test_train_oracle_cpu import GOLD, oracle_case
    tag, B, H, W, k, dc, ds, lam, only = M.CASES[i]
    st_sd, dec_sd, vgg_sd, fc, fs, content, style = M.case_inputs(i, B, H, W)
    (total, cl, sl), ref32, ref_out = oracle_case(i)
    _, ref64, _ = oracle_case(i, torch.float64)
    m = _model(cuda)
    m.style_transformer.load_state_dict(st_sd); m.decoder.load_state_dict(dec_sd)
    m = m.to(cuda).train()
    named = [("style_transformer." + n, p) for n, p in m.style_transformer.named_parameters()] + \
            [("decoder." + n, p) for n, p in m.decoder.named_parameters()]
    if only:
        for n, p in named:
            p.requires_grad = n.startswith(only)
    loss_fn = custom_loss(default_lambda_value=lam, distance_content=dc, distance_style=ds, vgg_state_dict=vgg_sd).to(cuda)
    TF.sd_masks.set_masks(M.case_masks(i, B, k))
    try:
        decoded = m.decoder(m.style_transformer(fc.to(cuda), fs.to(cuda), k).permute(0, 3, 1, 2))
        assert TF.sd_masks.remaining() == 0
    finally:
        TF.sd_masks.set_masks(None)
    check("decoded", decoded, ref_out, 1e-3)
    t2, c2, s2 = loss_fn(content.to(cuda), style.to(cuda), decoded, output_content_and_style_loss=True)
    for got, want in ((t2, total), (c2, cl), (s2, sl)):
        assert abs(float(got) - want) <= 5e-4 * abs(want), (float(got), want)
    assert np.allclose([float(t2), float(c2), float(s2)], GOLD[f"{tag}_loss"], rtol=1e-3)     # the reference's own numbers
    t2.backward()
    wh, mh, wc, mc = _compare_grads(named, ref64, ref32)
    print(f"{tag}: gradient error vs float64 oracle: HIP worst {wh:.2e} median {mh:.2e} | CPU fp32 oracle worst {wc:.2e} median {mc:.2e}")


def test_graphed_train_step_matches_the_eager_step(cuda):
    """GraphedTrainStep (one hipGraph per k: forward, loss, backward, FusedAdam with its step count on the device) against the same steps
    run eagerly with the host-scalar Adam: same kernels in the same order, so the losses and the trained weights agree to rounding of
    the bias corrections.  StochasticDepth off (the two runs would draw different masks)."""
    from mastermetastyletransfer_amd import FusedAdam, GraphedTrainStep, custom_loss
    sds = (seeded_state_dict(O.swin_shapes(), 1), seeded_state_dict(O.style_transformer_shapes(), 2), seeded_state_dict(O.decoder_shapes(), 3))
    vgg = seeded_state_dict(L.vgg19_shapes(), 4)
    B, S = 2, 64
    batches = [(seeded_input(f"g_c{i}", (B, 3, S, S), 20 + i, kind="image").to(cuda), seeded_input(f"g_s{i}", (B, 3, S, S), 40 + i, kind="image").to(cuda))
               for i in range(3)]
    ks = [2, 1, 2, 2, 1, 2, 1]

    def build(capturable):
        m = _model(cuda)
        m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
        m = m.to(cuda).train()
        for p in m.swin_encoder.parameters():
            p.requires_grad = False
        for mod in m.modules():                                    # no random masks
            for name in ("stochastic_depth_prob", "encoder_stochastic_depth_prob", "decoder_stochastic_depth_prob"):
                if hasattr(mod, name):
                    setattr(mod, name, 0.0)
        loss_fn = custom_loss(default_lambda_value=10.0, vgg_state_dict=vgg).to(cuda)
        opt = FusedAdam([{"params": m.style_transformer.parameters()}, {"params": m.decoder.parameters()}], lr=1e-3, capturable=capturable)
        return m, loss_fn, opt
    m1, l1, o1 = build(False)
    m2, l2, o2 = build(True)
    step2 = GraphedTrainStep(m2, l2, o2)
    for i, k in enumerate(ks):
        c, s_ = batches[i % len(batches)]
        out = m1(c, s_, k); t1 = l1(c, s_, out); o1.zero_grad(); t1.backward(); o1.step()
        t2 = step2(c, s_, k)
        assert abs(float(t2) - float(t1)) <= 1e-4 * abs(float(t1)), (i, k, float(t1), float(t2))
        # param.grad after the call is THIS step's gradient, whichever k was captured or replayed last (each captured k owns its
        # gradient buffers; a replay re-binds param.grad to them)
        for (n, p1), (_, p2) in zip(m1.decoder.named_parameters(), m2.decoder.named_parameters()):
            if n in ("decoder.0.weight", "decoder.19.bias"):
                check(f"grad {n} after step {i} (k={k})", p2.grad, p1.grad, 2e-2)
        g1, g2 = m1.style_transformer.decoder.last_MLP[3].weight.grad, m2.style_transformer.decoder.last_MLP[3].weight.grad
        check(f"grad last_MLP.3.weight after step {i} (k={k})", g2, g1, 2e-2)
    assert set(step2.graphs) == {1, 2}                             # both layer counts were captured and replayed
    # loading an optimizer state after the captures keeps the moment buffers the graphs write to (values copied in place)
    addr = {p: o2.state[p]["exp_avg"].data_ptr() for g_ in o2.param_groups for p in g_["params"] if o2.state.get(p)}
    o2.load_state_dict(o2.state_dict())
    assert all(o2.state[p]["exp_avg"].data_ptr() == a for p, a in addr.items())
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if p1.requires_grad:
            check(n, p2, p1, 2e-4)
    o2.sync_steps()
    assert {o2.state[p]["step"] for g_ in o2.param_groups for p in g_["params"] if o2.state.get(p)} == {len(ks)}
    # an eval forward right after the replays sees the trained weights (the replay bumps the parameters' version counters)
    m1.eval(); m2.eval()
    with torch.no_grad():
        check("eval after graph replays", m2(batches[0][0], batches[0][1], 1), m1(batches[0][0], batches[0][1], 1), 2e-3)


@pytest.mark.parametrize("B,S", [(2, 64), (8, 256), (3, (88, 136))])
def test_full_train_step(cuda, B, S):
    """MasterStyleTransferModel in train mode (frozen Swin encoder with its StochasticDepth, k from random.seed(0)), custom_loss,
    backward, FusedAdam step - against the oracle's autograd step with the same masks (BASELINE config 4 at B=8, 256^2)."""
    from mastermetastyletransfer_amd import FusedAdam, custom_loss
    from mastermetastyletransfer_amd import train_forward as TF
    sds = (seeded_state_dict(O.swin_shapes(), 1), seeded_state_dict(O.style_transformer_shapes(), 2), seeded_state_dict(O.decoder_shapes(), 3))
    vgg = seeded_state_dict(L.vgg19_shapes(), 4)
    # (88, 136): a non-square image with an 11 x 17 feature map - every window of the StyleTransformer padded, forward and backward (r05; sides that
    # are not multiples of 8 change the output size, which the reference's loss cannot take either)
    SH, SW = S if isinstance(S, tuple) else (S, S)
    c = seeded_input("fts_c", (B, 3, SH, SW), 5, kind="image"); s = seeded_input("fts_s", (B, 3, SH, SW), 6, kind="image")
    random.seed(0)
    k = random.randint(1, 2)                                        # train_only_inner_loop.py:519 with max_layers = 2
    g = torch.Generator().manual_seed(7)
    probs = [p for p in O.SWIN_B_SD_PROBS if p > 0]
    swin_masks = [torch.empty(B).bernoulli_(1 - p, generator=g).div_(1 - p) for _ in range(2) for p in probs for _ in range(2)]
    masks = O.SDMasks.draw(O.SD_CALLS_PER_LAYER * k, B, 0.1, g)
    losses, ref32 = T.train_step_grads(c, s, k, *sds, vgg, masks, 10.0, swin_masks=swin_masks)
    _, ref64 = T.train_step_grads(c.double(), s.double(), k, *sds, vgg, masks, 10.0, swin_masks=swin_masks)
    m = _model(cuda)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(cuda).train()
    for p in m.swin_encoder.parameters():                           # train_only_inner_loop.py:254-257
        p.requires_grad = False
    loss_fn = custom_loss(default_lambda_value=10.0, vgg_state_dict=vgg).to(cuda)
    opt = FusedAdam([{"params": m.style_transformer.parameters()}, {"params": m.decoder.parameters()}], lr=1e-3)
    TF.sd_masks.set_masks(swin_masks + masks)
    try:
        out = m(c.to(cuda), s.to(cuda), k)
        assert TF.sd_masks.remaining() == 0
    finally:
        TF.sd_masks.set_masks(None)
    check("output", out, losses["output"], 1e-3)
    total, cl, sl = loss_fn(c.to(cuda), s.to(cuda), out, output_content_and_style_loss=True)
    assert abs(float(total) - float(losses["total"])) <= 5e-4 * abs(float(losses["total"]))
    opt.zero_grad(); total.backward()
    named = [("style_transformer." + n, p) for n, p in m.style_transformer.named_parameters()] + \
            [("decoder." + n, p) for n, p in m.decoder.named_parameters()]
    # the 11 x 17 map has 187 pixels per image: ONE ReLU of the first decoder layer that switches differently in fp32 than in float64 moves
    # that layer's weight gradient by ~0.5 % (measured: 6.0e-3 on decoder.0.weight, every other tensor <= 2.6x the CPU-fp32 error; the
    # decoder's backward chain replayed in float64 torch from the HIP path's own tensors agrees to 2e-7, profiles/r05_edge_sweep_mi355x.txt)
    wh, mh, wc, mc = _compare_grads(named, ref64, ref32, worst=1e-2 if isinstance(S, tuple) else 3e-3)
    print(f"B={B} S={S} k={k}: loss {float(total):.4f}; gradient error vs float64 oracle: HIP worst {wh:.2e} median {mh:.2e} | "
          f"CPU fp32 oracle worst {wc:.2e} median {mc:.2e}")
    # Adam: the device optimizer on ITS gradients vs the oracle's Adam on the same (device) gradients
    before = {n: p.detach().cpu().clone() for n, p in named}
    dev_grads = {n: p.grad.detach().cpu().clone() for n, p in named}
    opt.step()
    T.adam_step(before, dev_grads, {}, 1e-3, 1)
    for n, p in named:
        assert torch.allclose(p.detach().cpu(), before[n], rtol=1e-5, atol=2e-7), n


def test_full_train_step_swin_t(cuda):
    """the same step with the swin_T encoder (embed 96, 3 / 6 heads, StochasticDepth 0.2 over 12 blocks) and a 192-wide
    StyleTransformer / decoder with 6 heads: the widths that are not powers of two, and the decoder layers that run zero-padded"""
    from mastermetastyletransfer_amd import MasterStyleTransferModel, custom_loss
    from mastermetastyletransfer_amd import train_forward as TF
    B, S, k = 2, 128, 1                 # 16 x 16 tokens per image: the 8 x 8 maps of a 64-pixel image leave InstanceNorm 64 samples per channel
    sds = (seeded_state_dict(O.swin_shapes(embed=96, heads=(3, 6)), 11), seeded_state_dict(O.style_transformer_shapes(dim=192, heads=6), 12),
           seeded_state_dict(O.decoder_shapes(192), 13))
    vgg = seeded_state_dict(L.vgg19_shapes(), 4)
    cfg = dict(O.DEFAULT_ST_CFG, encoder_num_heads=6, decoder_num_heads=6)
    c = seeded_input("ftt_c", (B, 3, S, S), 5, kind="image"); s = seeded_input("ftt_s", (B, 3, S, S), 6, kind="image")
    g = torch.Generator().manual_seed(9)
    probs = [0.2 * n / 11.0 for n in range(1, 4)]
    swin_masks = [torch.empty(B).bernoulli_(1 - p, generator=g).div_(1 - p) for _ in range(2) for p in probs for _ in range(2)]
    masks = O.SDMasks.draw(O.SD_CALLS_PER_LAYER * k, B, 0.1, g)
    losses, ref32 = T.train_step_grads(c, s, k, *sds, vgg, masks, 10.0, cfg=cfg, swin_masks=swin_masks, swin_heads=(3, 6))
    _, ref64 = T.train_step_grads(c.double(), s.double(), k, *sds, vgg, masks, 10.0, cfg=cfg, swin_masks=swin_masks, swin_heads=(3, 6))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(swin_variant="swin_T", style_encoder_dim=192, style_decoder_dim=192, style_encoder_num_heads=6,
                                     style_decoder_num_heads=6, style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7],
                                     allow_random_swin_encoder=True)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(cuda).train()
    for p in m.swin_encoder.parameters():
        p.requires_grad = False
    loss_fn = custom_loss(default_lambda_value=10.0, vgg_state_dict=vgg).to(cuda)
    TF.sd_masks.set_masks(swin_masks + masks)
    try:
        out = m(c.to(cuda), s.to(cuda), k)
        assert TF.sd_masks.remaining() == 0
    finally:
        TF.sd_masks.set_masks(None)
    check("output", out, losses["output"], 1e-3)
    total = loss_fn(c.to(cuda), s.to(cuda), out)
    assert abs(float(total) - float(losses["total"])) <= 5e-4 * abs(float(losses["total"]))
    total.backward()
    named = [("style_transformer." + n, p) for n, p in m.style_transformer.named_parameters()] + \
            [("decoder." + n, p) for n, p in m.decoder.named_parameters()]
    wh, mh, wc, mc = _compare_grads(named, ref64, ref32)
    print(f"swin_T B={B} S={S} k={k}: gradient error vs float64 oracle: HIP worst {wh:.2e} median {mh:.2e} | CPU fp32 oracle worst {wc:.2e} median {mc:.2e}")


def test_reptile_outer_step_and_fast_adaptation(cuda):
    """meta.ReptileTrainer (train.py:402-534): omega <- theta, inner Adam steps on omega, theta += eta (omega - theta).  The
    orchestration is replayed on the CPU from the DEVICE gradients of every inner step (gradient parity is covered above): oracle Adam
    on omega, then the interpolation - theta must match.  Then fast adaptation: only StyleEncoder parameters receive gradients."""
    from mastermetastyletransfer_amd import ReptileTrainer, custom_loss, set_fast_adaptation
    from mastermetastyletransfer_amd import train_forward as TF
    B, S = 2, 64
    sds = (seeded_state_dict(O.swin_shapes(), 1), seeded_state_dict(O.style_transformer_shapes(), 2), seeded_state_dict(O.decoder_shapes(), 3))
    m = _model(cuda)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(cuda).train()
    loss_fn = custom_loss(default_lambda_value=10.0, vgg_state_dict=seeded_state_dict(L.vgg19_shapes(), 4)).to(cuda)
    tr = ReptileTrainer(m, loss_fn, inner_lr=1e-3, outer_lr=0.25, num_inner_updates=2, max_layers=2)
    names = ["style_transformer." + n for n, _ in m.style_transformer.named_parameters()] + ["decoder." + n for n, _ in m.decoder.named_parameters()]
    theta0 = {n: p.detach().cpu().clone() for n, p in zip(names, list(m.style_transformer.parameters()) + list(m.decoder.parameters()))}
    omega_params = list(tr.omega_st.parameters()) + list(tr.omega_dec.parameters())
    grads_seen = []
    tr.on_inner_step = lambda i, total: grads_seen.append({n: p.grad.detach().cpu().clone() for n, p in zip(names, omega_params)})
    style = seeded_input("rep_s", (B, 3, S, S), 6, kind="image").to(cuda)
    contents = [seeded_input(f"rep_c{i}", (B, 3, S, S), 7 + i, kind="image").to(cuda) for i in range(2)]
    torch.manual_seed(0)
    losses = tr.outer_step(style, contents, layer_counts=[1, 2])
    assert len(grads_seen) == 2 and all(np.isfinite(l) for l in losses)
    omega = {n: t.clone() for n, t in theta0.items()}
    state = {}
    for step, g in enumerate(grads_seen, 1):
        T.adam_step(omega, g, state, 1e-3, step)
    for n, p in zip(names, list(m.style_transformer.parameters()) + list(m.decoder.parameters())):
        want = theta0[n] + 0.25 * (omega[n] - theta0[n])
        assert torch.allclose(p.detach().cpu(), want, rtol=1e-5, atol=3e-7), n
    # fast adaptation (train_only_inner_loop.py:306-318)
    trainable = set_fast_adaptation(m)
    assert len(trainable) == sum(1 for _ in m.style_transformer.encoder.parameters())
    out = m(contents[0], style, 1)
    loss_fn(contents[0], style, out).backward()
    for n, p in m.named_parameters():
        if n.startswith("style_transformer.encoder."):
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n


@pytest.mark.parametrize("B,H,W,dc,ds", [(2, 64, 64, "euclidian", "euclidian"), (2, 128, 128, "euclidian_squared", "euclidian_squared")])
def test_perceptual_loss_backward_bf16_network(cuda, B, H, W, dc, ds):
    """the frozen loss network in bf16 (custom_loss.set_compute_dtype(bfloat16)): an OPTIONAL throughput mode of the train step.  Loss
    values within 3e-2; the PER-PIXEL image gradient is only roughly aligned with the float64 oracle's (measured cosine 0.93 with L1
    distances on 64x64 images, 0.79 with squared distances on 128x128): 8-bit mantissas through 13 convolutions and their data
    gradients leave uncorrelated noise on every pixel.  It averages out in the PARAMETER gradients, which are sums over all pixels
    (test_train_step_with_bf16_loss_network: cosine 1.0000 for the whole gradient, 0.998 for the worst tensor), and that is what the
    mode is for: 1.5x train-step throughput.  It is never the parity configuration; the bounds here only catch a broken kernel."""
    from mastermetastyletransfer_amd import custom_loss
    sd = seeded_state_dict(L.vgg19_shapes(), 51)
    c, s = (seeded_input(f"plb_{n}", (B, 3, H, W), 52 + i, kind="image") for i, n in enumerate("cs"))
    o = seeded_input("plb_o", (B, 3, H, W), 55, kind="image")
    o64 = o.double().requires_grad_(True)
    total, cl, sl = L.overall_loss(c.double(), s.double(), o64, {k_: v.double() for k_, v in sd.items()}, 7.0, dc, ds)
    total.backward()
    loss = custom_loss(default_lambda_value=7.0, distance_content=dc, distance_style=ds, vgg_state_dict=sd).to(cuda)
    loss.set_compute_dtype(torch.bfloat16)
    od = o.to(cuda).requires_grad_(True)
    t2, c2, s2 = loss(c.to(cuda), s.to(cuda), od, output_content_and_style_loss=True)
    assert abs(float(t2.detach()) - float(total)) <= 3e-2 * abs(float(total))
    t2.backward()
    got, ref = od.grad.cpu().double().flatten(), o64.grad.flatten()
    cos = float(torch.dot(got, ref) / (got.norm() * ref.norm()))
    l2 = float((got - ref).norm() / ref.norm())
    print(f"bf16 loss network {B}x{H}x{W} {dc}: image-gradient cosine {cos:.4f}, relative L2 {l2:.3f}")
    assert cos >= 0.7, (cos, l2)


def test_train_step_with_bf16_loss_network(cuda):
    """whole train step with the frozen VGG-19 in bf16: parameter gradients (sums over all pixels, so the per-pixel sign noise of the
    L1 distances averages out) against the float64 oracle: cosine of the concatenated gradient and the worst tensor's cosine"""
    from mastermetastyletransfer_amd import custom_loss
    from mastermetastyletransfer_amd import train_forward as TF
    B, S, k = 4, 128, 1
    sds = (seeded_state_dict(O.swin_shapes(), 1), seeded_state_dict(O.style_transformer_shapes(), 2), seeded_state_dict(O.decoder_shapes(), 3))
    vgg = seeded_state_dict(L.vgg19_shapes(), 4)
    c = seeded_input("bfl_c", (B, 3, S, S), 5, kind="image"); s = seeded_input("bfl_s", (B, 3, S, S), 6, kind="image")
    g = torch.Generator().manual_seed(7)
    probs = [p for p in O.SWIN_B_SD_PROBS if p > 0]
    swin_masks = [torch.empty(B).bernoulli_(1 - p, generator=g).div_(1 - p) for _ in range(2) for p in probs for _ in range(2)]
    masks = O.SDMasks.draw(O.SD_CALLS_PER_LAYER * k, B, 0.1, g)
    _, ref64 = T.train_step_grads(c.double(), s.double(), k, *sds, vgg, masks, 10.0, swin_masks=swin_masks)
    m = _model(cuda)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(cuda).train()
    for p in m.swin_encoder.parameters():
        p.requires_grad = False
    loss_fn = custom_loss(default_lambda_value=10.0, vgg_state_dict=vgg).to(cuda).set_compute_dtype(torch.bfloat16)
    TF.sd_masks.set_masks(swin_masks + masks)
    try:
        out = m(c.to(cuda), s.to(cuda), k)
    finally:
        TF.sd_masks.set_masks(None)
    loss_fn(c.to(cuda), s.to(cuda), out).backward()
    named = [("style_transformer." + n, p) for n, p in m.style_transformer.named_parameters()] + [("decoder." + n, p) for n, p in m.decoder.named_parameters()]
    cosines, dots, n1, n2 = [], 0.0, 0.0, 0.0
    for n, p in named:
        if n in M.ZERO_GRAD:
            continue
        a, b = p.grad.cpu().double().flatten(), ref64[n].flatten()
        cosines.append((float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)), n))
        dots += float(torch.dot(a, b)); n1 += float(a.norm() ** 2); n2 += float(b.norm() ** 2)
    whole = dots / (n1 ** 0.5 * n2 ** 0.5)
    worst = min(cosines)
    print(f"bf16 loss network, B={B} {S}x{S}: cosine of the whole parameter gradient {whole:.4f}; worst tensor {worst[0]:.4f} ({worst[1]})")
    assert whole >= 0.999 and worst[0] >= 0.99, (whole, worst)


def test_training_reduces_the_loss_and_checkpoints_roundtrip(cuda, tmp_path):
    """functional check of the whole loop the reference's trainer runs (train_only_inner_loop.py:519-575, :370-385): 12 Adam steps on one
    fixed batch with device-drawn StochasticDepth masks lower the loss; checkpoints written the reference's way
    (state_dict of style_transformer / decoder) reload into a fresh model that reproduces the eval output bit for bit"""
    from mastermetastyletransfer_amd import FusedAdam, custom_loss
    B, S = 2, 96                                                   # 96 = 12 x 8: a 12 x 12 map -> padded windows + shift
    sds = (seeded_state_dict(O.swin_shapes(), 1), seeded_state_dict(O.style_transformer_shapes(), 2), seeded_state_dict(O.decoder_shapes(), 3))
    m = _model(cuda)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(cuda).train()
    for p in m.swin_encoder.parameters():
        p.requires_grad = False
    loss_fn = custom_loss(default_lambda_value=10.0, vgg_state_dict=seeded_state_dict(L.vgg19_shapes(), 4)).to(cuda)
    opt = FusedAdam([{"params": m.style_transformer.parameters()}, {"params": m.decoder.parameters()}], lr=2e-4)
    c = seeded_input("fit_c", (B, 3, S, S), 5, kind="image").to(cuda); s = seeded_input("fit_s", (B, 3, S, S), 6, kind="image").to(cuda)
    torch.manual_seed(0)
    losses = []
    for it in range(12):
        total = loss_fn(c, s, m(c, s, 1 + it % 2))
        opt.zero_grad(); total.backward(); opt.step()
        losses.append(float(total.detach()))
    assert all(np.isfinite(losses)), losses
    assert min(losses[-3:]) < 0.9 * losses[0], losses
    torch.save(m.style_transformer.state_dict(), tmp_path / "st.pt"); torch.save(m.decoder.state_dict(), tmp_path / "dec.pt")
    m.eval()
    with torch.no_grad():
        want = m(c, s, 2)
    m2 = _model(cuda)
    m2.swin_encoder.load_state_dict(sds[0])
    m2.style_transformer.load_state_dict(torch.load(tmp_path / "st.pt")); m2.decoder.load_state_dict(torch.load(tmp_path / "dec.pt"))
    m2 = m2.to(cuda).eval()
    with torch.no_grad():
        assert torch.equal(m2(c, s, 2), want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eval_after_training_in_one_process_uses_the_new_weights(cuda, dtype):
    """FusedAdam.step and reptile_update write the parameters through raw pointers.  The inference path caches tensors derived from
    them (fused qkv concatenations, packed convolution weights, bf16 copies) keyed on the parameters' version counters, so both
    updates must bump those: validation BEFORE training, then train steps (or a Reptile outer step), then validation again has to
    see the trained weights - compared with a fresh model loaded from the state dict, bit for bit."""
    from mastermetastyletransfer_amd import FusedAdam, ReptileTrainer, custom_loss
    B, S = 2, 64
    sds = (seeded_state_dict(O.swin_shapes(), 1), seeded_state_dict(O.style_transformer_shapes(), 2), seeded_state_dict(O.decoder_shapes(), 3))
    m = _model(cuda)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(cuda)
    for p in m.swin_encoder.parameters():
        p.requires_grad = False
    loss_fn = custom_loss(default_lambda_value=10.0, vgg_state_dict=seeded_state_dict(L.vgg19_shapes(), 4)).to(cuda)
    opt = FusedAdam([{"params": m.style_transformer.parameters()}, {"params": m.decoder.parameters()}], lr=1e-3)
    c = seeded_input("ev_c", (B, 3, S, S), 5, kind="image").to(cuda); s = seeded_input("ev_s", (B, 3, S, S), 6, kind="image").to(cuda)

    def validate(model):
        model.eval(); model.set_compute_dtype(dtype)
        with torch.no_grad():
            out = model(c, s, 1).clone()
        model.set_compute_dtype(torch.float32)
        return out

    def fresh():
        m2 = _model(cuda)
        m2.swin_encoder.load_state_dict(sds[0])
        m2.style_transformer.load_state_dict({k: v.clone() for k, v in m.style_transformer.state_dict().items()})
        m2.decoder.load_state_dict({k: v.clone() for k, v in m.decoder.state_dict().items()})
        return validate(m2.to(cuda))

    before = validate(m)                                            # populates every derived-weight cache with the INITIAL weights
    m.train()
    torch.manual_seed(0)
    for _ in range(3):
        total = loss_fn(c, s, m(c, s, 1))
        opt.zero_grad(); total.backward(); opt.step()
    after = validate(m)
    assert not torch.equal(after, before), "three Adam steps at lr 1e-3 did not change the eval output: stale derived weights"
    assert torch.equal(after, fresh())
    # the Reptile outer update (theta += eta (omega - theta)) writes theta the same way
    m.train()
    tr = ReptileTrainer(m, loss_fn, inner_lr=1e-3, outer_lr=0.5, num_inner_updates=1, max_layers=1)
    tr.outer_step(s, [c], layer_counts=[1])
    after2 = validate(m)
    assert not torch.equal(after2, after)
    assert torch.equal(after2, fresh())


@pytest.mark.parametrize("i", [0, 1, 4])
def test_non_default_decoder_branches_backward(cuda, i):
    """train-mode gradients through the affine-InstanceNorm and regular-MHA decoder branches vs autograd through the oracle (fp32,
    StochasticDepth off): every parameter gradient within 1e-3 of its max-abs (a smooth loss: sum of squares of the output)"""
    import make_golden_branches as MB
    from mastermetastyletransfer_amd import StyleTransformer
    tag, affine, regular, after = MB.CASES[i]
    sd, fc, fs, cfg = MB.case(i)
    p = {k_: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k_, v in sd.items()}
    if affine and not regular:          # shared parameters: one tensor under both names, like the reference module tree
        for n, src in (("instance_norm_q", "instance_norm_Query"), ("instance_norm_k", "instance_norm_Key")):
            for wb in ("weight", "bias"):
                p[f"decoder.decoder_MHA_for_sigma_and_mu.{n}.{wb}"] = p[f"decoder.{src}.{wb}"]
    out = O.style_transformer(fc, fs, 2, p, cfg)
    (out * out).sum().backward()
    m = StyleTransformer(MB.DIM, MB.DIM, MB.NH, MB.NH, [7, 7], [7, 7], [4, 4], [4, 4], encoder_stochastic_depth_prob=0.0,
                         decoder_stochastic_depth_prob=0.0, decoder_use_instance_norm_with_affine=affine,
                         decoder_use_regular_MHA_instead_of_Swin_at_the_end=regular,
                         decoder_use_Key_instance_norm_after_linear_transformation=after)
    m.load_state_dict(sd)
    m = m.to(cuda).train()
    got = m(fc.to(cuda), fs.to(cuda), 2)
    check("out", got, out, 1e-4)
    (got * got).sum().backward()
    gscale = max(float(v.grad.abs().max()) for v in p.values() if v.is_floating_point() and v.grad is not None)
    seen = set()
    for n, prm in m.named_parameters():
        if n.endswith("Wk.bias") or id(prm) in seen:
            continue
        seen.add(id(prm))
        check(n, prm.grad, p[n].grad, floor=1e-3 * gscale)


def test_two_forwards_before_one_backward_keep_their_shared_gradients(cuda):
    """the use counts of shared parameters (k layers, 2k uses of the StyleEncoder block) belong to the forward that counted
    them.  Two train-mode forwards followed by ONE backward through both graphs must give the gradients of the two losses taken one
    after the other (before r04 the second forward reset a process-wide registry and the first graph's sums were dropped silently)."""
    m = _model(cuda).to(cuda).train()
    for p in m.swin_encoder.parameters():
        p.requires_grad = False
    c1, s1, c2, s2 = (seeded_input(f"two_{i}", (2, 3, 64, 64), 70 + i, kind="image").to(cuda) for i in range(4))
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad]

    def run(joint: bool):
        torch.manual_seed(123); torch.cuda.manual_seed(123)                 # the same StochasticDepth masks in both runs
        m.zero_grad(set_to_none=True)
        o1 = m(c1, s1, 2)
        if not joint:
            o1.square().mean().backward()
        o2 = m(c2, s2, 1)
        if joint:
            (o1.square().mean() + o2.square().mean()).backward()
        else:
            o2.square().mean().backward()
        return {n: p.grad.detach().clone() for n, p in named if p.grad is not None}
    seq, joint = run(False), run(True)
    assert seq.keys() == joint.keys() and len(seq) > 40
    gscale = max(float(g.abs().max()) for g in seq.values())
    errs = sorted((float((joint[n] - seq[n]).abs().max()) / max(float(seq[n].abs().max()), 1e-6 * gscale), n) for n in seq if n not in M.ZERO_GRAD)
    print("two forwards, one backward: worst tensors", errs[-3:])
    assert errs[-1][0] <= 1e-4, errs[-3:]         # (the three Wk.bias gradients that are zero in exact arithmetic are rounding noise)


def test_capturable_adam_refuses_a_parameter_that_joins_late(cuda):
    """with capturable=True the param group has ONE device step counter; a tensor whose first gradient arrives after the group
    has stepped would be bias-corrected for the wrong step - raise, as the non-capturable path does for mismatching step counts"""
    from mastermetastyletransfer_amd import FusedAdam
    from mastermetastyletransfer_amd.ops import MasterHipError
    a, b = (torch.nn.Parameter(torch.randn(64, device=cuda)) for _ in range(2))
    opt = FusedAdam([a, b], lr=1e-3, capturable=True)
    a.grad = torch.randn_like(a)
    opt.step()
    b.grad = torch.randn_like(b)
    with pytest.raises(MasterHipError, match="share their step count"):
        opt.step()


def test_capturable_adam_keeps_the_late_joiner_guard_across_a_resume(cuda):
    """the guard's counter is not part of the state dict; load_state_dict rebuilds it from the loaded step counts, so a
    parameter without saved state still cannot join a resumed group at device step t > 1"""
    from mastermetastyletransfer_amd import FusedAdam
    from mastermetastyletransfer_amd.ops import MasterHipError
    a, b = (torch.nn.Parameter(torch.randn(64, device=cuda)) for _ in range(2))
    opt = FusedAdam([a, b], lr=1e-3, capturable=True)
    a.grad = torch.randn_like(a)
    opt.step(); opt.step()
    sd = opt.state_dict()
    a2, b2 = (torch.nn.Parameter(t.detach().clone()) for t in (a, b))
    fresh = FusedAdam([a2, b2], lr=1e-3, capturable=True)
    fresh.load_state_dict(sd)
    a2.grad = a.grad.clone()
    fresh.step()                                                   # the resumed tensor steps on
    assert int(fresh.state_dict()["state"][0]["step"]) == 3
    b2.grad = torch.randn_like(b2)
    with pytest.raises(MasterHipError, match="share their step count"):
        fresh.step()
