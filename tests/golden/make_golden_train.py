"""Golden vectors of one TRAIN STEP from the *imported reference* (build container only):
    python tests/golden/make_golden_train.py   ->   tests/golden/golden_train.npz

What runs is the reference's own code: codes/style_transformer.py::StyleTransformer in train mode, codes/decoder.py::Decoder,
codes/loss.py::custom_loss, then total_loss.backward() (train_only_inner_loop.py:519-575) and torch.optim.Adam(lr).step()
(:468-478, :573-575).  torchvision is absent from this image, so the two helpers style_transformer.py imports from it are
in-process stand-ins (as in make_golden.py); StochasticDepth follows torchvision's row mode
(noise = bernoulli(1-p)/(1-p) per sample, x * noise) but takes its noise from an explicit list, because CPU and GPU random
streams differ and the parity tests must hand the same masks to every implementation.  The frozen Swin encoder
(train_only_inner_loop.py:254-257; torchvision arithmetic, not importable here) is replaced by seeded feature maps Fc, Fs.

While generating, oracle/train_oracle.py must agree with the reference on the loss and on EVERY parameter gradient
(<= 2e-6 of the gradient's max-abs) and on the parameters after one Adam step - that is its pin.  Stored: seeds, losses and
for every parameter (L2 norm, sum, first 8 values) of the gradient: 69 tensors x 10 floats.
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

# (tag, B, feature H, W, k, distance_content, distance_style, lambda, trainable prefix or None)
CASES = [("k1_l1", 2, 8, 8, 1, "euclidian", "euclidian", 10.0, None),
         ("k2_l2", 2, 12, 8, 2, "euclidian_squared", "euclidian_squared", 4.0, None),
         ("k1_fast_adaptation", 2, 8, 8, 1, "euclidian", "euclidian", 10.0, "style_transformer.encoder.")]
# Batch 1 is avoided on purpose: with B = 1 this image's torch (2.10, CPU) returns gradients for the reference graph that
# disagree with central finite differences of the reference's own forward (the permuted NHWC->NCHW view handed to InstanceNorm2d
# is ambiguous between memory formats when N = 1), while the oracle's gradients match the finite differences.  B >= 2 agrees
# to 6e-7.  The BASELINE config trains with batch 8.
SD_P = 0.1
LR = 1e-3

MASK_QUEUE = []

# Gradients that are ZERO in exact arithmetic: a key bias shifts every logit of a softmax row by the same q.b (Wk.bias of the two
# q/k/v attentions), and the sigma/mu attention instance-normalises K after Wk, which removes a constant.  What autograd returns for
# them is rounding noise (1e-8 of the other gradients) that differs between any two implementations, and Adam turns that noise into
# +-lr steps: they are checked against zero, not against each other.
ZERO_GRAD = ("style_transformer.encoder.shared_MHA_without_MLP.attn.Wk.bias", "style_transformer.decoder.MHA_self_attn.attn.Wk.bias",
             "style_transformer.decoder.decoder_MHA_for_sigma_and_mu.Wk.bias")


def _install_torchvision_stub():
    class StochasticDepth(nn.Module):          # torchvision.ops.StochasticDepth(p, "row") with the noise supplied
        def __init__(self, p, mode):
            super().__init__(); self.p, self.mode = p, mode
        def forward(self, x):
            if not self.training or self.p == 0.0:
                return x
            assert self.mode == "row"
            noise = MASK_QUEUE.pop(0)
            return x * noise.view(-1, *([1] * (x.dim() - 1)))

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
                                             "torchvision.ops.misc", "torchvision.transforms", "torchvision.models")}
    mods["torchvision.ops.stochastic_depth"].StochasticDepth = StochasticDepth
    mods["torchvision.ops.misc"].MLP = MLP
    for n in ("swin_transformer", "vgg19", "VGG19_Weights", "vgg19_bn", "VGG19_BN_Weights"):
        setattr(mods["torchvision.models"], n, None)
    mods["torchvision"].transforms = mods["torchvision.transforms"]; mods["torchvision"].models = mods["torchvision.models"]
    sys.modules.update(mods)


def case_inputs(i, B, H, W):
    """everything a test needs to rebuild the case without the reference"""
    from oracle import master_oracle as O, loss_oracle as L
    from oracle.seeded import seeded_input, seeded_state_dict
    st = seeded_state_dict(O.style_transformer_shapes(256, 8), 140 + i)
    dec = seeded_state_dict(O.decoder_shapes(256), 150 + i)
    vgg = seeded_state_dict(L.vgg19_shapes(), 160 + i)
    fc = seeded_input(f"train{i}_fc", (B, H, W, 256), 170 + i)
    fs = seeded_input(f"train{i}_fs", (B, H, W, 256), 171 + i)
    content = seeded_input(f"train{i}_c", (B, 3, 8 * H, 8 * W), 172 + i, kind="image")
    style = seeded_input(f"train{i}_s", (B, 3, 8 * H, 8 * W), 173 + i, kind="image")
    return st, dec, vgg, fc, fs, content, style


def case_masks(i, B, k):
    from oracle import master_oracle as O
    g = torch.Generator().manual_seed(180 + i)
    return O.SDMasks.draw(O.SD_CALLS_PER_LAYER * k, B, SD_P, g)


def grad_summary(g: torch.Tensor) -> np.ndarray:
    f = g.detach().double().flatten()
    head = torch.zeros(8, dtype=torch.float64); head[: min(8, f.numel())] = f[:8]
    return torch.cat([torch.stack([f.norm(), f.sum()]), head]).numpy()


def main():
    _install_torchvision_stub()
    sys.path.insert(0, REF)
    from codes import style_transformer as R            # the reference, unmodified
    from codes.decoder import Decoder as RefDecoder
    from codes import loss as RL
    from oracle import loss_oracle as L, train_oracle as T

    out = {}
    for i, (tag, B, H, W, k, dc, ds, lam, only) in enumerate(CASES):
        st_sd, dec_sd, vgg_sd, fc, fs, content, style = case_inputs(i, B, H, W)
        masks = case_masks(i, B, k)

        # ---- the reference modules --------------------------------------------------------------------------------------
        st = R.StyleTransformer(256, 256, 8, 8, [7, 7], [7, 7], [4, 4], [4, 4], encoder_stochastic_depth_prob=SD_P,
                                decoder_stochastic_depth_prob=SD_P)
        st.load_state_dict(st_sd)
        dec = RefDecoder(); dec.load_state_dict(dec_sd)
        st.train(); dec.train()
        layers = []
        for j in range(30):
            if j in L.VGG19_CONVS:
                ci, co = L.VGG19_CONVS[j]; conv = nn.Conv2d(ci, co, 3, padding=1)
                with torch.no_grad():
                    conv.weight.copy_(vgg_sd[f"{j}.weight"]); conv.bias.copy_(vgg_sd[f"{j}.bias"])
                layers.append(conv)
            elif j in L.VGG19_POOLS:
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers.append(nn.ReLU(inplace=True))
        seq = nn.Sequential(*layers)
        real_load, real_exists = torch.load, os.path.exists
        torch.load = lambda *a, **kw: seq
        os.path.exists = lambda p: True
        try:
            loss_fn = RL.custom_loss("/nonexistent", default_lambda_value=lam, distance_content=dc, distance_style=ds)
        finally:
            torch.load, os.path.exists = real_load, real_exists
        named = [("style_transformer." + n, p) for n, p in st.named_parameters()] + [("decoder." + n, p) for n, p in dec.named_parameters()]
        if only:                                                    # fast adaptation: train_only_inner_loop.py:306-318
            for n, p in named:
                p.requires_grad = n.startswith(only)
        opt = torch.optim.Adam([{"params": st.parameters()}, {"params": dec.parameters()}], lr=LR)

        MASK_QUEUE[:] = [m.clone() for m in masks]
        decoded = dec(st(fc, fs, k).permute(0, 3, 1, 2))            # full_model.py:221-224 after the (frozen) encoder
        total, cl, sl = loss_fn(content, style, decoded, output_content_and_style_loss=True)
        opt.zero_grad(); total.backward(); opt.step()
        assert not MASK_QUEUE, "mask count mismatch"
        ref_grads = {n: p.grad for n, p in named if p.requires_grad}

        # ---- the oracle on the same inputs ------------------------------------------------------------------------------
        def leaf(prefix, sd):
            o = {}
            for n, t in sd.items():
                t = t.clone()
                if t.is_floating_point() and (only is None or (prefix + n).startswith(only)):
                    t.requires_grad_(True)
                o[n] = t
            return o
        o_st, o_dec = leaf("style_transformer.", st_sd), leaf("decoder.", dec_sd)
        t2, c2, s2, out2 = T.train_forward_loss(fc, fs, content, style, k, o_st, o_dec, vgg_sd, masks, lam, dc, ds)
        t2.backward()
        o_grads = {("style_transformer." + n): t.grad for n, t in o_st.items() if t.requires_grad}
        o_grads.update({("decoder." + n): t.grad for n, t in o_dec.items() if t.requires_grad})
        assert set(o_grads) == set(ref_grads), set(o_grads) ^ set(ref_grads)
        dl = max(abs(float(a.detach() - b.detach())) / abs(float(a.detach())) for a, b in ((total, t2), (cl, c2), (sl, s2)))
        worst, gscale = 0.0, max(float(g.abs().max()) for g in ref_grads.values())
        for n, g in ref_grads.items():
            if n in ZERO_GRAD:
                assert float(g.abs().max()) <= 1e-6 * gscale and float(o_grads[n].abs().max()) <= 1e-6 * gscale, n
                continue
            d = float((g - o_grads[n]).abs().max()) / max(float(g.abs().max()), 1e-6 * gscale)
            if d > 1e-6:
                print(f"   {n}: |g|max {float(g.abs().max()):.3e} max|d| {float((g - o_grads[n]).abs().max()):.3e}")
            worst = max(worst, d)
        # the oracle's Adam on the REFERENCE's gradients vs the reference optimizer's result (a first Adam step is
        # lr * g / (|g| + eps): on elements with |g| ~ eps it amplifies last-bit gradient differences, so the optimizer is
        # pinned separately from the gradients)
        params = {n: t.detach().clone() for n, t in {**{"style_transformer." + a: b for a, b in st_sd.items()},
                                                     **{"decoder." + a: b for a, b in dec_sd.items()}}.items() if n in o_grads}
        T.adam_step(params, {n: g.detach() for n, g in ref_grads.items()}, {}, LR, 1)
        da = max(float((params[n] - p.detach()).abs().max()) for n, p in named if p.requires_grad)
        total, cl, sl = total.detach(), cl.detach(), sl.detach()
        print(f"{tag}: total {float(total):.6f} content {float(cl):.6f} style {float(sl):.6f} | oracle: loss rel {dl:.2e}, "
              f"grad worst rel {worst:.2e} ({len(ref_grads)} tensors), Adam max|d| {da:.2e}")
        assert dl <= 1e-6 and worst <= 2e-6 and da <= 1e-7, "oracle/train_oracle.py disagrees with the reference train step"
        out[f"{tag}_loss"] = np.array([float(total), float(cl), float(sl)], np.float64)
        out[f"{tag}_names"] = np.array(sorted(ref_grads))
        out[f"{tag}_grads"] = np.stack([grad_summary(ref_grads[n]) for n in sorted(ref_grads)])
        out[f"{tag}_adam"] = np.stack([grad_summary(dict(named)[n].detach() - (st_sd if n.startswith("style_transformer.") else dec_sd)[n.split(".", 1)[1]])
                                       for n in sorted(ref_grads)])
    path = os.path.join(HERE, "golden_train.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
