"""Golden vectors of the perceptual loss from the *imported reference* codes/loss.py (build container only):
    python tests/golden/make_golden_loss.py   ->   tests/golden/golden_loss.npz

loss.py imports torchvision (absent): in-process stand-ins provide `torchvision.transforms` and the two names it takes from
codes.utils; torch.load / os.path.exists are patched for the constructor so that it receives an nn.Sequential with torchvision's
vgg19().features[:30] layout and oracle.seeded weights.  Everything the loss computes is the reference's own code.
While generating, oracle/loss_oracle.py must agree bit for bit - that is its pin.  Only seeds, sizes and outputs are stored."""
import os, sys, types
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
REF = os.environ.get("MASTER_REFERENCE", "/root/reference")

CASES = [("euclidian", "euclidian", 2, 64, 64), ("euclidian_squared", "euclidian_squared", 1, 48, 80), ("euclidian", "euclidian_squared", 2, 32, 32)]


def seeded_vgg_and_images(i, B, H, W):
    from oracle.seeded import seeded_state_dict, seeded_input
    from oracle import loss_oracle as L
    sd = seeded_state_dict(L.vgg19_shapes(), 70 + i)
    imgs = [seeded_input(f"loss{i}_{n}", (B, 3, H, W), 80 + 3 * i + j, kind="image") for j, n in enumerate("cso")]
    return sd, imgs


def main():
    from oracle import loss_oracle as L
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms"); tvm = types.ModuleType("torchvision.models")
    for n in ("swin_transformer", "vgg19", "VGG19_Weights", "vgg19_bn", "VGG19_BN_Weights"):
        setattr(tvm, n, None)
    tv.transforms = tvt; tv.models = tvm
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.models": tvm})
    sys.path.insert(0, REF)
    from codes import loss as R                                   # the reference, unmodified
    torch.set_grad_enabled(False)
    out = {}
    for i, (dc, ds, B, H, W) in enumerate(CASES):
        sd, (c, s, o) = seeded_vgg_and_images(i, B, H, W)
        layers = []
        for j in range(30):
            if j in L.VGG19_CONVS:
                ci, co = L.VGG19_CONVS[j]; conv = nn.Conv2d(ci, co, 3, padding=1)
                conv.weight.copy_(sd[f"{j}.weight"]); conv.bias.copy_(sd[f"{j}.bias"]); layers.append(conv)
            elif j in L.VGG19_POOLS:
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers.append(nn.ReLU(inplace=True))
        seq = nn.Sequential(*layers)
        real_load, real_exists = torch.load, os.path.exists
        torch.load = lambda *a, **k: seq
        os.path.exists = lambda p: True
        try:
            ref = R.custom_loss("/nonexistent", distance_content=dc, distance_style=ds).eval()
        finally:
            torch.load, os.path.exists = real_load, real_exists
        total, cl, sl = ref(c, s, o, output_content_and_style_loss=True)
        t2, c2, s2 = L.overall_loss(c, s, o, sd, 10.0, dc, ds)
        d = max(abs(float(total - t2)), abs(float(cl - c2)), abs(float(sl - s2)))
        print(f"case {i} {dc}/{ds} {B}x{H}x{W}: total {float(total):.6f} content {float(cl):.6f} style {float(sl):.6f}  oracle max|d| = {d:.3e}")
        assert d == 0.0, "oracle/loss_oracle.py disagrees with the reference loss"
        taps = ref.feature_extractor_model(o)
        out[f"case{i}"] = np.array([float(total), float(cl), float(sl)], np.float64)
        out[f"case{i}_relu5_1"] = taps[3].numpy()
    path = os.path.join(HERE, "golden_loss.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
