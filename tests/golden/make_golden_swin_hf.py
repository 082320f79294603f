"""Pin for the torchvision stage of the path (Swin-B ``features[:4]``; reference codes/utils.py:89-93, codes/full_model.py:219-220).

torchvision is absent from the reference tree, from this image and from the GPU box, so the oracle's restatement of that stage
(oracle/master_oracle.py: swin_window_attention / swin_block / patch_merging / swin_encoder) cannot be checked against torchvision here.
This script checks it against an INDEPENDENT implementation of the same published architecture that IS in the image: Hugging Face
``transformers``' ``SwinModel`` (modeling_swin.py; the version is stored in the fixture).  It

  1. builds a two-stage SwinModel with Swin-B's geometry (patch 4, embed 128, depths (2, 2), heads (4, 8), window 7, mlp ratio 4, qkv bias,
     LayerNorm eps 1e-5, exact GELU, no dropout / stochastic depth), in eval mode;
  2. loads it with the oracle's seeded weights (oracle.seeded, keyed by torchvision's names) through the key map below - the fused
     ``attn.qkv`` rows are cut into the q | k | v thirds, ``features.2`` is the first stage's PatchMerging;
  3. runs it on seeded images and stores the hidden state behind the second stage's blocks, BEFORE the model's final LayerNorm (that is
     what ``features[:4]`` returns), as NHWC [B, H/8, W/8, 256].

Sizes: 64 (8x8 maps: smaller than two windows, everything padded), 128 (shifted windows with real neighbours), 200 (50x50 -> padded to 56;
25x25 at stage 2: odd, so PatchMerging pads and the windows pad again), 256 (the benchmark's geometry, one image).  Only seeds, the
version string and expected outputs are stored (tests/golden/golden_swin_hf.npz, ~1.3 MB); the weights are re-created from seeds.

    python tests/golden/make_golden_swin_hf.py        (CPU, ~20 s; needs `transformers`)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import master_oracle as O                                  # noqa: E402
from oracle.seeded import seeded_input, seeded_state_dict              # noqa: E402

CASES = [("s64", 1, 64, 64, 11), ("s128", 2, 128, 128, 12), ("s200", 1, 200, 200, 13), ("s256", 1, 256, 256, 14), ("s96x160", 1, 96, 160, 15)]
WEIGHTS_SEED = 1


def hf_state_dict(tv):
    """torchvision-keyed Swin weights (oracle.swin_shapes names) -> the keys of transformers' two-stage SwinModel"""
    out = {"embeddings.patch_embeddings.projection.weight": tv["0.0.weight"], "embeddings.patch_embeddings.projection.bias": tv["0.0.bias"],
           "embeddings.norm.weight": tv["0.2.weight"], "embeddings.norm.bias": tv["0.2.bias"],
           "encoder.layers.0.downsample.reduction.weight": tv["2.reduction.weight"],
           "encoder.layers.0.downsample.norm.weight": tv["2.norm.weight"], "encoder.layers.0.downsample.norm.bias": tv["2.norm.bias"]}
    for stage, feat in ((0, 1), (1, 3)):
        for i in range(2):
            t, h = f"{feat}.{i}.", f"encoder.layers.{stage}.blocks.{i}."
            C = tv[t + "attn.proj.weight"].shape[0]
            for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                out[h + f"attention.{n}.weight"] = tv[t + "attn.qkv.weight"][j * C:(j + 1) * C]
                out[h + f"attention.{n}.bias"] = tv[t + "attn.qkv.bias"][j * C:(j + 1) * C]
            out[h + "attention.o_proj.weight"] = tv[t + "attn.proj.weight"]; out[h + "attention.o_proj.bias"] = tv[t + "attn.proj.bias"]
            out[h + "attention.relative_position_bias.relative_position_bias_table"] = tv[t + "attn.relative_position_bias_table"]
            for a, b in (("norm1", "layernorm_before"), ("norm2", "layernorm_after"), ("mlp.0", "mlp.fc1"), ("mlp.3", "mlp.fc2")):
                out[h + b + ".weight"] = tv[t + a + ".weight"]; out[h + b + ".bias"] = tv[t + a + ".bias"]
    return out


def build_hf_model(tv):
    import transformers
    from transformers import SwinConfig, SwinModel
    cfg = SwinConfig(image_size=256, patch_size=4, num_channels=3, embed_dim=128, depths=[2, 2], num_heads=[4, 8], window_size=7, mlp_ratio=4.0,
                     qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, drop_path_rate=0.0, hidden_act="gelu",
                     layer_norm_eps=1e-5, use_absolute_embeddings=False)
    m = SwinModel(cfg, add_pooling_layer=False).eval()
    sd = hf_state_dict(tv)
    own = m.state_dict()
    # everything the model owns is either mapped or one of: the final LayerNorm (not part of features[:4]; its output is not used) and
    # index buffers the model computes for itself (an independent check of the oracle's relative_position_index)
    unmapped = [k for k in own if k not in sd and not k.startswith("layernorm.") and "relative_position_index" not in k]
    assert not unmapped, unmapped
    assert all(k in own and own[k].shape == v.shape for k, v in sd.items()), [k for k in sd if k not in own or own[k].shape != sd[k].shape]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("layernorm.") or "relative_position_index" in k for k in missing), (missing, unexpected)
    return m, transformers.__version__


@torch.no_grad()
def hf_features(m, img):
    """hidden state behind the last stage's blocks, before SwinModel's final LayerNorm: NHWC [B, H/8, W/8, 256]"""
    out = m(pixel_values=img, output_hidden_states=True, return_dict=True, interpolate_pos_encoding=False)
    h = out.hidden_states[-1]                              # [B, tokens, 256]
    B, _, Hi, Wi = img.shape
    Ho, Wo = -(-(Hi // 4) // 2), -(-(Wi // 4) // 2)        # PatchMerging pads odd maps: ceil
    assert h.shape == (B, Ho * Wo, 256), (h.shape, Ho, Wo)
    return h.reshape(B, Ho, Wo, 256)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tv = seeded_state_dict(O.swin_shapes(), WEIGHTS_SEED)
    m, version = build_hf_model(tv)
    data = {"transformers_version": np.array(version), "weights_seed": np.array(WEIGHTS_SEED),
            "cases": np.array([f"{n}:{b}:{h}:{w}:{s}" for n, b, h, w, s in CASES])}
    for name, B, H, W, seed in CASES:
        img = seeded_input("swin_hf_" + name, (B, 3, H, W), seed, kind="image")
        ref = hf_features(m, img)
        mine = O.swin_encoder(img, tv)
        d = float((ref - mine).abs().max())
        print(f"{name}: {tuple(ref.shape)} max|ref| {float(ref.abs().max()):.3f}  oracle vs transformers {version} max|d| {d:.3e}", flush=True)
        data["out_" + name] = ref.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_swin_hf.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
