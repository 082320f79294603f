"""Golden vectors of the StyleDecoder's NON-DEFAULT branches from the *imported reference* (build container only):
    python tests/golden/make_golden_branches.py   ->   tests/golden/golden_branches.npz

decoder_use_instance_norm_with_affine (style_transformer.py:983-984), decoder_use_regular_MHA_instead_of_Swin_at_the_end (:1060-1119)
and decoder_use_Key_instance_norm_after_linear_transformation = False (:470-472), through the unmodified reference StyleTransformer
(eval mode).  The oracle must agree while generating (<= 1e-5 of max|ref|): that is its pin for these branches."""
import os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
REF = os.environ.get("MASTER_REFERENCE", "/root/reference")

# (tag, affine, regular_mha, key_in_after_linear)
CASES = [("affine", True, False, True), ("regular_mha", False, True, True), ("regular_mha_key_in_before", False, True, False),
         ("swin_key_in_before", False, False, False), ("affine_key_in_before", True, False, False)]
DIM, NH, SEED = 64, 2, 21


def case(i):
    """(state dict, Fc, Fs, cfg) of case i - rebuilt anywhere from seeds"""
    from oracle import master_oracle as O
    from oracle.seeded import seeded_input, seeded_state_dict
    tag, affine, regular, after = CASES[i]
    shapes = O.style_transformer_shapes(DIM, NH, affine_in=affine, regular_mha=regular)
    sd = seeded_state_dict(shapes, SEED + i)
    if affine and not regular:        # the attention module's instance_norm_q / _k ARE the decoder's modules (shared parameters)
        for n, src in (("instance_norm_q", "instance_norm_Query"), ("instance_norm_k", "instance_norm_Key")):
            for wb in ("weight", "bias"):
                sd[f"decoder.decoder_MHA_for_sigma_and_mu.{n}.{wb}"] = sd[f"decoder.{src}.{wb}"]
    fc = seeded_input(f"br{i}_fc", (2, 9, 12, DIM), SEED + i); fs = seeded_input(f"br{i}_fs", (2, 9, 12, DIM), SEED + 10 + i)
    cfg = dict(O.DEFAULT_ST_CFG, encoder_num_heads=NH, decoder_num_heads=NH, decoder_use_instance_norm_with_affine=affine,
               decoder_use_regular_MHA_instead_of_Swin_at_the_end=regular, decoder_use_Key_instance_norm_after_linear_transformation=after)
    return sd, fc, fs, cfg


def main():
    from make_golden import _install_torchvision_stub
    _install_torchvision_stub()
    sys.path.insert(0, REF)
    from codes import style_transformer as R
    from oracle import master_oracle as O
    torch.set_grad_enabled(False)
    out = {}
    for i, (tag, affine, regular, after) in enumerate(CASES):
        sd, fc, fs, cfg = case(i)
        m = R.StyleTransformer(DIM, DIM, NH, NH, [7, 7], [7, 7], [4, 4], [4, 4], decoder_use_instance_norm_with_affine=affine,
                               decoder_use_regular_MHA_instead_of_Swin_at_the_end=regular,
                               decoder_use_Key_instance_norm_after_linear_transformation=after).eval()
        assert set(m.state_dict()) == set(sd), sorted(set(m.state_dict()) ^ set(sd))
        m.load_state_dict(sd)
        for k in (1, 2):
            ref = m(fc, fs, k)
            got = O.style_transformer(fc, fs, k, sd, cfg)
            err = float((got - ref).abs().max())
            print(f"{tag} k={k}: oracle-vs-reference max|d| = {err:.3e} (ref max {float(ref.abs().max()):.2f})")
            assert err <= 1e-5 * max(1.0, float(ref.abs().max())), tag
            out[f"{tag}_k{k}"] = ref.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "golden_branches.npz"), **out)
    print("wrote golden_branches.npz", len(out), "vectors")


if __name__ == "__main__":
    main()
