"""The golden-vector cases (name -> how to rebuild inputs/weights), shared by the CPU oracle test and the
GPU parity test.  Mirrors tests/golden/make_golden.py; weights/inputs come from oracle.seeded."""
from oracle import master_oracle as O
from oracle.seeded import seeded_input, seeded_state_dict

DIM, NH = 64, 2
SWA_CASES = ((7, 7, (0, 0)), (14, 14, (0, 0)), (14, 14, (3, 3)), (9, 9, (4, 4)), (9, 12, (4, 4)), (14, 9, (4, 0)))
BLOCK_CASES = ((False, True), (True, False), (True, True), (False, False))     # (use LayerNorm, exclude_MLP_after)


def attn_shapes(dim=DIM, nh=NH, names=("Wq", "Wk", "Wv", "proj")):
    s = {"relative_position_bias_table": (169, nh), "relative_position_index": (2401,)}
    for n in names:
        s[f"{n}.weight"] = (dim, dim)
        s[f"{n}.bias"] = (dim,)
    return s


def block_shapes(norm: bool, excl: bool, dim=DIM, nh=NH):
    s = {}
    if norm:
        s["norm1.weight"] = (dim,); s["norm1.bias"] = (dim,)
        if not excl:
            s["norm2.weight"] = (dim,); s["norm2.bias"] = (dim,)
    s.update({"attn." + k: v for k, v in attn_shapes(dim, nh).items()})
    if not excl:
        s.update({"mlp.0.weight": (4 * dim, dim), "mlp.0.bias": (4 * dim,), "mlp.3.weight": (dim, 4 * dim), "mlp.3.bias": (dim,)})
    return s


def swa_case(H, W, shift):
    tag = f"swa_H{H}_W{W}_s{shift[0]}{shift[1]}"
    sd = seeded_state_dict(attn_shapes(), seed=11)
    xs = [seeded_input(f"{tag}_{n}", (2, H, W, DIM), 11) for n in "qkv"]
    return tag, sd, xs


def block_case(norm, excl):
    tag = f"block_norm{int(norm)}_excl{int(excl)}"
    sd = seeded_state_dict(block_shapes(norm, excl), seed=13)
    xs = [seeded_input(f"{tag}_{n}", (2, 9, 9, DIM), 13) for n in "qkv"]
    return tag, sd, xs


def st_small():
    sd = seeded_state_dict(O.style_transformer_shapes(DIM, NH), seed=14)
    abcd = [seeded_input(f"sigmu_{n}", (2, 9, 9, DIM), 14) for n in "abcd"]
    fc, fs = seeded_input("st_fc", (2, 12, 16, DIM), 14), seeded_input("st_fs", (2, 12, 16, DIM), 14)
    cfg = dict(O.DEFAULT_ST_CFG, encoder_num_heads=NH, decoder_num_heads=NH)
    return sd, abcd, fc, fs, cfg


def st_256():
    sd = seeded_state_dict(O.style_transformer_shapes(256, 8), seed=15)
    fc, fs = seeded_input("st256_fc", (1, 32, 32, 256), 15), seeded_input("st256_fs", (1, 32, 32, 256), 15)
    return sd, fc, fs


def decoder_case(hw):
    return seeded_state_dict(O.decoder_shapes(256), seed=16), seeded_input(f"dec_x{hw}", (1, 256, hw, hw), 16)
