"""Golden mapping of the reference's MS-Swin -> StyleTransformer weight remapper (build container only):
    python tests/golden/make_golden_remap.py   ->   tests/golden/golden_remap.npz

Runs the unmodified reference codes/load_pretrained_weights_to_style_transformer.py (it imports only torch) on a seeded fake block
checkpoint and records, for every StyleTransformer key the reference overwrote, WHICH piece of the block it now holds (found by
equality): a table of names, no tensors.  mastermetastyletransfer_amd.weights.remap_ms_swin_block must agree bit for bit here."""
import os, sys, tempfile
from collections import OrderedDict
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
REF = os.environ.get("MASTER_REFERENCE", "/root/reference")

BLOCK_SHAPES = {"0.weight": (256,), "0.bias": (256,), "1.relative_position_bias_table": (169, 8), "1.relative_position_index": (49, 49),
                "1.qkv.weight": (768, 256), "1.qkv.bias": (768,), "1.proj.weight": (256, 256), "1.proj.bias": (256,), "3.weight": (256,),
                "3.bias": (256,), "4.fc1.weight": (1024, 256), "4.fc1.bias": (1024,), "4.fc2.weight": (256, 1024), "4.fc2.bias": (256,)}


def seeded_block(seed=77):
    from oracle import master_oracle as O
    from oracle.seeded import seeded_state_dict
    return {k: (O.relative_position_index((7, 7)).view(49, 49) if "index" in k else seeded_state_dict({k: s}, seed)[k]) for k, s in BLOCK_SHAPES.items()}


def main():
    from oracle import master_oracle as O
    from oracle.seeded import seeded_state_dict
    from mastermetastyletransfer_amd.weights import remap_ms_swin_block, split_block
    sys.path.insert(0, REF)
    from codes.load_pretrained_weights_to_style_transformer import get_pretained_weight_loaded_style_transformer_state_dict as ref
    blk = seeded_block()
    path = os.path.join(tempfile.mkdtemp(), "block.pth")
    torch.save(blk, path)
    st = seeded_state_dict(O.style_transformer_shapes(), 5)
    r = ref(OrderedDict((k, v.clone()) for k, v in st.items()), path, 256, 256, 4, 4, [7, 7], [7, 7])
    m = remap_ms_swin_block(blk, OrderedDict((k, v.clone()) for k, v in st.items()), decoder_norm_layer=nn.LayerNorm)
    assert set(r) == set(m) and all(torch.equal(r[k], m[k]) for k in r), "weights.remap_ms_swin_block disagrees with the reference loader"
    pieces = split_block(blk)
    dst, src = [], []
    for k in r:
        hits = [n for n, t in pieces.items() if t.shape == r[k].shape and torch.equal(t, r[k])]
        if hits and not k.endswith("relative_position_index"):
            assert len(hits) == 1, (k, hits)
            dst.append(k); src.append(hits[0])
    np.savez_compressed(os.path.join(HERE, "golden_remap.npz"), dst=np.array(dst), src=np.array(src))
    print(len(dst), "remapped keys recorded")


if __name__ == "__main__":
    main()
