"""Parity of the three fp32-storage modes over many content / style pairs (MEASUREMENT TOOL; GPU + the CPU oracle as checker).

For every (size, seed, k) the stylized image of the HIP path in the exact-fp32 mode, the split mode with IEEE-half halves ("f16x3") and
the split mode with bf16 halves ("bf16x3") is compared with the CPU oracle (oracle/master_oracle.py) on the same seeded weights and
inputs: max |d|, next to the largest output magnitude of the pair.  The worst case per mode is what DESIGN.md quotes.

    python profiles/tools/parity_sweep.py [--sizes 64 128 200 256 320 512] [--seeds 4 14 24 34 44] [--ks 1 2] [--weights-seed 1]
"""
import argparse
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mastermetastyletransfer_amd import MasterStyleTransferModel  # noqa: E402
from oracle import master_oracle as O                             # noqa: E402  (checker)
from oracle.seeded import seeded_input, seeded_state_dict         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[64, 128, 200, 256, 320, 512])
    ap.add_argument("--seeds", type=int, nargs="+", default=[4, 14, 24, 34, 44])
    ap.add_argument("--ks", type=int, nargs="+", default=[1])
    ap.add_argument("--weights-seed", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    ws = args.weights_seed
    sds = (seeded_state_dict(O.swin_shapes(), ws), seeded_state_dict(O.style_transformer_shapes(), ws + 1), seeded_state_dict(O.decoder_shapes(), ws + 2))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True)
    m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
    m = m.to(dev).eval()
    modes = [("exact fp32", torch.float32), ("f16x3", "f16x3"), ("bf16x3", "bf16x3")]
    worst = {n: (0.0, None) for n, _ in modes}
    worst_rel = {n: 0.0 for n, _ in modes}
    print(f"seeded weights {ws}..{ws + 2}; max |d| of the stylized image against the CPU oracle;  columns: " + " | ".join(n for n, _ in modes), flush=True)
    n = 0
    for k in args.ks:
        for size in args.sizes:
            for seed in args.seeds:
                c = seeded_input("smoke_c" if seed == 4 else f"sim_c{seed}", (1, 3, size, size), seed, kind="image")
                s = seeded_input("smoke_s" if seed == 4 else f"sim_s{seed}", (1, 3, size, size), seed + 1, kind="image")
                with torch.no_grad():
                    ref = O.full_forward(c, s, k, *sds)
                    top = float(ref.abs().max())
                    errs = []
                    for name, mode in modes:
                        m.set_compute_dtype(mode)
                        got = m(c.to(dev), s.to(dev), k).cpu()
                        assert torch.isfinite(got).all(), (name, size, seed, k)
                        e = float((got - ref).abs().max())
                        errs.append(e)
                        if e > worst[name][0]:
                            worst[name] = (e, (size, seed, k, top))
                        worst_rel[name] = max(worst_rel[name], e / top)
                n += 1
                print(f"k {k} size {size:4d} seed {seed:3d} max|ref| {top:9.3f}: " + " | ".join(f"{e:.2e}" for e in errs), flush=True)
    print(f"{n} pairs.  worst max |d| (size, seed, k, max|ref|):")
    for name, _ in modes:
        print(f"  {name:10s} {worst[name][0]:.3e} at {worst[name][1]};  worst max|d| / max|ref| {worst_rel[name]:.2e}")


if __name__ == "__main__":
    main()
