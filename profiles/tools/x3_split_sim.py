"""CPU simulation of the split-operand parity modes (TEST / MEASUREMENT TOOL, not product code).

Runs the CPU oracle's whole forward (oracle/master_oracle.py) with every contraction (Linear, convolution, Q.K^T, P.V) replaced by the
three-product split form the HIP kernels use,

    a.w ~= a_lo.w_hi + a_hi.w_lo + a_hi.w_hi,   x_hi = round16(x), x_lo = round16(x - x_hi),

for two 16-bit operand formats - bf16 (8 significant bits, fp32's exponent range: what `set_compute_dtype("bf16x3")` ships) and IEEE half
(11 significant bits, exponent range 2^-24 .. 65504) - and compares the stylized image with the float64 evaluation of the same oracle.
Products and sums inside a contraction are evaluated in float64 and rounded to fp32 once (the MFMA's fp32 accumulation adds fp32-level
error on top; the exact-fp32 column shows that level).  Everything that is not a contraction runs in fp32 as in the kernels.

    python profiles/tools/x3_split_sim.py [--sizes 128 256] [--seeds 4 5 6] [--k 1]
"""
import argparse
import os
import sys
import types

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import master_oracle as O                     # noqa: E402
from oracle.seeded import seeded_input, seeded_state_dict  # noqa: E402


def split(x, fmt, ftz=False):
    """x (fp32) -> (hi, lo) as float64 tensors holding 16-bit-representable values."""
    dt = torch.bfloat16 if fmt == "bf16" else torch.float16
    hi = x.to(dt).float()
    lo = (x - hi).to(dt).float()
    if ftz and fmt == "f16":
        tiny = 2.0 ** -14
        hi = torch.where(hi.abs() < tiny, torch.zeros_like(hi), hi)
        lo = torch.where(lo.abs() < tiny, torch.zeros_like(lo), lo)
    return hi.double(), lo.double()


class Mode:
    def __init__(self, fmt, ftz=False, wscale=0):
        self.fmt, self.ftz, self.wscale = fmt, ftz, wscale

    def contract(self, fn, a, w):
        if self.fmt == "f32":
            return fn(a, w)
        s = 2.0 ** self.wscale
        ah, al = split(a, self.fmt, self.ftz)
        wh, wl = split(w * s, self.fmt, self.ftz)
        out = fn(al, wh) + fn(ah, wl) + fn(ah, wh)
        return (out / s).float()


def run(mode, c, s, k, sds):
    real_F = TF
    shim = types.SimpleNamespace(**{n: getattr(real_F, n) for n in dir(real_F) if not n.startswith("__")})

    def linear(x, w, b=None):
        y = mode.contract(lambda a, ww: a @ ww.t(), x, w)
        return y if b is None else y + b

    def conv2d(x, w, b=None, stride=1, padding=0):
        y = mode.contract(lambda a, ww: real_F.conv2d(a, ww, None, stride=stride, padding=padding), x, w)
        return y if b is None else y + b.view(1, -1, 1, 1)

    shim.linear, shim.conv2d = linear, conv2d
    orig_mm = torch.Tensor.__matmul__

    def matmul(a, b):
        if a.dtype != torch.float32 or mode.fmt == "f32":
            return orig_mm(a, b)
        return mode.contract(lambda x, y: orig_mm(x, y), a, b)

    old_F = O.F
    O.F = shim
    torch.Tensor.__matmul__ = matmul
    try:
        with torch.no_grad():
            return O.full_forward(c, s, k, *sds)
    finally:
        O.F = old_F
        torch.Tensor.__matmul__ = orig_mm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[128, 256])
    ap.add_argument("--seeds", type=int, nargs="+", default=[4, 14, 24])
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    sds = (seeded_state_dict(O.swin_shapes(), 1), seeded_state_dict(O.style_transformer_shapes(), 2), seeded_state_dict(O.decoder_shapes(), 3))
    sds64 = tuple({n: (t.double() if t.is_floating_point() else t) for n, t in sd.items()} for sd in sds)
    modes = [("exact fp32", Mode("f32")), ("bf16 x3", Mode("bf16")), ("half x3", Mode("f16")), ("half x3, subnormals flushed", Mode("f16", ftz=True)),
             ("half x3, weights * 2^8", Mode("f16", wscale=8)), ("half x3, weights * 2^8, flushed", Mode("f16", ftz=True, wscale=8))]
    print(f"max |d| of the stylized image against the float64 oracle, k = {args.k}, batch {args.batch}; columns: " + " | ".join(n for n, _ in modes))
    worst = [0.0] * len(modes)
    for size in args.sizes:
        for seed in args.seeds:
            c = seeded_input("smoke_c" if seed == 4 else f"sim_c{seed}", (args.batch, 3, size, size), seed, kind="image")
            s = seeded_input("smoke_s" if seed == 4 else f"sim_s{seed}", (args.batch, 3, size, size), seed + 1, kind="image")
            with torch.no_grad():
                ref = O.full_forward(c.double(), s.double(), args.k, *sds64)
            errs = []
            for i, (_, m) in enumerate(modes):
                e = float((run(m, c, s, args.k, sds).double() - ref).abs().max())
                errs.append(e); worst[i] = max(worst[i], e)
            print(f"size {size:4d} seed {seed:3d} ref max {float(ref.abs().max()):6.3f}: " + " | ".join(f"{e:.2e}" for e in errs), flush=True)
    print("worst: " + " | ".join(f"{e:.2e}" for e in worst))


if __name__ == "__main__":
    main()
