"""folded LayerNorm + Linear of the split-bf16 mode against the LayerNorm launch + Linear pair, per shape.  usage: python profiles/tools/lnfold_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mastermetastyletransfer_amd import ops
dev = torch.device("cuda:0")


def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M, N, K in [(262144, 384, 128), (65536, 768, 256), (32768, 768, 256)]:
    x = torch.randn(M, K, device=dev) + 0.5; w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    g = torch.rand(K, device=dev) + 0.5; bt = torch.randn(K, device=dev)
    ws, cs, bf = ops.fold_layernorm(w, b, g, bt); w0 = ops.split_weights(w)
    tl = t(lambda: ops.layernorm(x, g, bt)); xn = ops.layernorm(x, g, bt)
    tg = t(lambda: ops.linear(xn, w0, b)); tf = t(lambda: ops.linear_ln_folded(x, ws, cs, bf, 1e-5))
    print(f"{M}x{N}x{K}: LayerNorm {tl:.1f} us + Linear {tg:.1f} us = {tl + tg:.1f}; folded {tf:.1f} us")
