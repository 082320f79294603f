"""Fused attention + proj + residual (master_window_attn_proj_fwd) against the two launches it replaces, on the step's shapes.
usage: python profiles/tools/attn_proj_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mastermetastyletransfer_amd import ops
from mastermetastyletransfer_amd.window_plan import window_plan

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B, H, shift in [(32, 32, 4), (64, 32, 3), (64, 32, 0)]:
    C, heads, T = 256, 8, H * H
    plan = window_plan(H, H, (7, 7), (shift, shift), dev)
    qkv = torch.randn(B * T, 3 * C, device=dev).to(BF)
    res = torch.randn(B * T, C, device=dev).to(BF)
    pad = torch.randn(3 * C, device=dev).to(BF)
    wp = (torch.randn(C, C, device=dev) / 16).to(BF); bp = torch.randn(C, device=dev)
    bias = torch.randn(heads, 64, 64, device=dev) * 0.02; bias[:, :, 49:] = -1e30
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]

    def fused():
        return ops.window_attn_proj(q, k, v, pad[C:2 * C], pad[2 * C:], bias, plan.src, plan.region, wp, bp, res, B, T, heads, plan.nW, plan.N, 32 ** -0.5)

    def attn_only():
        return ops.window_attn(q, k, v, None, pad[C:2 * C], pad[2 * C:], None, bias, plan.src, plan.region, B, T, heads, plan.nW, plan.N, 32 ** -0.5)[0]
    o = attn_only()

    def proj_only():
        return ops.linear(o, wp, bp, r2=res)

    def split():
        return ops.linear(attn_only(), wp, bp, r2=res)
    a, b_ = fused(), split()
    err = float((a.float() - b_.float()).abs().max())
    print(f"B={B} {H}x{H} shift {shift} ({B * plan.nW} windows): fused {timeit(fused):7.1f} us | attention {timeit(attn_only):7.1f} + proj {timeit(proj_only):7.1f} "
          f"= split {timeit(split):7.1f} us | max diff {err:.3e}", flush=True)
