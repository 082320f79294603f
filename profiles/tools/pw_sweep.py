"""window attention (persistent bf16 kernel) time vs number of windows: is there a fixed cost?  usage: python profiles/tools/pw_sweep.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mastermetastyletransfer_amd import ops
from mastermetastyletransfer_amd.window_plan import window_plan
dev = torch.device("cuda:0"); dt = torch.bfloat16
H, heads, shift = 32, 8, 4
C = heads * 32; T = H * H
plan = window_plan(H, H, (7, 7), (shift, shift), dev)
bias = torch.zeros(heads, 64, 64, device=dev); bias[:, :, 49:] = -1e30
for B in (1, 2, 4, 8, 16, 32, 64, 128):
    qkv = torch.randn(B * T, 3 * C, device=dev).to(dt); pad = torch.randn(3 * C, device=dev).to(dt)
    fn = lambda: ops.window_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], None, pad[C:2 * C], pad[2 * C:], None, bias, plan.src, plan.region,
                                 B, T, heads, plan.nW, plan.N, 32 ** -0.5)
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph()                      # graph replay: GPU time without the host's launch cost
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"B={B:4d} windows={B * plan.nW:6d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per launch (20 launches in one graph)")
