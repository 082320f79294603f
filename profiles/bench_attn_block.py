import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mastermetastyletransfer_amd import ops
from mastermetastyletransfer_amd.window_plan import window_plan
dev = torch.device("cuda:0"); C, heads = 128, 4
for B, H, shift in [(64, 64, 0), (64, 64, 3), (1, 64, 3), (4, 64, 3)]:
    T = H * H; plan = window_plan(H, H, (7, 7), (shift, shift), dev)
    x = torch.randn(B * T, C, device=dev).bfloat16(); wqkv = (torch.randn(3 * C, C, device=dev) / 11).bfloat16(); wp = (torch.randn(C, C, device=dev) / 11).bfloat16()
    bqkv = torch.randn(3 * C, device=dev); bp = torch.randn(C, device=dev); g = torch.ones(C, device=dev)
    bias = torch.zeros(heads, 64, 64, device=dev); bias[:, :, 49:] = -1e30
    f = lambda: ops.attn_block(x, g, g, wqkv, bqkv, wp, bp, bias, plan.src, plan.region, B, T, heads, plan.nW, plan.N, 32 ** -0.5)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1) / 20 * 1e3
    nwin = B * plan.nW; per = us / max(1, -(-nwin // 256))
    print(f"attn_block B={B} H={H} shift={shift}: {us:8.1f} us  windows={nwin} -> {per:.2f} us per window per WG")
