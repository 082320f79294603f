"""split-mode window attention timing on the forward's shapes (A/B two builds with MASTER_HIP_LIB; MEASUREMENT TOOL).
usage: python profiles/tools/attn_x3_time.py [bf16|f16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mastermetastyletransfer_amd import ops
from mastermetastyletransfer_amd.window_plan import window_plan
dev = torch.device("cuda:0")
FMT = sys.argv[1] if len(sys.argv) > 1 else "f16"
# (images, H = W of the token map, channels, heads, shift): stage 1 of the Swin encoder (both images: 64), stage 2, the style transformer
for B, H, C, heads, shift in [(64, 64, 128, 4, 0), (64, 64, 128, 4, 3), (64, 32, 256, 8, 0), (32, 32, 256, 8, 3)]:
    T = H * H
    qkv = torch.randn(B * T, 3 * C, device=dev)
    plan = window_plan(H, H, [7, 7], [shift, shift], dev)
    kp = torch.randn(3, C, device=dev)
    b64 = torch.zeros(heads, 64, 64, device=dev); b64[:, :, 49:] = -1e30
    f = lambda: ops.window_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], None, kp[0], kp[1], None, b64, plan.src, plan.region if shift else None,
                                B, T, heads, plan.nW, plan.N, 32 ** -0.5, x3=FMT)
    for _ in range(3): f()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    mb = B * T * 4 * C * 4 / 1e6
    print(f"window attention x3 B={B} {H}x{H} C={C} heads={heads} shift={shift}: {best:7.1f} us  {mb:6.0f} MB  {mb / best:.2f} TB/s", flush=True)
