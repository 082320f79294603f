"""per-launch floor inside a hipGraph (20 launches per graph): a near-empty kernel, the token-stationary linear and the fused MLP vs M.
usage: python profiles/tools/launch_floor.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mastermetastyletransfer_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16

def graph_time(fn, n=20):
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best

small = torch.zeros(256, device=dev, dtype=dt); out_small = torch.zeros(256, device=dev)
print(f"convert of 256 elements (one workgroup): {graph_time(lambda: ops.convert(small, torch.float32)):6.1f} us per launch")
K = N = 256
w = (torch.randn(N, K, device=dev) / 16).to(dt); b = torch.randn(N, device=dev)
w1 = (torch.randn(1024, 256, device=dev) / 16).to(dt); w2 = (torch.randn(256, 1024, device=dev) / 32).to(dt); b1 = torch.randn(1024, device=dev); b2 = torch.randn(256, device=dev)
for M in (128, 1024, 4096, 8192, 16384, 32768, 65536, 131072):
    x = torch.randn(M, K, device=dev).to(dt)
    t_lin = graph_time(lambda: ops.linear(x, w, b))
    t_mlp = graph_time(lambda: ops.mlp(x, w1, b1, w2, b2, res=x))
    print(f"M={M:7d} ({(M + 127) // 128:5d} tiles): linear 256->256 {t_lin:7.1f} us   fused MLP {t_mlp:7.1f} us")
