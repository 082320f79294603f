"""fp32 linear forward at the train step's matrix sizes (8 192 tokens), 20 launches per hipGraph"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mastermetastyletransfer_amd import ops
dev = torch.device("cuda:0")
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
for M, N, K in [(8192, 256, 256), (8192, 768, 256), (8192, 1024, 256), (8192, 256, 1024), (2048, 256, 256), (32768, 256, 256)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / 16; b = torch.randn(N, device=dev)
    t = graph_time(lambda: ops.linear(a, w, b))
    print(f"M={M:6d} N={N:5d} K={K:5d}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TFLOP/s")
