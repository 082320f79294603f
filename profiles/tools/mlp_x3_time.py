"""fused split-bf16 MLP timing on the forward's shapes (A/B two builds with MASTER_HIP_LIB).  usage: python profiles/tools/mlp_x3_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mastermetastyletransfer_amd import ops
dev = torch.device("cuda:0")
for M, C, ln in [(32768, 256, 1), (32768, 256, 0), (65536, 256, 1), (262144, 128, 1)]:
    x = torch.randn(M, C, device=dev)
    FM = os.environ.get("X3_FMT", "f16"); w1 = ops.split_weights(torch.randn(4 * C, C, device=dev) / C ** 0.5, fmt=FM); w2 = ops.split_weights(torch.randn(C, 4 * C, device=dev) / (4 * C) ** 0.5, layout=1, fmt=FM)
    b1 = torch.randn(4 * C, device=dev) * 0.1; b2 = torch.randn(C, device=dev) * 0.1
    lg = torch.ones(C, device=dev) if ln else None; lb = torch.zeros(C, device=dev) if ln else None
    f = lambda: ops.mlp(x, w1, b1, w2, b2, res=x, ln_gamma=lg, ln_beta=lb)
    for _ in range(3): f()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(f"mlp x3 M={M} C={C} ln={ln}: {best:7.1f} us  {2 * 2 * M * C * 4 * C / best / 1e6:6.1f} TF", flush=True)
