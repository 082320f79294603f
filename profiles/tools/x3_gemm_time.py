"""Per-shape time of the split-operand GEMM / convolution at the shapes of the bs = 32 forward (MEASUREMENT TOOL): HIP events around REPS
back-to-back launches per shape.  The tile form is read once per process, so run it once per MASTER_X3_TILE setting and compare rows:
    for t in 0 7; do MASTER_X3_TILE=$t python profiles/tools/x3_gemm_time.py --fmt f16; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mastermetastyletransfer_amd import ops

dev = torch.device("cuda:0")
FMT = sys.argv[sys.argv.index("--fmt") + 1] if "--fmt" in sys.argv else "f16"
REPS = 20


def t_us(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(REPS): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / REPS)
    return best


print(f"MASTER_X3_TILE={os.environ.get('MASTER_X3_TILE', '0')} fmt {FMT}")
for M, N, K, act, res in [(262144, 384, 128, 0, 0), (262144, 128, 128, 0, 1), (65536, 256, 512, 0, 0), (65536, 768, 256, 0, 0), (65536, 256, 256, 0, 1),
                          (32768, 768, 256, 0, 0), (32768, 512, 256, 0, 0), (32768, 256, 256, 0, 1), (32768, 1024, 256, 1, 0), (32768, 256, 1024, 0, 1)]:
    a = torch.randn(M, K, device=dev); w = ops.split_weights(torch.randn(N, K, device=dev) / K ** 0.5, fmt=FMT)
    b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev) if res else None
    us = t_us(lambda: ops.linear(a, w, b, r2=r, act=act))
    print(f"linear {M:7d} x {N:4d} x {K:4d} res{res}: {us:8.1f} us  {3 * 2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s (3 products)", flush=True)
for B, H, Cin, Cout, up in [(32, 32, 256, 128, 0), (32, 32, 128, 128, 1), (32, 64, 128, 128, 0), (32, 64, 128, 64, 0), (32, 64, 64, 64, 1), (32, 128, 64, 32, 0)]:
    x = torch.randn(B, H, H, Cin, device=dev); w = ops.split_weights(torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** .5), fmt=FMT)
    b = torch.randn(Cout, device=dev)
    us = t_us(lambda: ops.conv3x3_reflect(x, w, b, bool(up), True))
    Ho = H * 2 if up else H
    print(f"conv {B}x{H}x{H} {Cin:3d}->{Cout:3d} up{up}: {us:8.1f} us  {3 * 2.0 * B * Ho * Ho * Cout * 9 * Cin / us / 1e6:7.1f} TFLOP/s (3 products)", flush=True)
