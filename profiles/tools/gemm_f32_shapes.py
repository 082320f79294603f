"""fp32 (exact-fp32 MFMA) GEMM / implicit-GEMM convolution rates on the shapes of the parity mode (bs=32 forward), BASELINE config 5 and the
train step (config 4).  Run on the GPU box; select the kernel with MASTER_GEMM_F32=reg|dma (read once per process).
usage: python profiles/tools/gemm_f32_shapes.py [check] [x3]      x3: the split-bf16 kernels (gemm_x3.hip; MASTER_X3_TILE=1..3 forces a tile)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mastermetastyletransfer_amd import ops

dev = torch.device("cuda:0")
check = "check" in sys.argv[1:]
X3 = "x3" in sys.argv[1:]
prep = (lambda w: ops.split_weights(w)) if X3 else (lambda w: w)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


print(f"variant MASTER_GEMM_F32={os.environ.get('MASTER_GEMM_F32', '(default dma)')} SMALL={os.environ.get('MASTER_GEMM_F32_SMALL', '(default 1)')}")
print(f"--- linear {'x3 (MASTER_X3_TILE=' + os.environ.get('MASTER_X3_TILE', '0') + ')' if X3 else 'fp32'}: M N K act res | us | TFLOP/s")
tot_f = tot_t = 0.0
for M, N, K, act, res in [(262144, 384, 128, 0, 0), (262144, 128, 128, 0, 1), (262144, 512, 128, 1, 0), (262144, 128, 512, 0, 1),
                          (65536, 256, 512, 0, 0), (65536, 768, 256, 0, 0), (65536, 256, 256, 0, 1), (65536, 1024, 256, 1, 0), (65536, 256, 1024, 0, 1),
                          (32768, 768, 256, 0, 0), (32768, 256, 256, 0, 1), (32768, 1024, 256, 1, 0), (32768, 256, 1024, 0, 1),
                          (8192, 768, 256, 0, 0), (8192, 256, 256, 0, 1), (8192, 1024, 256, 1, 0), (8192, 256, 1024, 0, 1), (8000, 256, 256, 0, 0)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev)
    wk = prep(w)
    us = timeit(lambda: ops.linear(a, wk, b, r2=r, act=act, out=out))
    tot_f += 2.0 * M * N * K; tot_t += us
    line = f"{M:7d} {N:5d} {K:5d} {act} {res} | {us:8.1f} | {2.0*M*N*K/us/1e6:8.1f}"
    if check and M <= 65536:
        ref = a.double() @ w.double().t() + b.double()
        if act == 1: ref = torch.nn.functional.gelu(ref)
        if res: ref = ref + r.double()
        line += f" | max err {float((out.double() - ref).abs().max()):.2e}"
    print(line, flush=True)
print(f"linear total: {tot_f/tot_t/1e6:.1f} TFLOP/s")

print("--- conv3x3 fp32: B H W Cin Cout up zero | us | TFLOP/s")
tot_f = tot_t = 0.0
convs = [(32, 32, 256, 128, 0, 0), (32, 32, 128, 128, 1, 0), (32, 64, 128, 128, 0, 0), (32, 64, 128, 64, 0, 0), (32, 64, 64, 64, 1, 0),
         (32, 128, 64, 32, 0, 0), (32, 128, 32, 32, 1, 0), (32, 256, 32, 3, 0, 0),
         (8, 32, 256, 128, 0, 0), (8, 64, 128, 128, 0, 0),
         (24, 256, 32, 64, 0, 1), (24, 256, 64, 64, 0, 1), (24, 128, 64, 128, 0, 1), (24, 128, 128, 128, 0, 1), (24, 64, 128, 256, 0, 1),
         (24, 64, 256, 256, 0, 1), (24, 32, 256, 512, 0, 1), (24, 32, 512, 512, 0, 1), (24, 16, 512, 512, 0, 1)]
for B, H, Cin, Cout, up, zero in convs:
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** .5)
    b = torch.randn(Cout, device=dev)
    Ho = H * 2 if up else H
    if X3 and (zero or Cout == 3):
        continue
    wk = prep(w)
    if zero:
        fn = lambda: ops.conv3x3_zero(x, w, b, relu=True)
    else:
        fn = lambda: ops.conv3x3_reflect(x, wk, b, bool(up), True, out_nchw=(Cout == 3))
    us = timeit(fn, reps=5)
    fl = 2.0 * B * Ho * Ho * Cout * 9 * Cin
    tot_f += fl; tot_t += us
    line = f"{B:3d} {H:4d} {Cin:4d} {Cout:4d} {up} {zero} | {us:8.1f} | {fl/us/1e6:8.1f}"
    if check and B * Ho * Ho <= 8 * 64 * 64 * 4:
        got = fn()
        xi = x.permute(0, 3, 1, 2).double()
        if up: xi = torch.nn.functional.interpolate(xi, scale_factor=2, mode="nearest")
        wi = w.permute(0, 3, 1, 2).double()
        if zero: ref = torch.nn.functional.conv2d(xi, wi, b.double(), padding=1)
        else: ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xi, (1, 1, 1, 1), mode="reflect"), wi, b.double())
        ref = torch.relu(ref)
        g = got.double() if Cout == 3 else got.permute(0, 3, 1, 2).double()
        line += f" | max err {float((g - ref).abs().max()):.2e}"
    print(line, flush=True)
print(f"conv total: {tot_f/tot_t/1e6:.1f} TFLOP/s")
