"""Per-op microbenchmark on the shapes of the bs=32 256x256 forward (run on the GPU box).
usage: python profiles/bench_ops.py [fp32|bf16] [gemm|attn|ln|conv|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mastermetastyletransfer_amd import ops
from mastermetastyletransfer_amd.window_plan import window_plan

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
what = sys.argv[2] if len(sys.argv) > 2 else "all"
dev = torch.device("cuda:0")
es = 2 if dt == torch.bfloat16 else 4


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


if what in ("gemm", "all"):
    print(f"--- linear ({dt}) : M N K act res | us | TFLOP/s | alg GB/s")
    shapes = [(262144, 384, 128, 0, 0), (262144, 128, 128, 0, 1), (262144, 512, 128, 1, 0), (262144, 128, 512, 0, 1),
              (65536, 256, 512, 0, 0), (65536, 768, 256, 0, 0), (65536, 256, 256, 0, 1), (65536, 1024, 256, 1, 0),
              (65536, 256, 1024, 0, 1), (32768, 768, 256, 0, 0), (32768, 512, 256, 0, 0), (32768, 256, 256, 0, 1),
              (32768, 1024, 256, 1, 0), (32768, 256, 1024, 0, 1)]
    for M, N, K, act, res in shapes:
        a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).to(dt) if res else None
        out = torch.empty(M, N, device=dev, dtype=dt)
        us = timeit(lambda: ops.linear(a, w, b, r2=r, act=act, out=out))
        by = es * (M * K + N * K + M * N * (1 + res))
        print(f"{M:7d} {N:5d} {K:5d} {act} {res} | {us:8.1f} | {2.0*M*N*K/us/1e6:8.1f} | {by/us/1e3:8.1f}")

if what in ("conv", "all"):
    print(f"--- conv3x3 ({dt}) : B H W Cin Cout up | us | TFLOP/s | alg GB/s")
    for B, H, Cin, Cout, up in [(32, 32, 256, 128, 0), (32, 32, 128, 128, 1), (32, 64, 128, 128, 0), (32, 64, 128, 64, 0),
                                (32, 64, 64, 64, 1), (32, 128, 64, 32, 0), (32, 128, 32, 32, 1), (32, 256, 32, 3, 0)]:
        x = torch.randn(B, H, H, Cin, device=dev).to(dt); w = (torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** .5)).to(dt)
        b = torch.randn(Cout, device=dev)
        Ho = H * 2 if up else H
        us = timeit(lambda: ops.conv3x3_reflect(x, w, b, bool(up), True, out_nchw=(Cout == 3)))
        by = es * (x.numel() + w.numel()) + (4 if Cout == 3 else es) * B * Ho * Ho * Cout
        print(f"{B:3d} {H:4d} {Cin:4d} {Cout:4d} {up} | {us:8.1f} | {2.0*B*Ho*Ho*Cout*9*Cin/us/1e6:8.1f} | {by/us/1e3:8.1f}")

if what in ("attn", "all"):
    print(f"--- window_attn ({dt}) : B H heads nv shift | us | alg TFLOP/s | alg GB/s")
    for B, H, heads, nv, shift in [(64, 64, 4, 1, 0), (64, 64, 4, 1, 3), (64, 32, 8, 1, 0), (64, 32, 8, 1, 3), (32, 32, 8, 1, 4), (32, 32, 8, 2, 4)]:
        C = heads * 32; T = H * H
        plan = window_plan(H, H, (7, 7), (shift, shift), dev)
        qkv = torch.randn(B * T, 3 * C, device=dev).to(dt)
        v1 = torch.randn(B * T, C, device=dev).to(dt) if nv == 2 else None
        pad = torch.randn(3 * C, device=dev).to(dt)
        bias = torch.zeros(heads, 64, 64, device=dev); bias[:, :, 49:] = -1e30
        fn = lambda: ops.window_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:] if nv == 1 else v1, v1 if nv == 2 else None,
                                     pad[C:2 * C], pad[2 * C:], pad[2 * C:] if nv == 2 else None, bias, plan.src, plan.region,
                                     B, T, heads, plan.nW, plan.N, 32 ** -0.5)
        us = timeit(fn)
        fl = 2.0 * B * plan.nW * heads * 49 * 49 * 32 * (1 + nv)
        by = es * B * T * C * (2 + 2 * nv)
        print(f"{B:3d} {H:3d} {heads} {nv} {shift} | {us:8.1f} | {fl/us/1e6:8.2f} | {by/us/1e3:8.1f}")

if what in ("ln", "all"):
    print(f"--- layernorm / instnorm ({dt}) : M C | us | alg GB/s")
    for M, C in [(262144, 128), (65536, 256), (32768, 256), (65536, 512)]:
        x = torch.randn(M, C, device=dev).to(dt); g = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
        us = timeit(lambda: ops.layernorm(x, g, b))
        print(f"LN {M:7d} {C:5d} | {us:8.1f} | {2*es*M*C/us/1e3:8.1f}")
    x = torch.randn(32 * 1024, 256, device=dev).to(dt)
    us = timeit(lambda: ops.instnorm(x, 32, 1024))
    print(f"IN 32x1024 256 | {us:8.1f} | {2*es*x.numel()/us/1e3:8.1f}")
    img = torch.randn(64, 3, 256, 256, device=dev); w = torch.randn(128, 48, device=dev); v = torch.randn(128, device=dev)
    us = timeit(lambda: ops.patch_embed(img, w, v, v, v, dt))
    print(f"patch_embed 64x3x256x256 | {us:8.1f} | {(img.numel()*4 + 64*4096*128*es)/us/1e3:8.1f}")
