"""Bitwise repeatability of every split (F32X3 / F16X3) kernel at the sizes of the bs = 32 forward: each op runs REPS times on the same inputs
under load (no host sync between launches) and all outputs must be identical.  A race between LDS-DMA / ds_write / ds_read phases of a
kernel shows up here long before it shows up in a tolerance test.
usage: python profiles/tools/x3_determinism.py [reps] [--fmt bf16|f16] [--only linear,conv,convs,mlp,tail,attn]
The tile forms behind MASTER_X3_TILE=1..6 and MASTER_MLP_X3_C128=8w are read once per process: run the tool once per setting
(tests/test_gpu_x3.py::test_x3_tile_variants_repeatable does)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mastermetastyletransfer_amd import ops

dev = torch.device("cuda:0")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
FMT = sys.argv[sys.argv.index("--fmt") + 1] if "--fmt" in sys.argv else "bf16"
ONLY = set(sys.argv[sys.argv.index("--only") + 1].split(",")) if "--only" in sys.argv else {"linear", "conv", "convs", "mlp", "tail", "attn"}
bad = 0
_split = ops.split_weights
ops_split = lambda w, layout=0: _split(w, layout=layout, fmt=FMT)


def check(name, fn):
    global bad
    outs = [fn() for _ in range(REPS)]
    torch.cuda.synchronize()
    n = sum(0 if torch.equal(outs[0], o) else 1 for o in outs[1:])
    if n:
        d = max(float((outs[0] - o).abs().max()) for o in outs[1:])
        cnt = max(int((outs[0] != o).sum()) for o in outs[1:])
        bad += 1
        print(f"NONDETERMINISTIC {name}: {n} of {REPS - 1} repeats differ, max |d| {d:.3e}, up to {cnt} elements", flush=True)
    else:
        print(f"ok {name}", flush=True)


g = torch.Generator(device="cpu").manual_seed(0)
for M, N, K, act, res in [] if "linear" not in ONLY else [(262144, 384, 128, 0, 0), (262144, 128, 128, 0, 1), (65536, 256, 512, 0, 0), (65536, 768, 256, 0, 0), (65536, 256, 256, 0, 1),
                          (32768, 768, 256, 0, 0), (32768, 512, 256, 0, 0), (32768, 256, 256, 0, 1), (32768, 1024, 256, 1, 0), (32768, 256, 1024, 0, 1)]:
    a = torch.randn(M, K, device=dev); w = ops_split(torch.randn(N, K, device=dev) / K ** 0.5)
    b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev) if res else None
    check(f"linear {M}x{N}x{K} act{act} res{res}", lambda: ops.linear(a, w, b, r2=r, act=act))
for B, H, Cin, Cout, up in [] if "conv" not in ONLY else [(32, 32, 256, 128, 0), (32, 32, 128, 128, 1), (32, 64, 128, 128, 0), (32, 64, 128, 64, 0), (32, 64, 64, 64, 1), (32, 128, 64, 32, 0),
                            (32, 128, 32, 32, 1)]:
    x = torch.randn(B, H, H, Cin, device=dev); w = ops_split(torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** .5))
    b = torch.randn(Cout, device=dev)
    check(f"conv {B}x{H}x{H} {Cin}->{Cout} up{up}", lambda: ops.conv3x3_reflect(x, w, b, bool(up), True))
for B, H, Cin, Cout, up in [] if "convs" not in ONLY else [(32, 32, 128, 128, 1), (32, 64, 128, 64, 0), (32, 64, 64, 64, 1), (32, 128, 64, 32, 0), (32, 128, 32, 32, 1)]:
    x0 = torch.randn(B, H, H, 32, device=dev); w0 = ops_split(torch.randn(Cin, 3, 3, 32, device=dev) / 17)
    xs = ops.conv3x3_reflect(x0, w0, torch.zeros(Cin, device=dev), False, True, out_split=True)           # a map in the split layout
    w = ops_split(torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** .5)); b = torch.randn(Cout, device=dev)
    check(f"conv split-in {B}x{H}x{H} {Cin}->{Cout} up{up}", lambda: ops.conv3x3_reflect(xs, w, b, bool(up), True, out_split=Cout % 32 == 0 and up == 0).view(torch.int16)
          if (Cout % 32 == 0 and up == 0) else ops.conv3x3_reflect(xs, w, b, bool(up), True))
for M, C, ln in [] if "mlp" not in ONLY else [(32768, 256, 0), (32768, 256, 1), (65536, 256, 1), (262144, 128, 1), (262144, 128, 0)]:
    x = torch.randn(M, C, device=dev)
    w1 = ops_split(torch.randn(4 * C, C, device=dev) / C ** 0.5); w2 = ops_split(torch.randn(C, 4 * C, device=dev) / (4 * C) ** 0.5, layout=1)
    b1 = torch.randn(4 * C, device=dev) * 0.1; b2 = torch.randn(C, device=dev) * 0.1
    lg = torch.ones(C, device=dev) if ln else None; lb = torch.zeros(C, device=dev) if ln else None
    check(f"mlp M={M} C={C} ln{ln}", lambda: ops.mlp(x, w1, b1, w2, b2, res=x, ln_gamma=lg, ln_beta=lb))
if "tail" in ONLY:
  xt = torch.randn(32, 128, 128, 32, device=dev)
  w8 = ops_split(torch.randn(32, 3, 3, 32, device=dev) / 17); w9 = ops_split(torch.randn(3, 3, 3, 32, device=dev) / 17)
  b8 = torch.randn(32, device=dev) * 0.1; b9 = torch.randn(3, device=dev) * 0.1
  check("decoder tail 32x128x128 -> 256x256", lambda: ops.decoder_tail(xt, w8, b8, w9, b9))
if "attn" in ONLY:
  q = torch.randn(32 * 1024, 1024, device=dev); kp = torch.randn(3, 256, device=dev)
  from mastermetastyletransfer_amd.window_plan import window_plan
  plan = window_plan(32, 32, [7, 7], [4, 4], dev)
  b64 = torch.zeros(8, 64, 64, device=dev); b64[:, :, 49:] = -1e30
  check("window attention x3 dual", lambda: torch.cat(ops.window_attn(q[:, :256], q[:, 256:512], q[:, 512:768], q[:, 768:], kp[0], kp[1], kp[2], b64, plan.src, plan.region,
                                                                       32, 1024, 8, plan.nW, plan.N, 32 ** -0.5, x3=FMT)))
print("FAILED" if bad else "all repeatable")
sys.exit(1 if bad else 0)
