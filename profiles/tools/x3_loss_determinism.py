"""Bitwise repeatability of the split-bf16 loss network (VGG-19 forward on 3B images, data-gradient chain on B) at small and full map sizes.
usage: python profiles/tools/x3_loss_determinism.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mastermetastyletransfer_amd import ops, custom_loss
from mastermetastyletransfer_amd.ops import _check, _p, _stream, load_library
from oracle import loss_oracle as L
from oracle.seeded import seeded_input, seeded_state_dict

dev = torch.device("cuda:0")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
bad = 0


def check(name, fn):
    global bad
    outs = []
    for i in range(REPS):
        junk = torch.full((1 << 22,), float("nan") if i & 1 else 1e30, device=dev); del junk      # recycled allocations come back holding garbage
        outs.append(fn())
    torch.cuda.synchronize()
    n = sum(0 if torch.equal(outs[0], o) else 1 for o in outs[1:])
    if n:
        d = max(float((outs[0] - o).abs().max()) for o in outs[1:]); cnt = max(int((outs[0] != o).sum()) for o in outs[1:])
        bad += 1
        print(f"NONDETERMINISTIC {name}: {n} of {REPS - 1} repeats differ, max |d| {d:.3e}, up to {cnt} of {outs[0].numel()} elements", flush=True)
    else:
        print(f"ok {name}", flush=True)


lib = load_library()
for B, H, Cin, Cout in [(6, 64, 64, 64), (6, 32, 64, 128), (6, 32, 128, 128), (6, 16, 128, 256), (6, 16, 256, 256), (6, 8, 256, 512), (6, 8, 512, 512), (6, 4, 512, 512),
                        (24, 256, 64, 64), (24, 128, 128, 128), (24, 16, 512, 512)]:
    x = torch.randn(B, H, H, Cin, device=dev); w = ops.split_weights(torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** .5)); b = torch.randn(Cout, device=dev)
    check(f"conv zero-pad {B}x{H}x{H} {Cin}->{Cout}", lambda: ops.conv3x3_zero(x, w, b, relu=True))
for B, H, Cg, Cx, mask, add in [(2, 4, 512, 512, 0, 0), (2, 8, 512, 512, 1, 0), (2, 8, 512, 256, 0, 0), (2, 16, 256, 256, 1, 1), (2, 16, 256, 128, 0, 0), (2, 32, 128, 128, 1, 1),
                                (2, 32, 128, 64, 0, 0), (2, 64, 64, 64, 1, 0), (2, 64, 64, 32, 0, 0), (8, 256, 64, 32, 0, 0), (8, 128, 128, 128, 1, 1)]:
    g = torch.randn(B, H, H, Cg, device=dev); wf = ops.split_weights(torch.randn(Cx, 3, 3, Cg, device=dev) / (3 * Cg ** .5))
    rs = torch.randn(B, H, H, Cx, device=dev) if mask else None; ad = torch.randn(B, H, H, Cx, device=dev) if add else None

    def run():
        dx = torch.empty((B, H, H, Cx), device=dev)
        _check(lib.master_conv3x3_dgrad(_p(g), _p(wf), _p(dx), _p(rs), _p(ad), B, H, H, Cg, Cx, 0, ops.F32X3, _stream()), "dgrad")
        return dx
    check(f"dgrad {B}x{H}x{H} {Cg}->{Cx} mask{mask} add{add}", run)
sd = seeded_state_dict(L.vgg19_shapes(), 51)
for mode in (torch.float32, "bf16x3"):
    for B, S in [(2, 64), (2, 128), (8, 256)]:
        c, s, o = (seeded_input(f"plb_{n}", (B, 3, S, S), 52 + i, kind="image").to(dev) for i, n in enumerate("cso"))
        loss = custom_loss(default_lambda_value=7.0, vgg_state_dict=sd).to(dev).set_compute_dtype(mode)

        def run():
            od = o.clone().requires_grad_(True)
            t = loss(c, s, od)
            t.backward()
            return torch.cat([od.grad.flatten(), t.detach().reshape(1)])
        check(f"loss forward + backward {mode} {B}x{S}x{S}", run)
print("FAILED" if bad else "all repeatable")
sys.exit(1 if bad else 0)
