import sys, torch
sys.path.insert(0, "/root/repo")
from mastermetastyletransfer_amd import ops
dev = torch.device("cuda:0")
for E in (128, 96):
    for B, S in ((32, 256), (16, 512), (1, 200)):
        img = torch.randn(B, 3, S - S % 4, S - S % 4, device=dev); w = torch.randn(E, 48, device=dev) / 7; b = torch.randn(E, device=dev)
        g = torch.rand(E, device=dev) + 0.5; bt = torch.randn(E, device=dev)
        out = ops.patch_embed(img, w, b, g, bt, torch.float32)
        ref = torch.nn.functional.layer_norm(torch.nn.functional.conv2d(img.double(), w.double().view(E, 3, 4, 4), b.double(), stride=4).permute(0, 2, 3, 1), (E,), g.double(), bt.double(), 1e-5)
        err = float((out.double() - ref).abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): ops.patch_embed(img, w, b, g, bt, torch.float32, out=out)
        e0.record()
        for _ in range(20): ops.patch_embed(img, w, b, g, bt, torch.float32, out=out)
        e1.record(); torch.cuda.synchronize()
        print(f"E={E} {B}x{S}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us, max err {err:.2e}")
