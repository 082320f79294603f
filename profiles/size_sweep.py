"""No-crash / parity sweep over odd image sizes in both compute modes (run on the GPU box): python profiles/size_sweep.py"""
import sys, torch, warnings
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from bench import build_model
import oracle.master_oracle as O
dev = torch.device("cuda:0")
for dt in ("bf16", "fp32"):
    m = build_model(dev, dt)
    sd = [ {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in mod.state_dict().items()} for mod in (m.swin_encoder, m.style_transformer, m.decoder)]
    for (B, H, W) in [(1, 1024, 1024), (1, 384, 640), (3, 264, 200), (1, 16, 16), (2, 56, 56)]:
        g = torch.Generator().manual_seed(H + W)
        c = torch.rand(B, 3, H, W, generator=g); s = torch.rand(B, 3, H, W, generator=g)
        with torch.no_grad():
            out = m(c.to(dev), s.to(dev), 1).cpu()
        msg = f"{dt} {B}x{H}x{W}: finite={bool(torch.isfinite(out).all())} max|out|={out.abs().max():.3f}"
        if H * W <= 384 * 640:
            with torch.no_grad(): ref = O.full_forward(c, s, 1, *sd)
            msg += f"  max|d| vs oracle = {(out - ref).abs().max():.3e} (ref max {ref.abs().max():.3f})"
        print(msg, flush=True)
