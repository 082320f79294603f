"""Perceptual-loss forward (VGG-19 features of content, style, output + content/style terms) at bs=32, 256x256.
    python profiles/bench_loss.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mastermetastyletransfer_amd import custom_loss
from mastermetastyletransfer_amd.loss import VGG19_CONVS
dev = torch.device("cuda:0")
B, S = 32, 256
torch.manual_seed(0)
sd = {}
for i, (ci, co) in VGG19_CONVS.items():
    sd[f"{i}.weight"] = torch.randn(co, ci, 3, 3) * (2.0 / (9 * ci)) ** 0.5; sd[f"{i}.bias"] = torch.randn(co) * 0.01
c, s, o = (torch.rand(B, 3, S, S, device=dev) for _ in range(3))
flops = 0.0; h = S
for i in range(30):
    if i in VGG19_CONVS: flops += 2.0 * h * h * 9 * VGG19_CONVS[i][0] * VGG19_CONVS[i][1]
    elif i in (4, 9, 18, 27): h //= 2
for dt in (torch.bfloat16, torch.float32):
    crit = custom_loss(vgg_state_dict=sd, compute_dtype=dt).to(dev).eval()
    with torch.no_grad():
        for _ in range(3): out = crit(c, s, o, output_content_and_style_loss=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): out = crit(c, s, o, output_content_and_style_loss=True)
        torch.cuda.synchronize(); dt_ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{str(dt):15s} bs={B} {S}x{S}: {dt_ms:7.2f} ms per loss evaluation ({3 * B} images through VGG-19[:30], {3 * B * flops / 1e12:.2f} TFLOP) "
          f"= {3 * B * flops / dt_ms / 1e9:.0f} TFLOP/s;  total {float(out[0]):.4f} content {float(out[1]):.4f} style {float(out[2]):.4f}")
