"""End-to-end serving rate including PCIe: uint8 content/style images in pinned host memory -> H2D -> device preprocess
(Pillow-exact resize + ToTensor + Normalize) -> forward -> device postprocess -> uint8 D2H.  Double-buffered over two streams.
    python profiles/bench_e2e.py [src_h src_w]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from mastermetastyletransfer_amd import Preprocess, postprocess
dev = torch.device("cuda:0")
SH, SW = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
B, steps = 32, 30
m = build_model(dev, "bf16")
pre = Preprocess((256, 256), imagenet_normalization=True)
host_c = [torch.randint(0, 256, (B, SH, SW, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
host_s = [torch.randint(0, 256, (B, SH, SW, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
host_o = [torch.empty((B, 256, 256, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def step(i):
    j = i & 1
    with torch.cuda.stream(streams[j]), torch.no_grad():
        c = host_c[j].to(dev, non_blocking=True); s = host_s[j].to(dev, non_blocking=True)
        out = postprocess(m(pre(c), pre(s), 1))
        host_o[j].copy_(out, non_blocking=True)

for i in range(4): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps): step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# device-only pieces
c = host_c[0].to(dev); x = pre(c)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print(f"source {SH}x{SW} uint8, bs={B}: end-to-end {dt*1e3:.3f} ms/step = {B/dt:.0f} images/s  (host uint8 in -> host uint8 out, 2 streams)")
print(f"  preprocess (one batch of {B}): {timeit(lambda: pre(c)):.3f} ms   postprocess: {timeit(lambda: postprocess(x)):.3f} ms   "
      f"H2D bytes/step {2*B*SH*SW*3/1e6:.1f} MB, D2H {B*256*256*3/1e6:.1f} MB")
