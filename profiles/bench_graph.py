"""Eager vs hipGraph replay of the forward at serving batch sizes.  python profiles/bench_graph.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, synthetic_batch
from mastermetastyletransfer_amd import GraphedStylizer
dev = torch.device("cuda:0")
m = build_model(dev, "bf16")
def timeit(fn, n):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B in (1, 2, 4, 8, 32):
    c, s = synthetic_batch(B, 256, dev, 0)
    g = GraphedStylizer(m, B, 256, 256, 1)
    with torch.no_grad():
        e = timeit(lambda: m(c, s, 1), 50)
    r = timeit(lambda: g(c, s), 50)
    print(f"bs={B:3d} 256x256 bf16: eager {e:7.3f} ms ({B / e * 1e3:8.0f} img/s)   graph replay {r:7.3f} ms ({B / r * 1e3:8.0f} img/s)")
