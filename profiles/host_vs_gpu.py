"""Host enqueue time vs device time of one forward (is the step launch-bound?).  usage: python profiles/host_vs_gpu.py [bs] [hw]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mastermetastyletransfer_amd import ops
from bench import build_model, synthetic_batch  # noqa

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
model = build_model(dev, "bf16")
c, s = synthetic_batch(bs, hw, dev, 0) if os.environ.get('NORM', '1') == '1' else (torch.rand(bs, 3, hw, hw, device=dev), torch.rand(bs, 3, hw, hw, device=dev))
with torch.no_grad():
    for _ in range(3): model(c, s, 1)
    torch.cuda.synchronize()
    ops.count_reset() if hasattr(ops, "count_reset") else None
    host = []; tot = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model(c, s, 1); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); tot.append(t2 - t0)
    print(f"host enqueue {1e3 * min(host):.3f} ms  (median {1e3 * sorted(host)[5]:.3f})   enqueue+drain {1e3 * min(tot):.3f} ms")
    t0 = time.perf_counter()
    for _ in range(20): model(c, s, 1)
    torch.cuda.synchronize(); print(f"back-to-back {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step")
