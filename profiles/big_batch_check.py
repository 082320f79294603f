"""Large-batch sanity (index arithmetic): the first 4 samples of a batch of 160 must equal the same samples run as a batch of 4,
bit for bit, in both modes.  python profiles/big_batch_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, synthetic_batch
dev = torch.device("cuda:0")
for dt in ("bf16", "fp32"):
    m = build_model(dev, dt)
    c, s = synthetic_batch(160, 256, dev, 0)
    with torch.no_grad():
        big = m(c, s, 1); small = m(c[:4].contiguous(), s[:4].contiguous(), 1); tail = m(c[156:].contiguous(), s[156:].contiguous(), 1)
    print(dt, "first4 bitwise:", bool((big[:4] == small).all()), " last4 bitwise:", bool((big[156:] == tail).all()), " finite:", bool(torch.isfinite(big).all()))
