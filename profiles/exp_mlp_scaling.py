import sys, os, torch
sys.path.insert(0, os.getcwd())
from mastermetastyletransfer_amd import ops
dev = torch.device("cuda:0")
def t(M, C, HID):
    x = torch.randn(M, C, device=dev).bfloat16(); w1 = (torch.randn(HID, C, device=dev)/16).bfloat16(); w2 = (torch.randn(C, HID, device=dev)/32).bfloat16()
    b1 = torch.randn(HID, device=dev); b2 = torch.randn(C, device=dev)
    for _ in range(3): ops.mlp(x, w1, b1, w2, b2, res=x)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): ops.mlp(x, w1, b1, w2, b2, res=x)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/20*1e3
for M in (128, 1024, 4096, 16384, 32768, 65536):
    for HID in (256, 1024, 4096):
        us = t(M, 256, HID)
        steps = HID // 128 * 4
        print(f"M={M:6d} (WGs {M//128:4d}) HID={HID:5d} steps={steps:4d}: {us:8.1f} us  -> {us/steps:6.2f} us/step")
