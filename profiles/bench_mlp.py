import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mastermetastyletransfer_amd import ops
dev = torch.device("cuda:0")
shapes = [(262144,128,512)] if "c128" in sys.argv else [(32768,256,1024),(65536,256,1024),(262144,128,512)]
print("MASTER_MLP_FORM =", os.environ.get("MASTER_MLP_FORM"))
for M, C, HID in shapes:
    x = torch.randn(M, C, device=dev).bfloat16(); w1 = (torch.randn(HID, C, device=dev)/16).bfloat16(); w2 = (torch.randn(C, HID, device=dev)/32).bfloat16()
    b1 = torch.randn(HID, device=dev); b2 = torch.randn(C, device=dev); g = torch.ones(C, device=dev)
    for ln in (False, True):
        kw = dict(ln_gamma=g, ln_beta=g) if ln else {}
        for _ in range(3): ops.mlp(x, w1, b1, w2, b2, res=x, **kw)
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(20): ops.mlp(x, w1, b1, w2, b2, res=x, **kw)
        e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1)/20*1e3
        print(f"fused mlp M={M} C={C} HID={HID} ln={ln}: {us:.1f} us  {4.0*M*C*HID/us/1e6:.1f} TFLOP/s")
