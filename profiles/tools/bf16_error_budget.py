"""Where does the bf16 mode's error come from?  Runs the path stage by stage (Swin encoder | StyleEncoder | StyleDecoder | image decoder)
in the fp32 parity mode and in bf16, and for every stage feeds the bf16 kernels with the fp32-mode input of that stage, so the error
each stage ADDS is separated from the error it inherits.  Run on the GPU box: python profiles/tools/bf16_error_budget.py"""
import os, sys, warnings
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import master_oracle as O
from oracle.seeded import seeded_input, seeded_state_dict
from mastermetastyletransfer_amd import MasterStyleTransferModel, ops

dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True)
sds = (seeded_state_dict(O.swin_shapes(), 31), seeded_state_dict(O.style_transformer_shapes(), 32), seeded_state_dict(O.decoder_shapes(), 33))
m.swin_encoder.load_state_dict(sds[0]); m.style_transformer.load_state_dict(sds[1]); m.decoder.load_state_dict(sds[2])
m = m.to(dev).eval()
B, S = 2, 256
c = seeded_input("c256", (B, 3, S, S), 8, kind="image").to(dev); s = seeded_input("s256", (B, 3, S, S), 9, kind="image").to(dev)
BF, F32 = torch.bfloat16, torch.float32


def stages(dtype, inject=None):
    """returns dict of stage outputs (fp32 copies); inject: dict stage -> fp32 tensor to use as that stage's INPUT instead"""
    m.set_compute_dtype(dtype)
    out = {}
    cv = lambda t: ops.convert(t.contiguous(), dtype)
    with torch.no_grad():
        x, _, h, w = m.swin_encoder.run(c, s)
        out["swin"] = x.float()
        if inject and "enc" in inject: x = cv(inject["enc"])
        fc, fs = x[: B * h * w], x[B * h * w:]
        key, scale, shift = m.style_transformer.encoder.run(fs, fs, fs, B, h, w)
        out["enc_key"], out["enc_scale"], out["enc_shift"] = key.float(), scale.float(), shift.float()
        if inject and "dec" in inject:
            fc, key, scale, shift = (cv(t) for t in inject["dec"])
        fcs = m.style_transformer.decoder.run(fc, key, scale, shift, B, h, w)
        out["st"] = fcs.float()
        if inject and "img" in inject: fcs = cv(inject["img"])
        out["image"] = m.decoder.run_nhwc(fcs.view(B, h, w, -1)).float()
    return out


rel = lambda a, b: float((a - b).norm() / b.norm())
ref = stages(F32)
bf = stages(BF)
print("cumulative relative L2 error of the bf16 mode vs the fp32 mode at stage boundaries:")
for k in ("swin", "enc_key", "enc_scale", "enc_shift", "st", "image"):
    print(f"   {k:10s} {rel(bf[k], ref[k]):.3e}")
print("error ADDED by one stage when its input is the fp32-mode tensor (rounded to bf16):")
e = stages(BF, {"enc": ref["swin"]})
print(f"   StyleEncoder : key {rel(e['enc_key'], ref['enc_key']):.3e} scale {rel(e['enc_scale'], ref['enc_scale']):.3e} shift {rel(e['enc_shift'], ref['enc_shift']):.3e}")
x = ref["swin"]; n = x.shape[0] // 2
d = stages(BF, {"dec": (x[:n], ref["enc_key"], ref["enc_scale"], ref["enc_shift"])})
print(f"   StyleDecoder : {rel(d['st'], ref['st']):.3e}")
i = stages(BF, {"img": ref["st"]})
print(f"   image decoder: {rel(i['image'], ref['image']):.3e}")
print(f"   (input rounding alone: swin {rel(ref['swin'].to(BF).float(), ref['swin']):.3e})")
