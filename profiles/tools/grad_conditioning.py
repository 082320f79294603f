"""Is a gradient difference between the HIP train step and the fp32 CPU oracle a kernel error or the conditioning of the problem?
Computes the same gradients three ways - float64 oracle (CPU), float32 oracle (CPU), HIP fp32 - and prints each fp32 result's error
against float64.  Run on the GPU box:  python profiles/tools/grad_conditioning.py"""
import os, sys, warnings
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import loss_oracle as L, master_oracle as O, train_oracle as T
from oracle.seeded import seeded_input, seeded_state_dict
import make_golden_train as M

dev = torch.device("cuda:0")


def relmax(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def rell2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def perceptual(B, H, W, dc, ds):
    from mastermetastyletransfer_amd import custom_loss
    sd = seeded_state_dict(L.vgg19_shapes(), 51)
    c, s = (seeded_input(f"plb_{n}", (B, 3, H, W), 52 + i, kind="image") for i, n in enumerate("cs"))
    o = seeded_input("plb_o", (B, 3, H, W), 55, kind="image")
    res = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        oo = o.detach().clone().to(dt).requires_grad_(True)
        t, _, _ = L.overall_loss(c.to(dt), s.to(dt), oo, {k: v.to(dt) for k, v in sd.items()}, 7.0, dc, ds)
        t.backward(); res[name] = oo.grad
    loss = custom_loss(default_lambda_value=7.0, distance_content=dc, distance_style=ds, vgg_state_dict=sd).to(dev)
    od = o.detach().clone().to(dev).requires_grad_(True)
    loss(c.to(dev), s.to(dev), od).backward()
    res["hip"] = od.grad.cpu()
    print(f"perceptual {B}x{H}x{W} {dc}/{ds}: vs f64  cpu-f32 max {relmax(res['f32'], res['f64']):.2e} l2 {rell2(res['f32'], res['f64']):.2e} | "
          f"hip max {relmax(res['hip'], res['f64']):.2e} l2 {rell2(res['hip'], res['f64']):.2e} | hip vs cpu-f32 max {relmax(res['hip'], res['f32']):.2e}")


def golden(i):
    from mastermetastyletransfer_amd import MasterStyleTransferModel, custom_loss, train_forward as TF
    tag, B, H, W, k, dc, ds, lam, only = M.CASES[i]
    st_sd, dec_sd, vgg_sd, fc, fs, content, style = M.case_inputs(i, B, H, W)
    masks = M.case_masks(i, B, k)
    grads = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        cv = lambda d: {n: (t.to(dt).requires_grad_(True) if t.is_floating_point() else t) for n, t in d.items()}
        st, dec = cv(st_sd), cv(dec_sd)
        total, *_ = T.train_forward_loss(fc.to(dt), fs.to(dt), content.to(dt), style.to(dt), k, st, dec, {n: t.to(dt) for n, t in vgg_sd.items()},
                                         [m.to(dt) for m in masks], lam, dc, ds)
        total.backward()
        grads[name] = {"style_transformer." + n: t.grad for n, t in st.items() if t.is_floating_point()}
        grads[name].update({"decoder." + n: t.grad for n, t in dec.items()})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True)
    m.style_transformer.load_state_dict(st_sd); m.decoder.load_state_dict(dec_sd)
    m = m.to(dev).train()
    loss_fn = custom_loss(default_lambda_value=lam, distance_content=dc, distance_style=ds, vgg_state_dict=vgg_sd).to(dev)
    TF.sd_masks.set_masks(masks)
    decoded = m.decoder(m.style_transformer(fc.to(dev), fs.to(dev), k).permute(0, 3, 1, 2))
    TF.sd_masks.set_masks(None)
    loss_fn(content.to(dev), style.to(dev), decoded).backward()
    hip = {"style_transformer." + n: p.grad.cpu() for n, p in m.style_transformer.named_parameters()}
    hip.update({"decoder." + n: p.grad.cpu() for n, p in m.decoder.named_parameters()})
    rows = []
    for n in hip:
        if n in M.ZERO_GRAD:
            continue
        rows.append((relmax(hip[n], grads["f64"][n]), relmax(grads["f32"][n], grads["f64"][n]), n))
    rows.sort(reverse=True)
    print(f"golden {tag}: worst 6 tensors  (hip vs f64 | cpu-f32 vs f64)")
    for h, c, n in rows[:6]:
        print(f"   {h:.2e} | {c:.2e}  {n}")
    print(f"   median hip {sorted(r[0] for r in rows)[len(rows)//2]:.2e}  median cpu-f32 {sorted(r[1] for r in rows)[len(rows)//2]:.2e}")


if __name__ == "__main__":
    perceptual(2, 64, 64, "euclidian", "euclidian")
    perceptual(1, 48, 80, "euclidian_squared", "euclidian_squared")
    perceptual(2, 32, 32, "euclidian", "euclidian_squared")
    perceptual(2, 128, 128, "euclidian_squared", "euclidian_squared")
    golden(0); golden(1)
