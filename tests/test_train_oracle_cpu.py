"""Pins oracle/train_oracle.py against the train step of the imported reference (tests/golden/golden_train.npz, written by
tests/golden/make_golden_train.py): loss, every parameter gradient (norm, sum, first values) and the Adam update."""
import os, sys
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from oracle import train_oracle as T
import make_golden_train as M

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_train.npz"))


def oracle_case(i, dtype=torch.float32):
    """(losses, grads, output) of golden case i from the oracle; shared with the GPU parity test (dtype=float64: the reference
    gradients the two fp32 implementations are measured against)"""
    tag, B, H, W, k, dc, ds, lam, only = M.CASES[i]
    st_sd, dec_sd, vgg_sd, fc, fs, content, style = M.case_inputs(i, B, H, W)
    masks = [m.to(dtype) for m in M.case_masks(i, B, k)]
    fc, fs, content, style = (t.to(dtype) for t in (fc, fs, content, style))
    vgg_sd = {n: t.to(dtype) for n, t in vgg_sd.items()}

    def leaf(prefix, sd):
        return {n: (t.to(dtype).clone().requires_grad_(True) if t.is_floating_point() and (only is None or (prefix + n).startswith(only))
                    else (t.to(dtype) if t.is_floating_point() else t.clone())) for n, t in sd.items()}
    st, dec = leaf("style_transformer.", st_sd), leaf("decoder.", dec_sd)
    total, c, s, out = T.train_forward_loss(fc, fs, content, style, k, st, dec, vgg_sd, masks, lam, dc, ds)
    total.backward()
    grads = {"style_transformer." + n: t.grad for n, t in st.items() if t.requires_grad}
    grads.update({"decoder." + n: t.grad for n, t in dec.items() if t.requires_grad})
    return (float(total.detach()), float(c.detach()), float(s.detach())), grads, out.detach()


@pytest.mark.parametrize("i", range(len(M.CASES)))
def test_train_oracle_matches_reference_step(i):
    tag = M.CASES[i][0]
    losses, grads, _ = oracle_case(i)
    assert np.allclose(losses, GOLD[f"{tag}_loss"], rtol=1e-5, atol=0)
    names = [str(n) for n in GOLD[f"{tag}_names"]]
    assert sorted(grads) == names
    gold = GOLD[f"{tag}_grads"]
    gscale = gold[:, 0].max()
    for n, row in zip(names, gold):
        got = M.grad_summary(grads[n])
        if n in M.ZERO_GRAD:                     # zero in exact arithmetic: only rounding noise on either side
            assert got[0] <= 1e-6 * gscale and row[0] <= 1e-6 * gscale, n
            continue
        assert abs(got[0] - row[0]) <= 1e-4 * row[0], (n, got[0], row[0])                      # L2 norm
        assert np.abs(got[2:] - row[2:]).max() <= 1e-4 * max(np.abs(row[2:]).max(), 1e-3 * row[0]), n   # leading values


def test_adam_matches_torch_optim():
    g = torch.Generator().manual_seed(0)
    p0 = {"a": torch.randn(300, generator=g), "b": torch.randn(7, 5, generator=g)}
    params = {k: v.clone() for k, v in p0.items()}
    tp = [torch.nn.Parameter(v.clone()) for v in p0.values()]
    opt = torch.optim.Adam(tp, lr=3e-3)
    state = {}
    for step in range(1, 5):
        grads = {k: torch.randn(v.shape, generator=g) * 10 ** float(torch.randint(-6, 2, (1,), generator=g)) for k, v in p0.items()}
        for t, gr in zip(tp, grads.values()):
            t.grad = gr.clone()
        opt.step()
        T.adam_step(params, grads, state, 3e-3, step)
    for t, v in zip(tp, params.values()):
        assert torch.allclose(t.detach(), v, rtol=1e-6, atol=1e-8)
