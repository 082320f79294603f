"""Pins oracle/loss_oracle.py against the outputs of the imported reference codes/loss.py (tests/golden/golden_loss.npz)."""
import os, sys
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from oracle import loss_oracle as L
from make_golden_loss import CASES, seeded_vgg_and_images

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_loss.npz"))


@pytest.mark.parametrize("i", range(len(CASES)))
def test_loss_oracle_matches_reference_outputs(i):
    dc, ds, B, H, W = CASES[i]
    sd, (c, s, o) = seeded_vgg_and_images(i, B, H, W)
    with torch.no_grad():
        t, cl, sl = L.overall_loss(c, s, o, sd, 10.0, dc, ds)
        relu5 = L.vgg19_features(o, sd)[3]
    assert np.allclose([float(t), float(cl), float(sl)], GOLD[f"case{i}"], rtol=1e-6, atol=0)
    assert np.allclose(relu5.numpy(), GOLD[f"case{i}_relu5_1"], rtol=1e-5, atol=1e-6)
