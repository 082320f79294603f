"""Pins oracle/imageio_oracle.py: against the committed Pillow outputs (always) and against the installed Pillow (when importable),
and checks ToTensor/Normalize/postprocess against their torch / numpy definitions."""
import os, sys
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from oracle import imageio_oracle as IO
from make_golden_imageio import CASES, seeded_image

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_imageio.npz"))


@pytest.mark.parametrize("i", range(len(CASES)))
def test_resize_matches_committed_pillow_output(i):
    H, W, oh, ow = CASES[i]
    assert np.array_equal(IO.pil_bilinear_resize(seeded_image(i, H, W), oh, ow), GOLD[f"case{i}"])


def test_resize_matches_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for (H, W, oh, ow) in [(240, 320, 256, 256), (300, 200, 256, 256), (64, 48, 256, 256), (256, 256, 256, 256), (256, 500, 256, 256), (33, 400, 17, 31)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        assert np.array_equal(IO.pil_bilinear_resize(img, oh, ow), np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR)))


def test_totensor_normalize_and_postprocess_definitions():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    t = IO.preprocess(img, 40, 56, normalize=True)                          # no resize: ToTensor + Normalize only
    ref = torch.from_numpy(img).permute(2, 0, 1).to(torch.float32).div(255)
    ref = ref.sub(torch.tensor(IO.IMAGENET_MEAN).view(3, 1, 1)).div(torch.tensor(IO.IMAGENET_STD).view(3, 1, 1))
    assert np.array_equal(t, ref.numpy())
    x = rng.standard_normal((2, 3, 8, 9)).astype(np.float32) * 0.7 + 0.5
    assert np.array_equal(IO.postprocess(x), np.clip(x.transpose(0, 2, 3, 1) * 255, 0, 255).astype(np.uint8))
