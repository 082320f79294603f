"""Golden vectors for the pre-processing resize: outputs of the Pillow installed in the build container
(PIL.Image.resize(size, BILINEAR) = what torchvision's transforms.Resize does to a PIL image, reference test_model.py:39-44)
on small seeded uint8 images.  Inputs are regenerated from the seed; only sizes and Pillow's outputs are stored.
    python tests/golden/make_golden_imageio.py  ->  tests/golden/golden_imageio.npz"""
import os
import numpy as np

CASES = [(48, 64, 32, 32), (30, 20, 32, 32), (20, 24, 32, 48), (100, 77, 24, 24), (32, 50, 32, 16), (37, 32, 16, 32)]   # H, W, out_h, out_w


def seeded_image(i, H, W):
    return np.random.default_rng(1000 + i).integers(0, 256, (H, W, 3), dtype=np.uint8)


if __name__ == "__main__":
    import PIL                      # only the generator needs Pillow; the tests import CASES / seeded_image from this file
    from PIL import Image
    out = {"pillow_version": np.array(PIL.__version__)}
    for i, (H, W, oh, ow) in enumerate(CASES):
        out[f"case{i}"] = np.asarray(Image.fromarray(seeded_image(i, H, W)).resize((ow, oh), Image.BILINEAR))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_imageio.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")
