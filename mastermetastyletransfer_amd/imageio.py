"""Device-side pre/post-processing of the reference's test harness (test_model.py:39-48,110-125,:207), so that a serving
loop hands uint8 images to the GPU (4x smaller than fp32 over PCIe) and gets uint8 images back.

    pre = Preprocess((256, 256), imagenet_normalization=True)
    content = pre(content_u8)            # uint8 [B,H,W,3] RGB on the GPU -> fp32 [B,3,256,256]
    out_u8 = postprocess(model(content, style))

Preprocess reproduces transforms.Compose([ToPILImage(), Resize(size), ToTensor()]) + transforms.Normalize bit for bit
(Pillow's 8-bit fixed-point antialiased bilinear resample); postprocess is np.clip(x * 255, 0, 255).astype(np.uint8).
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from . import ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # test_model.py:48
IMAGENET_STD = (0.229, 0.224, 0.225)


class Preprocess:
    def __init__(self, size: Tuple[int, int] = (256, 256), imagenet_normalization: bool = True,
                 mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD):
        self.size = (int(size[0]), int(size[1]))
        self.mean, self.std = (tuple(mean), tuple(std)) if imagenet_normalization else (None, None)

    def __call__(self, images_u8_hwc: torch.Tensor) -> torch.Tensor:
        x = images_u8_hwc
        if x.dim() == 3:
            x = x.unsqueeze(0)
        return ops.preprocess(x.contiguous(), self.size[0], self.size[1], self.mean, self.std)


def postprocess(stylized: torch.Tensor) -> torch.Tensor:
    """fp32 NCHW -> uint8 NHWC (RGB order as produced; the reference's cv2.imwrite then stores it as BGR)"""
    return ops.postprocess(stylized)
