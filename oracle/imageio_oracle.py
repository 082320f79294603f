"""TEST INFRASTRUCTURE - CPU restatement of the reference's pre/post-processing around the forward.
Imported only by tests/ (and profiles/ scripts); the product path never touches it.

Pre  (reference test_model.py:39-46,48,110-125): uint8 HWC RGB image -> transforms.ToPILImage -> transforms.Resize((256,256))
      -> transforms.ToTensor -> optional transforms.Normalize(ImageNet mean/std).
      torchvision's Resize on a PIL image is PIL.Image.resize(size, BILINEAR): Pillow's two-pass (horizontal, then vertical)
      antialiased resample in 8-bit fixed point.  Pillow is a third-party dependency of the reference (not vendored, version not
      pinned by it); restated here from Pillow's published algorithm (src/libImaging/Resample.c: precompute_coeffs,
      normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc, bilinear_filter) and PINNED bit-for-bit against the
      Pillow installed in this image (12.2.0) by tests/test_imageio_cpu.py and the committed fixture tests/golden/golden_imageio.npz.
Post (reference test_model.py:207): np.clip(stylized.squeeze(0).permute(1,2,0).cpu().numpy() * 255, 0, 255).astype(np.uint8)
      (truncation, not rounding; the RGB-through-cv2.imwrite channel swap happens in the file writer, outside this function).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def resample_coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the whole axis.
    Returns bounds int32 [out,2] = (first source index, tap count) and coefficients int32 [out, ksize] (fixed point, 22 bits)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0.0 else a
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(ksize):                      # (int)(+-0.5 + k * 2^22): C truncation toward zero
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """one pass along axis 0 of a uint8 array [N, ...]: sum of taps in int32, + 2^21, >> 22, clip to [0,255]"""
    bounds, kk = resample_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[x0 + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_bilinear_resize(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [H,W,C] -> uint8 [out_h,out_w,C]; Pillow order: horizontal pass (skipped when the width is unchanged), then vertical."""
    assert img.dtype == np.uint8 and img.ndim == 3
    x = img
    if out_w != img.shape[1]:
        x = np.ascontiguousarray(_resample_axis0(np.ascontiguousarray(x.transpose(1, 0, 2)), out_w).transpose(1, 0, 2))
    if out_h != img.shape[0]:
        x = _resample_axis0(x, out_h)
    return x


def preprocess(img: np.ndarray, out_h: int = 256, out_w: int = 256, normalize: bool = True) -> np.ndarray:
    """uint8 HWC RGB -> float32 CHW: resize, ToTensor (/255 in fp32), optional ImageNet Normalize ((x - mean) / std in fp32)"""
    r = pil_bilinear_resize(img, out_h, out_w)
    t = r.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    if normalize:
        mean = np.asarray(IMAGENET_MEAN, np.float32).reshape(3, 1, 1)
        std = np.asarray(IMAGENET_STD, np.float32).reshape(3, 1, 1)
        t = (t - mean) / std
    return t


def postprocess(stylized: np.ndarray) -> np.ndarray:
    """float32 [B,3,H,W] -> uint8 [B,H,W,3]: np.clip(x * 255, 0, 255).astype(uint8)   (test_model.py:207)"""
    x = np.asarray(stylized, np.float32).transpose(0, 2, 3, 1)
    return np.clip(x * np.float32(255.0), 0, 255).astype(np.uint8)
