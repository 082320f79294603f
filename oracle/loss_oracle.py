"""TEST INFRASTRUCTURE - CPU restatement of the reference's perceptual loss forward (codes/loss.py), used by test_model.py:190-203
around every stylization and by the trainers.  Imported only by tests/ and tests/golden/; the product path never touches it.

VGG-19 arithmetic lives in torchvision (absent here, version not pinned by the reference): restated from the published cfg "E"
(conv3x3 zero-pad 1 + ReLU, MaxPool 2x2/2) with the slice indices the reference pins (loss.py:23-37: features[:7] = relu2_1,
[7:12] = relu3_1, [12:21] = relu4_1, [21:30] = relu5_1).  State-dict keys are torchvision's vgg19().features indices
("0.weight", "2.weight", "5.weight", ...), i.e. the keys of the reference's saved cut model (utils.py:10-57).
The LOSS arithmetic (loss.py:102-130,201-336) is PINNED: tests/golden/make_golden_loss.py imports the unmodified reference
loss.py (stand-ins for the absent torchvision imports, seeded VGG weights) and this restatement agreed bit for bit.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import Tensor

# torchvision vgg19 cfg "E" up to relu5_1: index in .features -> (Cin, Cout) for convs, "M" for max-pools
VGG19_CONVS = {0: (3, 64), 2: (64, 64), 5: (64, 128), 7: (128, 128), 10: (128, 256), 12: (256, 256), 14: (256, 256), 16: (256, 256),
               19: (256, 512), 21: (512, 512), 23: (512, 512), 25: (512, 512), 28: (512, 512)}
VGG19_POOLS = (4, 9, 18, 27)
VGG19_TAPS = (6, 11, 20, 29)              # ReLU outputs returned by VGG19_custom.forward (relu2_1, 3_1, 4_1, 5_1)


def vgg19_shapes() -> Dict[str, tuple]:
    out = {}
    for i, (ci, co) in VGG19_CONVS.items():
        out[f"{i}.weight"] = (co, ci, 3, 3); out[f"{i}.bias"] = (co,)
    return out


def vgg19_features(x: Tensor, sd: Dict[str, Tensor]) -> List[Tensor]:
    """loss.py:23-37 on features[:30]"""
    taps = []
    for i in range(30):
        if i in VGG19_CONVS:
            x = F.conv2d(x, sd[f"{i}.weight"], sd[f"{i}.bias"], padding=1)
        elif i in VGG19_POOLS:
            x = F.max_pool2d(x, 2, 2)
        else:
            x = F.relu(x)
            if i in VGG19_TAPS:
                taps.append(x)
    return taps


def _in(x: Tensor) -> Tensor:             # nn.InstanceNorm2d(C): no affine, no running stats, eps 1e-5 (loss.py:102-105)
    return F.instance_norm(x, eps=1e-5)


def content_loss(fc: List[Tensor], fcs: List[Tensor], distance: str = "euclidian") -> Tensor:
    """loss.py:112-118, 272-296"""
    f = torch.abs if distance == "euclidian" else torch.square
    return sum(torch.mean(f(torch.sub(_in(a), _in(b)))) for a, b in zip(fc, fcs))


def style_loss(fs: List[Tensor], fcs: List[Tensor], distance: str = "euclidian") -> Tensor:
    """loss.py:121-130, 300-322 (torch.std: unbiased)"""
    f = torch.abs if distance == "euclidian" else torch.square
    return sum(torch.mean(f(torch.sub(a.mean([2, 3]), b.mean([2, 3])))) + torch.mean(f(torch.sub(a.std([2, 3]), b.std([2, 3]))))
               for a, b in zip(fs, fcs))


def overall_loss(content: Tensor, style: Tensor, output: Tensor, sd: Dict[str, Tensor], lambda_value: float = 10.0,
                 distance_content: str = "euclidian", distance_style: str = "euclidian"):
    """get_overall_loss (loss.py:201-262): returns (total, content, style)"""
    fc, fs, fo = vgg19_features(content, sd), vgg19_features(style, sd), vgg19_features(output, sd)
    c = content_loss(fc, fo, distance_content)
    s = style_loss(fs, fo, distance_style)
    return c + lambda_value * s, c, s
