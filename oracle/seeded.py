"""Deterministic, torch-RNG-independent weights and inputs for parity tests (TEST INFRASTRUCTURE).

Golden vectors under ``tests/golden`` store only inputs' seeds and expected outputs; the weights are
re-created on any machine from (key name, shape, seed) with numpy's PCG64 stream, so the reference
never has to travel and fixtures stay small.  Scales are chosen so that activations stay O(1)
through ~40 chained layers and so that biases (which become the q/k/v of zero-padded window slots,
reference style_transformer.py:83-85,114-116) and relative-position tables actually matter.
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np
import torch

from .master_oracle import relative_position_index


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def seeded_tensor(key: str, shape: Tuple[int, ...], seed: int, ws=(7, 7)) -> torch.Tensor:
    g = _rng(seed, key)
    if key.endswith("relative_position_index"):
        return relative_position_index(ws)
    if key.endswith("relative_position_bias_table"):
        a = g.standard_normal(shape) * 0.5
    elif key.endswith("bias"):
        a = g.standard_normal(shape) * 0.1
    elif len(shape) == 1:                       # norm weight
        a = 1.0 + 0.1 * g.standard_normal(shape)
    elif len(shape) == 2:                       # Linear [out, in]
        a = g.standard_normal(shape) / np.sqrt(shape[1])
    elif len(shape) == 4:                       # Conv2d [out, in, kh, kw]
        a = g.standard_normal(shape) * np.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))
    else:
        raise ValueError(f"no rule for {key} {shape}")
    return torch.from_numpy(a.astype(np.float32))


def seeded_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int, ws=(7, 7)) -> Dict[str, torch.Tensor]:
    return {k: seeded_tensor(k, s, seed, ws) for k, s in shapes.items()}


def seeded_input(name: str, shape: Tuple[int, ...], seed: int, kind: str = "normal") -> torch.Tensor:
    g = _rng(seed, "input:" + name)
    if kind == "normal":
        a = g.standard_normal(shape)
    elif kind == "image":                       # ToTensor range then ImageNet normalisation (test_model.py:40-48)
        a = g.random(shape)
        mean = np.array([0.485, 0.456, 0.406]).reshape(1, 3, 1, 1)
        std = np.array([0.229, 0.224, 0.225]).reshape(1, 3, 1, 1)
        a = (a - mean) / std
    else:
        raise ValueError(kind)
    return torch.from_numpy(a.astype(np.float32))
