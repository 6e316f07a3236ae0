"""Deterministic synthetic weights.

No trained checkpoint is reachable offline (``model.pt`` is a download, README.en.md:30-33, and
``efficientnetv2-xl-21k.npz`` is absent so ``load_weight`` only prints ``not found``,
``/root/reference/models/detector.py:34-36``).  Parity tests and ``bench.py`` therefore use weights
regenerated from a seed on both sides: a counter-based generator (numpy Philox) keyed by the
``state_dict`` key, so any machine rebuilds bit-identical tensors without the reference and without
shipping ~1 GB of floats.

The distributions are chosen so that the BN-fold path is genuinely exercised (non-trivial gamma,
beta, running stats), activations stay O(1-10) through 100 residual blocks and the nine FPN heads,
and the key heat-map yields a realistic number of peaks.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

from .schema import text_detector_schema, detector_schema


def _rng(seed: int, name: str) -> np.random.Generator:
    key = (zlib.crc32(name.encode()) << 32) | (zlib.adler32(name.encode()) & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFFFFFFFFFF, key]))


def _is_residual_tail_bn(name: str) -> bool:
    # last BN of a backbone block (its output is added to the block input)
    parts = name.split(".")
    if "backbone" not in parts or "block" not in parts:
        return False
    i = parts.index("block")
    return parts[i + 1] in ("1", "3", "0") and parts[i + 2] == "1" and _is_last_layer(parts, i)


def _is_last_layer(parts, i) -> bool:
    # block.3.1 (MBConv project BN), block.1.1 of fused-e4 is project BN, block.0.1 of fused-e1
    stage = int(parts[parts.index("features") + 1])
    layer = parts[i + 1]
    if stage == 1:
        return layer == "0"
    if stage in (2, 3):
        return layer == "1"
    return layer == "3"


def fill_tensor(seed: int, name: str, shape: Tuple[int, ...], kind: str) -> torch.Tensor:
    g = _rng(seed, name)
    if kind in ("conv", "conv_proj", "conv_dw", "conv_top", "se_w1", "se_w2", "linear"):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        gain = {"conv": 1.5, "conv_proj": 1.0, "conv_dw": 1.6, "conv_top": 6.0,
                "se_w1": 1.0, "se_w2": 1.0, "linear": 1.0}[kind]
        a = g.standard_normal(shape, dtype=np.float32) * np.float32(gain / np.sqrt(fan_in))
    elif kind == "bn_weight":
        if _is_residual_tail_bn(name):
            a = g.uniform(0.15, 0.35, shape).astype(np.float32)
        else:
            a = g.uniform(0.7, 1.3, shape).astype(np.float32)
    elif kind in ("bn_bias", "bn_mean"):
        a = (g.standard_normal(shape, dtype=np.float32) * np.float32(0.15))
    elif kind == "bn_var":
        a = g.uniform(0.6, 1.6, shape).astype(np.float32)
    elif kind == "bn_count":
        return torch.tensor(1000, dtype=torch.long)
    elif kind in ("se_b1", "se_b2", "linear_bias"):
        a = (g.standard_normal(shape, dtype=np.float32) * np.float32(0.3))
    elif kind == "bias_top":
        # key-heat-map style heads sit slightly negative so that peaks are sparse
        a = (g.standard_normal(shape, dtype=np.float32) * np.float32(0.2)) - np.float32(0.8)
    else:
        raise KeyError(kind)
    return torch.from_numpy(np.ascontiguousarray(a))


def deterministic_state_dict(seed: int = 0, model_size: str = "xl", with_decoder: bool = True,
                             prefix_detector: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """Seeded ``state_dict`` with the reference's key set.

    ``prefix_detector=True, with_decoder=True`` gives ``TextDetectorModel.state_dict()`` keys
    (``detector.*``/``decoder.*``); ``prefix_detector=False`` gives ``CenterNetDetection`` keys.
    Tensor values depend only on (seed, un-prefixed detector key), so both forms agree.
    """
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    if prefix_detector:
        for k, (shape, kind) in text_detector_schema(model_size).items():
            if k.startswith("decoder.") and not with_decoder:
                continue
            base = k[len("detector."):] if k.startswith("detector.") else k
            out[k] = fill_tensor(seed, base, shape, kind)
    else:
        for k, (shape, kind) in detector_schema(model_size).items():
            out[k] = fill_tensor(seed, k, shape, kind)
    return out
