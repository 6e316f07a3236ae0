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


_FILLED: Dict[Tuple[int, str], Dict[str, torch.Tensor]] = {}


def deterministic_state_dict(seed: int = 0, model_size: str = "xl", with_decoder: bool = True,
                             prefix_detector: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """Seeded ``state_dict`` with the reference's key set.

    ``prefix_detector=True, with_decoder=True`` gives ``TextDetectorModel.state_dict()`` keys
    (``detector.*``/``decoder.*``); ``prefix_detector=False`` gives ``CenterNetDetection`` keys.
    Tensor values depend only on (seed, un-prefixed detector key), so both forms agree.
    """
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    memo = _FILLED.setdefault((seed, model_size), {})          # generated once per process; callers get their own copies
    if not memo:
        # every tensor has its own counter-based generator (keyed by its name): fill them on a few threads (numpy's generators release the
        # GIL while they fill) -- ~9 s -> ~2 s for "xl"; the values do not depend on the order
        from concurrent.futures import ThreadPoolExecutor
        import os
        todo = [(k[len("detector."):] if k.startswith("detector.") else k, shape, kind) for k, (shape, kind) in text_detector_schema(model_size).items()]
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            for (base, _, _), t in zip(todo, ex.map(lambda a: fill_tensor(seed, *a), todo)):
                memo[base] = t

    def filled(base, shape, kind):
        t = memo.get(base)
        if t is None:
            t = memo[base] = fill_tensor(seed, base, shape, kind)
        return t.clone()
    if prefix_detector:
        for k, (shape, kind) in text_detector_schema(model_size).items():
            if k.startswith("decoder.") and not with_decoder:
                continue
            base = k[len("detector."):] if k.startswith("detector.") else k
            out[k] = filled(base, shape, kind)
    else:
        for k, (shape, kind) in detector_schema(model_size).items():
            out[k] = filled(k, shape, kind)
    return out


def tf_efficientnetv2_npz_names(model_size: str = "xl"):
    """(state_dict key of the backbone, npz array name, permutation or None) for every tensor the
    reference's TF-checkpoint importer touches (``load_weight``, ``/root/reference/models/detector.py:30-121``):
    conv kernels are HWIO in the npz (-> ``permute(3,2,0,1)``), depthwise kernels HWC1-like (-> ``permute(2,3,0,1)``),
    BatchNorm gamma/beta/moving_mean/moving_variance map to weight/bias/running_mean/running_var.  As in
    the reference, the SE ``fc1``/``fc2`` *biases* are NOT imported (``apply_weights`` only assigns ``weight``)."""
    from .schema import STAGES, backbone_blocks
    root = f"efficientnetv2-{model_size}/"
    bn = [("weight", "gamma"), ("bias", "beta"), ("running_mean", "moving_mean"), ("running_var", "moving_variance")]
    out = []

    def conv(key, name):
        out.append((key, name + "kernel", (3, 2, 0, 1)))

    def norm(prefix, name):
        for a, b in bn:
            out.append((f"{prefix}.{a}", name + b, None))

    conv("backbone.features.0.0.weight", root + "stem/conv2d/")
    norm("backbone.features.0.1", root + "stem/tpu_batch_normalization/")
    idx = 0
    for stage in backbone_blocks(model_size):
        for blk in stage:
            p, t = blk.prefix + ".block", f"{root}blocks_{idx}/"
            if blk.kind == "fused" and blk.exp == blk.cin:
                conv(p + ".0.0.weight", t + "conv2d/")
                norm(p + ".0.1", t + "tpu_batch_normalization/")
            elif blk.kind == "fused":
                conv(p + ".0.0.weight", t + "conv2d/")
                norm(p + ".0.1", t + "tpu_batch_normalization/")
                conv(p + ".1.0.weight", t + "conv2d_1/")
                norm(p + ".1.1", t + "tpu_batch_normalization_1/")
            else:
                conv(p + ".0.0.weight", t + "conv2d/")
                norm(p + ".0.1", t + "tpu_batch_normalization/")
                out.append((p + ".1.0.weight", t + "depthwise_conv2d/depthwise_kernel", (2, 3, 0, 1)))
                norm(p + ".1.1", t + "tpu_batch_normalization_1/")
                conv(p + ".2.fc1.weight", t + "se/conv2d/")
                conv(p + ".2.fc2.weight", t + "se/conv2d_1/")
                conv(p + ".3.0.weight", t + "conv2d_1/")
                norm(p + ".3.1", t + "tpu_batch_normalization_2/")
            idx += 1
    n = len(STAGES[model_size]) + 1
    conv(f"backbone.features.{n}.0.weight", root + "head/conv2d/")
    norm(f"backbone.features.{n}.1", root + "head/tpu_batch_normalization/")
    return out


def load_tf_efficientnetv2_npz(detection_module, weight_path: str, model_size: str = "xl") -> bool:
    """Import a TF EfficientNetV2 ``.npz`` into the backbone of a ``CenterNetDetection`` (mirror of the
    reference's ``load_weight``; returns False and prints ``not found`` when the file is absent, as the
    reference does at ``models/detector.py:34-36``)."""
    import os
    if not os.path.exists(weight_path):
        print("not found:", weight_path)
        return False
    sd = detection_module.state_dict()
    with np.load(weight_path) as w:
        for key, name, perm in tf_efficientnetv2_npz_names(model_size):
            t = torch.from_numpy(w[name])
            sd[key] = t.permute(*perm).contiguous() if perm is not None else t
    detection_module.load_state_dict(sd)
    return True
