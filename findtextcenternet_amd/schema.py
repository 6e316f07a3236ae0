"""Structure of the detector network as plain data.

Single source of truth on the product side for (a) the ``state_dict`` key/shape schema the
reference checkpoints use (``/root/reference/models/detector.py:203-260``; key prefixes listed in
SURVEY.md Appendix D) and (b) the layer sequence the HIP plan is built from (``plan.py``).

Backbone rows follow ``efficientnet_v2_xl`` (``models/detector.py:12-28``) and, for the s/m/l sizes
the reference can also instantiate (``:131-136``), torchvision's published EfficientNetV2 tables.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple

# util_func.py:5-9
modulo_list = [1091, 1093, 1097]
width = 768
height = 768
scale = 4
feature_dim = 100

BACKBONE_BN_EPS = 1e-3   # models/detector.py:27 (and torchvision's v2 factories)
HEAD_BN_EPS = 1e-5       # nn.BatchNorm2d default, models/detector.py:161-184
FPN_DIM = 192            # models/detector.py:159

# (block kind, expand_ratio, kernel, stride, cin, cout, num_layers)
STAGES: Dict[str, List[Tuple[str, int, int, int, int, int, int]]] = {
    "xl": [("fused", 1, 3, 1, 32, 32, 4), ("fused", 4, 3, 2, 32, 64, 8), ("fused", 4, 3, 2, 64, 96, 8),
           ("mb", 4, 3, 2, 96, 192, 16), ("mb", 6, 3, 1, 192, 256, 24), ("mb", 6, 3, 2, 256, 512, 32),
           ("mb", 6, 3, 1, 512, 640, 8)],
    "l": [("fused", 1, 3, 1, 32, 32, 4), ("fused", 4, 3, 2, 32, 64, 7), ("fused", 4, 3, 2, 64, 96, 7),
          ("mb", 4, 3, 2, 96, 192, 10), ("mb", 6, 3, 1, 192, 224, 19), ("mb", 6, 3, 2, 224, 384, 25),
          ("mb", 6, 3, 1, 384, 640, 7)],
    "m": [("fused", 1, 3, 1, 24, 24, 3), ("fused", 4, 3, 2, 24, 48, 5), ("fused", 4, 3, 2, 48, 80, 5),
          ("mb", 4, 3, 2, 80, 160, 7), ("mb", 6, 3, 1, 160, 176, 14), ("mb", 6, 3, 2, 176, 304, 18),
          ("mb", 6, 3, 1, 304, 512, 5)],
    "s": [("fused", 1, 3, 1, 24, 24, 2), ("fused", 4, 3, 2, 24, 48, 4), ("fused", 4, 3, 2, 48, 64, 4),
          ("mb", 4, 3, 2, 64, 128, 6), ("mb", 6, 3, 1, 128, 160, 9), ("mb", 6, 3, 2, 160, 256, 15)],
}
LAST_CHANNEL = 1280
# Leafmap.in_dims, models/detector.py:151-158
TAP_DIMS = {"xl": [64, 96, 256, 1280], "l": [64, 96, 224, 1280], "m": [48, 80, 176, 1280], "s": [48, 64, 160, 1280]}
TAP_FEATURES = (2, 3, 5)   # BackboneModel.forward, models/detector.py:143 (+ the last one)

# CenterNetDetection heads in forward order (models/detector.py:207-230); note the reference's
# spelling "sepatator".  (name, out_dim, first channel in the 9-channel map or -1 for `feature`)
HEADS = [("keyheatmap", 1, 0), ("sizes", 2, 1), ("textline", 1, 3), ("sepatator", 1, 4),
         ("code1", 1, 5), ("code2", 1, 6), ("code4", 1, 7), ("code8", 1, 8), ("feature", feature_dim, -1)]


def make_divisible(v: float, divisor: int = 8) -> int:
    """torchvision ``_make_divisible`` as used by ``_MBConvConfig.adjust_channels``."""
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


@dataclass
class BlockSpec:
    kind: str        # "fused" | "mb"
    prefix: str      # e.g. "backbone.features.4.0"
    cin: int
    cout: int
    exp: int
    stride: int
    squeeze: int     # SE squeeze channels (mb only)
    residual: bool


def backbone_blocks(model_size: str = "xl") -> List[List[BlockSpec]]:
    """Blocks of ``features[1..]`` (one list per stage), mirroring torchvision's EfficientNet ctor:
    blocks after the first of a stage get ``input_channels=out_channels, stride=1``."""
    out: List[List[BlockSpec]] = []
    for si, (kind, e, k, s, cin, cout, n) in enumerate(STAGES[model_size]):
        assert k == 3
        stage = []
        for j in range(n):
            bcin = cin if j == 0 else cout
            bstride = s if j == 0 else 1
            stage.append(BlockSpec(kind, f"backbone.features.{si + 1}.{j}", bcin, cout,
                                   make_divisible(bcin * e), bstride, max(1, bcin // 4),
                                   bstride == 1 and bcin == cout))
        out.append(stage)
    return out


def _bn(d: "OrderedDict[str, Tuple[Tuple[int, ...], str]]", prefix: str, c: int) -> None:
    d[prefix + ".weight"] = ((c,), "bn_weight")
    d[prefix + ".bias"] = ((c,), "bn_bias")
    d[prefix + ".running_mean"] = ((c,), "bn_mean")
    d[prefix + ".running_var"] = ((c,), "bn_var")
    d[prefix + ".num_batches_tracked"] = ((), "bn_count")


def detector_schema(model_size: str = "xl") -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """name -> (shape, kind) for ``CenterNetDetection.state_dict()`` in the reference's key order."""
    d: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    stages = backbone_blocks(model_size)
    c0 = STAGES[model_size][0][4]
    d["backbone.features.0.0.weight"] = ((c0, 3, 3, 3), "conv")
    _bn(d, "backbone.features.0.1", c0)
    for stage in stages:
        for b in stage:
            p = b.prefix + ".block"
            if b.kind == "fused":
                if b.exp != b.cin:
                    d[f"{p}.0.0.weight"] = ((b.exp, b.cin, 3, 3), "conv")
                    _bn(d, f"{p}.0.1", b.exp)
                    d[f"{p}.1.0.weight"] = ((b.cout, b.exp, 1, 1), "conv_proj")
                    _bn(d, f"{p}.1.1", b.cout)
                else:
                    d[f"{p}.0.0.weight"] = ((b.cout, b.cin, 3, 3), "conv")
                    _bn(d, f"{p}.0.1", b.cout)
            else:
                d[f"{p}.0.0.weight"] = ((b.exp, b.cin, 1, 1), "conv")
                _bn(d, f"{p}.0.1", b.exp)
                d[f"{p}.1.0.weight"] = ((b.exp, 1, 3, 3), "conv_dw")
                _bn(d, f"{p}.1.1", b.exp)
                d[f"{p}.2.fc1.weight"] = ((b.squeeze, b.exp, 1, 1), "se_w1")
                d[f"{p}.2.fc1.bias"] = ((b.squeeze,), "se_b1")
                d[f"{p}.2.fc2.weight"] = ((b.exp, b.squeeze, 1, 1), "se_w2")
                d[f"{p}.2.fc2.bias"] = ((b.exp,), "se_b2")
                d[f"{p}.3.0.weight"] = ((b.cout, b.exp, 1, 1), "conv_proj")
                _bn(d, f"{p}.3.1", b.cout)
    nfeat = len(stages) + 1
    clast = STAGES[model_size][-1][5]
    d[f"backbone.features.{nfeat}.0.weight"] = ((LAST_CHANNEL, clast, 1, 1), "conv")
    _bn(d, f"backbone.features.{nfeat}.1", LAST_CHANNEL)
    taps = TAP_DIMS[model_size]
    for name, out_dim, _ in HEADS:
        for i, c in enumerate(taps):
            _bn(d, f"{name}.in_bn.{i}", c)
        for i, c in enumerate(reversed(taps)):
            cin = c if i == 0 else c + FPN_DIM
            d[f"{name}.upsamplers.{i}.0.weight"] = ((FPN_DIM, cin, 3, 3), "conv")
            _bn(d, f"{name}.upsamplers.{i}.1", FPN_DIM)
        d[f"{name}.top_conv.0.weight"] = ((out_dim, FPN_DIM, 3, 3), "conv_top")
        d[f"{name}.top_conv.0.bias"] = ((out_dim,), "bias_top")
    return d


def decoder_schema() -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """``SimpleDecoder.state_dict()`` (models/detector.py:232-254); carried for checkpoint
    compatibility only -- the decoder is outside the inference hot path."""
    d: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    mid = 2048
    for i, m in enumerate(modulo_list):
        p = f"blocks.{i}"
        d[f"{p}.0.weight"] = ((mid, feature_dim), "linear")
        d[f"{p}.1.weight"] = ((mid,), "bn_weight")
        d[f"{p}.1.bias"] = ((mid,), "bn_bias")
        d[f"{p}.1.running_mean"] = ((mid,), "bn_mean")
        d[f"{p}.1.running_var"] = ((mid,), "bn_var")
        d[f"{p}.1.num_batches_tracked"] = ((), "bn_count")
        d[f"{p}.3.weight"] = ((mid, mid), "linear")
        d[f"{p}.4.weight"] = ((mid,), "bn_weight")
        d[f"{p}.4.bias"] = ((mid,), "bn_bias")
        d[f"{p}.4.running_mean"] = ((mid,), "bn_mean")
        d[f"{p}.4.running_var"] = ((mid,), "bn_var")
        d[f"{p}.4.num_batches_tracked"] = ((), "bn_count")
        d[f"{p}.6.weight"] = ((m, mid), "linear")
        d[f"{p}.6.bias"] = ((m,), "linear_bias")
    return d


def text_detector_schema(model_size: str = "xl") -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """``TextDetectorModel.state_dict()`` = ``detector.*`` + ``decoder.*`` (models/detector.py:256-260)."""
    d: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    for k, v in detector_schema(model_size).items():
        d["detector." + k] = v
    for k, v in decoder_schema().items():
        d["decoder." + k] = v
    return d
