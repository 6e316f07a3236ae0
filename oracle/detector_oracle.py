"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the detector forward + 3x3 max-pool NMS.

A functional, pure-``torch`` (CPU, fp32) restatement of the reference's detector forward pass that
works directly on a ``CenterNetDetection`` ``state_dict`` (no nn.Module tree, no product code):

* backbone  -- torchvision EfficientNetV2 blocks as instantiated by
  ``/root/reference/models/detector.py:12-28`` (third-party, un-vendored: see
  ``oracle/tv_efficientnet.py`` for what is restated and why) and tapped by
  ``BackboneModel.forward`` (``models/detector.py:139-146``);
* nine FPN heads -- ``Leafmap.forward`` (``models/detector.py:192-201``, layers ``:164-190``);
* ``CenterNetDetection.forward`` (``models/detector.py:217-230``): ``x*2-1``, concat of the 8 map
  heads, separate ``feature`` head;
* ``CenterNetDetector.forward`` (``models/detector.py:289-296``): -inf pad, 3x3 max-pool, keep
  pixels not smaller than their neighbourhood maximum.

Pinning: ``tests/golden/gen_golden.py`` (run in the build container only) imports the reference's
own ``models/detector.py`` on top of ``oracle/tv_efficientnet.py`` and writes golden outputs;
``tests/test_oracle.py`` checks this oracle against them.  The Leafmap / CenterNetDetection /
CenterNetDetector arithmetic is thereby pinned by the reference's own code; the MBConv/FusedMBConv
block arithmetic is pinned only against the restated torchvision ("parity unpinned" vs the absent
wheel).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

BACKBONE_EPS = 1e-3   # models/detector.py:27
HEAD_EPS = 1e-5       # nn.BatchNorm2d default (models/detector.py:161-184)
HEAD_NAMES = ["keyheatmap", "sizes", "textline", "sepatator", "code1", "code2", "code4", "code8"]


def _bn(sd, p, x, eps):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, eps)


def _cna(sd, p, x, stride=1, groups=1, act=True):
    """Conv2dNormActivation: conv(pad=(k-1)//2, no bias) -> BN(eps 1e-3) -> SiLU?"""
    w = sd[p + ".0.weight"]
    x = F.conv2d(x, w, None, stride, (w.shape[-1] - 1) // 2, 1, groups)
    x = _bn(sd, p + ".1", x, BACKBONE_EPS)
    return F.silu(x) if act else x


def _stage_blocks(sd, prefix) -> List[str]:
    j, out = 0, []
    while f"{prefix}.{j}.block.0.0.weight" in sd:
        out.append(f"{prefix}.{j}")
        j += 1
    return out


def _block(sd, p, x):
    """One FusedMBConv / MBConv block in eval mode (StochasticDepth = identity)."""
    b = p + ".block"
    w0 = sd[b + ".0.0.weight"]
    if f"{b}.2.fc1.weight" in sd:                       # MBConv: expand, depthwise, SE, project
        dw = sd[b + ".1.0.weight"]
        stride = 2 if _mb_stride2(sd, p) else 1
        y = _cna(sd, b + ".0", x)
        y = _cna(sd, b + ".1", y, stride=stride, groups=dw.shape[0])
        s = F.adaptive_avg_pool2d(y, 1)
        s = F.silu(F.conv2d(s, sd[b + ".2.fc1.weight"], sd[b + ".2.fc1.bias"]))
        s = torch.sigmoid(F.conv2d(s, sd[b + ".2.fc2.weight"], sd[b + ".2.fc2.bias"]))
        y = s * y
        y = _cna(sd, b + ".3", y, act=False)
        cout = sd[b + ".3.0.weight"].shape[0]
    elif f"{b}.1.0.weight" in sd:                       # FusedMBConv, expand != 1
        stride = 2 if _mb_stride2(sd, p) else 1
        y = _cna(sd, b + ".0", x, stride=stride)
        y = _cna(sd, b + ".1", y, act=False)
        cout = sd[b + ".1.0.weight"].shape[0]
    else:                                               # FusedMBConv, expand == 1
        stride = 2 if _mb_stride2(sd, p) else 1
        y = _cna(sd, b + ".0", x, stride=stride)
        cout = w0.shape[0]
    if stride == 1 and x.shape[1] == cout:
        y = y + x
    return y


# Strides are not in the state_dict; they come from the config rows (models/detector.py:14-20 and
# torchvision's s/m/l tables): the first block of every stage except features[1] and the
# "stride 1" MBConv stages has stride 2.  Encoded by stage index -> (first-block stride).
_STAGE_STRIDE = {1: 1, 2: 2, 3: 2, 4: 2, 5: 1, 6: 2, 7: 1}


def _mb_stride2(sd, p) -> bool:
    parts = p.split(".")
    stage, j = int(parts[-2]), int(parts[-1])
    return j == 0 and _STAGE_STRIDE[stage] == 2


def backbone_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, prefix="backbone.features") -> List[torch.Tensor]:
    """BackboneModel.forward (models/detector.py:139-146): taps after features[2], [3], [5], last."""
    taps = []
    x = _cna(sd, f"{prefix}.0", x, stride=2)
    i = 1
    while f"{prefix}.{i}.0.block.0.0.weight" in sd:
        for p in _stage_blocks(sd, f"{prefix}.{i}"):
            x = _block(sd, p, x)
        if i in (2, 3, 5):
            taps.append(x)
        i += 1
    x = _cna(sd, f"{prefix}.{i}", x)
    taps.append(x)
    return taps


def leafmap_forward(sd, name: str, taps: List[torch.Tensor]) -> torch.Tensor:
    """Leafmap.forward (models/detector.py:192-201)."""
    y = None
    n = len(taps)
    for i, x in enumerate(reversed(taps)):
        x = _bn(sd, f"{name}.in_bn.{n - 1 - i}", x, HEAD_EPS)
        if y is not None:
            x = torch.cat([y, x], dim=1)
        y = F.conv2d(x, sd[f"{name}.upsamplers.{i}.0.weight"], None, 1, 1)
        y = _bn(sd, f"{name}.upsamplers.{i}.1", y, HEAD_EPS)
        y = F.gelu(y)                                                    # exact (erf) GELU, :169
        if i < n - 1:
            y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True)   # :170
    return F.conv2d(y, sd[f"{name}.top_conv.0.weight"], sd[f"{name}.top_conv.0.bias"], 1, 1)


def detection_forward(sd, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """CenterNetDetection.forward (models/detector.py:217-230). x: [B,3,H,W] in 0..1."""
    x = x * 2 - 1
    taps = backbone_forward(sd, x)
    maps = torch.cat([leafmap_forward(sd, h, taps) for h in HEAD_NAMES], dim=1)
    return maps, leafmap_forward(sd, "feature", taps)


def nms_forward(maps: torch.Tensor) -> torch.Tensor:
    """CenterNetDetector.forward tail (models/detector.py:291-296): [B,9,h,w] -> [B,10,h,w]."""
    keymap = maps[:, 0:1]
    minval = torch.tensor(float("-inf"), dtype=keymap.dtype)
    lp = F.pad(keymap, (1, 1, 1, 1), value=float("-inf"))
    lp = F.max_pool2d(lp, kernel_size=3, stride=1)
    detected = torch.where(keymap < lp, minval, keymap)
    return torch.cat([keymap, detected, maps[:, 1:]], dim=1)


@torch.no_grad()
def detector_forward(sd, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """CenterNetDetector.forward (models/detector.py:289-296) -> (heatmap[B,10,h,w], features[B,100,h,w])."""
    sd = {k: v for k, v in sd.items()}
    if any(k.startswith("detector.") for k in sd):
        sd = {k[len("detector."):]: v for k, v in sd.items() if k.startswith("detector.")}
    maps, feat = detection_forward(sd, x)
    return nms_forward(maps), feat


# ----------------------------------------------------------------------------------------------------------------------
# Training-mode forward without gradients: what the reference's end-of-epoch BN-refresh pass runs (train1.py:203-211 calls
# train_step under torch.no_grad() with the model in train()): every BatchNorm normalises with the statistics of the batch and
# moves its running statistics (momentum 0.1, unbiased variance), and the residual branches go through torchvision's
# StochasticDepth("row"): branch * keep[b] with keep[b] in {0, 1/(1-p)}.  The random draw is an INPUT here (`keep`: block
# prefix -> [B] tensor; missing = ones), so that a test can feed the reference and the GPU the same draw.
# ----------------------------------------------------------------------------------------------------------------------
class TrainCtx:
    def __init__(self, keep=None, momentum: float = 0.1):
        self.keep = keep or {}
        self.momentum = momentum
        self.new_stats: Dict[str, torch.Tensor] = {}


def _bn_t(sd, p, x, eps, ctx: TrainCtx):
    rm, rv = sd[p + ".running_mean"].clone(), sd[p + ".running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], True, ctx.momentum, eps)
    ctx.new_stats[p + ".running_mean"], ctx.new_stats[p + ".running_var"] = rm, rv
    return y


def _cna_t(sd, p, x, ctx, stride=1, groups=1, act=True):
    w = sd[p + ".0.weight"]
    x = F.conv2d(x, w, None, stride, (w.shape[-1] - 1) // 2, 1, groups)
    x = _bn_t(sd, p + ".1", x, BACKBONE_EPS, ctx)
    return F.silu(x) if act else x


def _block_t(sd, p, x, ctx):
    b = p + ".block"
    stride = 2 if _mb_stride2(sd, p) else 1
    if f"{b}.2.fc1.weight" in sd:
        dw = sd[b + ".1.0.weight"]
        y = _cna_t(sd, b + ".0", x, ctx)
        y = _cna_t(sd, b + ".1", y, ctx, stride=stride, groups=dw.shape[0])
        s = F.adaptive_avg_pool2d(y, 1)
        s = F.silu(F.conv2d(s, sd[b + ".2.fc1.weight"], sd[b + ".2.fc1.bias"]))
        s = torch.sigmoid(F.conv2d(s, sd[b + ".2.fc2.weight"], sd[b + ".2.fc2.bias"]))
        y = _cna_t(sd, b + ".3", s * y, ctx, act=False)
        cout = sd[b + ".3.0.weight"].shape[0]
    elif f"{b}.1.0.weight" in sd:
        y = _cna_t(sd, b + ".0", x, ctx, stride=stride)
        y = _cna_t(sd, b + ".1", y, ctx, act=False)
        cout = sd[b + ".1.0.weight"].shape[0]
    else:
        y = _cna_t(sd, b + ".0", x, ctx, stride=stride)
        cout = sd[b + ".0.0.weight"].shape[0]
    if stride == 1 and x.shape[1] == cout:
        k = ctx.keep.get(p)
        if k is not None:
            y = y * k.to(y.dtype).reshape(-1, 1, 1, 1)
        y = y + x
    return y


def residual_blocks(sd, prefix="backbone.features") -> List[str]:
    """Prefixes of the blocks that have a residual connection (and therefore a StochasticDepth draw), in network order."""
    out = []
    i = 1
    while f"{prefix}.{i}.0.block.0.0.weight" in sd:
        for p in _stage_blocks(sd, f"{prefix}.{i}"):
            b = p + ".block"
            last = ".3" if f"{b}.2.fc1.weight" in sd else (".1" if f"{b}.1.0.weight" in sd else ".0")
            cout, cin = sd[b + last + ".0.weight"].shape[0], sd[b + ".0.0.weight"].shape[1]
            if not _mb_stride2(sd, p) and cin == cout:
                out.append(p)
        i += 1
    return out


def stochastic_depth_probs(sd, prefix="backbone.features", p_total: float = 0.2) -> Dict[str, float]:
    """torchvision EfficientNet.__init__: sd_prob = stochastic_depth_prob * block_id / total_blocks, block_id counting every block."""
    blocks = []
    i = 1
    while f"{prefix}.{i}.0.block.0.0.weight" in sd:
        blocks += _stage_blocks(sd, f"{prefix}.{i}")
        i += 1
    return {p: p_total * j / len(blocks) for j, p in enumerate(blocks)}


def leafmap_forward_train(sd, name, taps, ctx):
    y = None
    n = len(taps)
    for i, x in enumerate(reversed(taps)):
        x = _bn_t(sd, f"{name}.in_bn.{n - 1 - i}", x, HEAD_EPS, ctx)
        if y is not None:
            x = torch.cat([y, x], dim=1)
        y = F.conv2d(x, sd[f"{name}.upsamplers.{i}.0.weight"], None, 1, 1)
        y = F.gelu(_bn_t(sd, f"{name}.upsamplers.{i}.1", y, HEAD_EPS, ctx))
        if i < n - 1:
            y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(y, sd[f"{name}.top_conv.0.weight"], sd[f"{name}.top_conv.0.bias"], 1, 1)


@torch.no_grad()
def detection_forward_train(sd, x: torch.Tensor, keep=None, momentum: float = 0.1):
    """CenterNetDetection.forward in train() mode -> (maps [B,9,h,w], features [B,100,h,w], {running-stat key: new value})."""
    sd = {k: v for k, v in sd.items()}
    if any(k.startswith("detector.") for k in sd):
        keep = {k[len("detector."):] if k.startswith("detector.") else k: v for k, v in (keep or {}).items()}
        sd = {k[len("detector."):]: v for k, v in sd.items() if k.startswith("detector.")}
    ctx = TrainCtx(keep, momentum)
    prefix = "backbone.features"
    x = x * 2 - 1
    taps = []
    x = _cna_t(sd, f"{prefix}.0", x, ctx, stride=2)
    i = 1
    while f"{prefix}.{i}.0.block.0.0.weight" in sd:
        for p in _stage_blocks(sd, f"{prefix}.{i}"):
            x = _block_t(sd, p, x, ctx)
        if i in (2, 3, 5):
            taps.append(x)
        i += 1
    taps.append(_cna_t(sd, f"{prefix}.{i}", x, ctx))
    maps = torch.cat([leafmap_forward_train(sd, h, taps, ctx) for h in HEAD_NAMES], dim=1)
    return maps, leafmap_forward_train(sd, "feature", taps, ctx), ctx.new_stats


@torch.no_grad()
def decoder_forward_train(sd, rows: torch.Tensor, momentum: float = 0.1, prefix="decoder."):
    """SimpleDecoder.forward in train() mode (models/detector.py:232-254): Linear -> BatchNorm1d (batch stats) -> GELU, twice, Linear."""
    outs, new = [], {}
    j = 0
    while f"{prefix}blocks.{j}.0.weight" in sd:
        b = f"{prefix}blocks.{j}"
        y = rows
        for li, bi in ((0, 1), (3, 4)):
            y = F.linear(y, sd[f"{b}.{li}.weight"])
            rm, rv = sd[f"{b}.{bi}.running_mean"].clone(), sd[f"{b}.{bi}.running_var"].clone()
            y = F.gelu(F.batch_norm(y, rm, rv, sd[f"{b}.{bi}.weight"], sd[f"{b}.{bi}.bias"], True, momentum, 1e-5))
            new[f"{b}.{bi}.running_mean"], new[f"{b}.{bi}.running_var"] = rm, rv
        outs.append(F.linear(y, sd[f"{b}.6.weight"], sd[f"{b}.6.bias"]))
        j += 1
    return outs, new
