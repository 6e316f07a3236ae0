"""TEST INFRASTRUCTURE ONLY -- restatement of the un-vendored third-party dependency.

The reference's detector (``/root/reference/models/detector.py:1-2, 12-28, 123-137``) builds its
backbone out of ``torchvision.models.efficientnet`` classes.  torchvision is a PyPI dependency with
no pinned version (``README.en.md:136``), it is not vendored under ``/root/reference`` and it is
not installed in this image.  This file restates torchvision's *published* EfficientNetV2 building
blocks (``torchvision/models/efficientnet.py``, ``torchvision/ops/misc.py``,
``torchvision/ops/stochastic_depth.py``, torchvision >= 0.13) from their documented behaviour:

* ``Conv2dNormActivation``: Sequential[Conv2d(pad=(k-1)//2, bias=norm is None), norm, act]
* ``SqueezeExcitation``: avgpool -> fc1 (1x1 conv, bias) -> act -> fc2 (1x1 conv, bias) -> sigmoid -> x * s
* ``StochasticDepth``: identity when not training
* ``MBConv`` / ``FusedMBConv`` and the ``EfficientNet.features`` stack

What pins the structure from the reference side: the attribute traversal of
``load_weight`` (``models/detector.py:69-120``: ``sec.block[0..3]``, ``block[2].fc1/.fc2``,
``len(sec.block) in {1,2,4}``), the tap indices (``:141-145``), ``Leafmap.in_dims`` (``:152``) and
the published parameter count (206,838,808 backbone parameters) -- all checked in
``tests/test_oracle.py``.

Only ``tests/``, ``__graft_entry__.smoke()``, ``bench.py``'s cpu_baseline leg and the golden-vector
generator may import this module.  The block arithmetic restated here is "parity unpinned" with
respect to the real torchvision wheel (absent); everything the reference itself owns (FPN heads,
NMS, decode) is pinned by golden vectors produced by the reference's own code.
"""
from __future__ import annotations

import math
from functools import partial
from typing import Callable, List, Optional, Sequence

import torch
from torch import nn


def _make_divisible(v: float, divisor: int, min_value: Optional[int] = None) -> int:
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class Conv2dNormActivation(nn.Sequential):
    def __init__(self, cin, cout, kernel_size=3, stride=1, groups=1,
                 norm_layer: Optional[Callable[..., nn.Module]] = nn.BatchNorm2d,
                 activation_layer: Optional[Callable[..., nn.Module]] = nn.ReLU):
        layers: List[nn.Module] = [
            nn.Conv2d(cin, cout, kernel_size, stride, padding=(kernel_size - 1) // 2,
                      groups=groups, bias=norm_layer is None)
        ]
        if norm_layer is not None:
            layers.append(norm_layer(cout))
        if activation_layer is not None:
            layers.append(activation_layer(inplace=True))
        super().__init__(*layers)
        self.out_channels = cout


class SqueezeExcitation(nn.Module):
    def __init__(self, input_channels, squeeze_channels, activation=nn.ReLU,
                 scale_activation=nn.Sigmoid):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(input_channels, squeeze_channels, 1)
        self.fc2 = nn.Conv2d(squeeze_channels, input_channels, 1)
        self.activation = activation()
        self.scale_activation = scale_activation()

    def _scale(self, x):
        s = self.avgpool(x)
        s = self.fc1(s)
        s = self.activation(s)
        s = self.fc2(s)
        return self.scale_activation(s)

    def forward(self, x):
        return self._scale(x) * x


class StochasticDepth(nn.Module):
    def __init__(self, p: float, mode: str):
        super().__init__()
        self.p = p
        self.mode = mode

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        keep = 1.0 - self.p
        shape = [x.shape[0]] + [1] * (x.ndim - 1) if self.mode == "row" else [1] * x.ndim
        noise = torch.empty(shape, dtype=x.dtype, device=x.device).bernoulli_(keep)
        if keep > 0.0:
            noise.div_(keep)
        return x * noise


class _MBConvConfig:
    def __init__(self, expand_ratio, kernel, stride, input_channels, out_channels, num_layers, block):
        self.expand_ratio = expand_ratio
        self.kernel = kernel
        self.stride = stride
        self.input_channels = input_channels
        self.out_channels = out_channels
        self.num_layers = num_layers
        self.block = block

    @staticmethod
    def adjust_channels(channels: int, width_mult: float, min_value: Optional[int] = None) -> int:
        return _make_divisible(channels * width_mult, 8, min_value)


class MBConvConfig(_MBConvConfig):
    def __init__(self, expand_ratio, kernel, stride, input_channels, out_channels, num_layers,
                 width_mult: float = 1.0, depth_mult: float = 1.0, block=None):
        input_channels = self.adjust_channels(input_channels, width_mult)
        out_channels = self.adjust_channels(out_channels, width_mult)
        num_layers = int(math.ceil(num_layers * depth_mult))
        super().__init__(expand_ratio, kernel, stride, input_channels, out_channels, num_layers,
                         block if block is not None else MBConv)


class FusedMBConvConfig(_MBConvConfig):
    def __init__(self, expand_ratio, kernel, stride, input_channels, out_channels, num_layers, block=None):
        super().__init__(expand_ratio, kernel, stride, input_channels, out_channels, num_layers,
                         block if block is not None else FusedMBConv)


class MBConv(nn.Module):
    def __init__(self, cnf: MBConvConfig, stochastic_depth_prob: float, norm_layer, se_layer=SqueezeExcitation):
        super().__init__()
        if not 1 <= cnf.stride <= 2:
            raise ValueError("illegal stride value")
        self.use_res_connect = cnf.stride == 1 and cnf.input_channels == cnf.out_channels
        act = nn.SiLU
        layers: List[nn.Module] = []
        exp = cnf.adjust_channels(cnf.input_channels, cnf.expand_ratio)
        if exp != cnf.input_channels:
            layers.append(Conv2dNormActivation(cnf.input_channels, exp, 1, norm_layer=norm_layer,
                                               activation_layer=act))
        layers.append(Conv2dNormActivation(exp, exp, cnf.kernel, cnf.stride, groups=exp,
                                           norm_layer=norm_layer, activation_layer=act))
        squeeze = max(1, cnf.input_channels // 4)
        layers.append(se_layer(exp, squeeze, activation=partial(nn.SiLU, inplace=True)))
        layers.append(Conv2dNormActivation(exp, cnf.out_channels, 1, norm_layer=norm_layer,
                                           activation_layer=None))
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = StochasticDepth(stochastic_depth_prob, "row")
        self.out_channels = cnf.out_channels

    def forward(self, x):
        r = self.block(x)
        if self.use_res_connect:
            r = self.stochastic_depth(r)
            r += x
        return r


class FusedMBConv(nn.Module):
    def __init__(self, cnf: FusedMBConvConfig, stochastic_depth_prob: float, norm_layer):
        super().__init__()
        if not 1 <= cnf.stride <= 2:
            raise ValueError("illegal stride value")
        self.use_res_connect = cnf.stride == 1 and cnf.input_channels == cnf.out_channels
        act = nn.SiLU
        layers: List[nn.Module] = []
        exp = cnf.adjust_channels(cnf.input_channels, cnf.expand_ratio)
        if exp != cnf.input_channels:
            layers.append(Conv2dNormActivation(cnf.input_channels, exp, cnf.kernel, cnf.stride,
                                               norm_layer=norm_layer, activation_layer=act))
            layers.append(Conv2dNormActivation(exp, cnf.out_channels, 1, norm_layer=norm_layer,
                                               activation_layer=None))
        else:
            layers.append(Conv2dNormActivation(cnf.input_channels, cnf.out_channels, cnf.kernel, cnf.stride,
                                               norm_layer=norm_layer, activation_layer=act))
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = StochasticDepth(stochastic_depth_prob, "row")
        self.out_channels = cnf.out_channels

    def forward(self, x):
        r = self.block(x)
        if self.use_res_connect:
            r = self.stochastic_depth(r)
            r += x
        return r


class EfficientNet(nn.Module):
    def __init__(self, inverted_residual_setting: Sequence[_MBConvConfig], dropout: float,
                 stochastic_depth_prob: float = 0.2, num_classes: int = 1000,
                 norm_layer: Optional[Callable[..., nn.Module]] = None,
                 last_channel: Optional[int] = None):
        super().__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        layers: List[nn.Module] = []
        first = inverted_residual_setting[0].input_channels
        layers.append(Conv2dNormActivation(3, first, 3, 2, norm_layer=norm_layer, activation_layer=nn.SiLU))
        total = sum(c.num_layers for c in inverted_residual_setting)
        block_id = 0
        for cnf in inverted_residual_setting:
            stage: List[nn.Module] = []
            for i in range(cnf.num_layers):
                c = _MBConvConfig(cnf.expand_ratio, cnf.kernel, cnf.stride, cnf.input_channels,
                                  cnf.out_channels, cnf.num_layers, cnf.block)
                if stage:
                    c.input_channels = c.out_channels
                    c.stride = 1
                sd = stochastic_depth_prob * float(block_id) / total
                stage.append(c.block(c, sd, norm_layer))
                block_id += 1
            layers.append(nn.Sequential(*stage))
        last_in = inverted_residual_setting[-1].out_channels
        last_out = last_channel if last_channel is not None else 4 * last_in
        layers.append(Conv2dNormActivation(last_in, last_out, 1, norm_layer=norm_layer, activation_layer=nn.SiLU))
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(nn.Dropout(p=dropout, inplace=True), nn.Linear(last_out, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                r = 1.0 / math.sqrt(m.out_features)
                nn.init.uniform_(m.weight, -r, r)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.features(x)
        x = self.avgpool(x)
        return self.classifier(torch.flatten(x, 1))


# torchvision's published EfficientNetV2 configurations (models/efficientnet.py `_efficientnet_conf`), used by the reference for
# model_size 's' / 'm' / 'l' (models/detector.py:131-136): (block, expand_ratio, kernel, stride, in, out, layers), last_channel 1280,
# norm_layer = BatchNorm2d(eps=1e-3), dropout 0.2 / 0.3 / 0.4, stochastic depth 0.2.
_V2_CONF = {
    "s": ([("f", 1, 3, 1, 24, 24, 2), ("f", 4, 3, 2, 24, 48, 4), ("f", 4, 3, 2, 48, 64, 4), ("m", 4, 3, 2, 64, 128, 6), ("m", 6, 3, 1, 128, 160, 9),
           ("m", 6, 3, 2, 160, 256, 15)], 0.2),
    "m": ([("f", 1, 3, 1, 24, 24, 3), ("f", 4, 3, 2, 24, 48, 5), ("f", 4, 3, 2, 48, 80, 5), ("m", 4, 3, 2, 80, 160, 7), ("m", 6, 3, 1, 160, 176, 14),
           ("m", 6, 3, 2, 176, 304, 18), ("m", 6, 3, 1, 304, 512, 5)], 0.3),
    "l": ([("f", 1, 3, 1, 32, 32, 4), ("f", 4, 3, 2, 32, 64, 7), ("f", 4, 3, 2, 64, 96, 7), ("m", 4, 3, 2, 96, 192, 10), ("m", 6, 3, 1, 192, 224, 19),
           ("m", 6, 3, 2, 224, 384, 25), ("m", 6, 3, 1, 384, 640, 7)], 0.4),
}


def _v2_factory(size):
    def make(weights=None, progress=True, **kwargs):
        if weights is not None:
            raise RuntimeError("pretrained torchvision weights are not available offline")
        rows, dropout = _V2_CONF[size]
        setting = [(FusedMBConvConfig if b == "f" else MBConvConfig)(e, k, s_, i, o, n) for b, e, k, s_, i, o, n in rows]
        from functools import partial
        return EfficientNet(setting, dropout, last_channel=1280, norm_layer=partial(nn.BatchNorm2d, eps=1e-03), **kwargs)
    return make


def install_as_torchvision() -> None:
    """Register this restatement under the module names ``models/detector.py:1-2`` imports, so the
    reference's own ``detector.py`` can be imported unchanged in this container (golden-vector
    generation and oracle validation only; never on the GPU box)."""
    import sys
    import types

    if "torchvision" in sys.modules and not getattr(sys.modules["torchvision"], "_ftc_restated", False):
        return  # a real torchvision exists: use it
    tv = types.ModuleType("torchvision")
    tv._ftc_restated = True
    models = types.ModuleType("torchvision.models")
    eff = types.ModuleType("torchvision.models.efficientnet")
    eff.EfficientNet = EfficientNet
    eff.MBConvConfig = MBConvConfig
    eff.FusedMBConvConfig = FusedMBConvConfig
    models.efficientnet = eff
    models.efficientnet_v2_s = _v2_factory("s")
    models.efficientnet_v2_m = _v2_factory("m")
    models.efficientnet_v2_l = _v2_factory("l")
    tv.models = models
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models
    sys.modules["torchvision.models.efficientnet"] = eff
