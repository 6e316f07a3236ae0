"""Drop-in mirror of the reference's detector modules, executing on hand-written HIP kernels.

Same names, constructor arguments, ``state_dict`` keys and tensor signatures as
``/root/reference/models/detector.py``:

* ``TextDetectorModel(pre_weights=True, model_size='xl')`` with ``.detector`` / ``.decoder`` (``:256-260``)
* ``CenterNetDetection.forward(x[B,3,H,W] in 0..1) -> (maps[B,9,H/4,W/4], feature[B,100,H/4,W/4])`` (``:217-230``)
* ``CenterNetDetector(detector).forward(x) -> (heatmap[B,10,H/4,W/4], features[B,100,H/4,W/4])`` (``:289-296``)

so ``process_ocr_torch.py:12-27`` / ``test_image1_torch.py:57-73`` work by changing one import.
The modules are parameter containers (so ``load_state_dict`` / ``.to()`` / ``.eval()`` behave as in
the reference); ``forward`` packs the weights once and enqueues the HIP plan through the C ABI
(``include/ftc.h``).  There is NO eager-PyTorch or CPU forward: without a gfx950 device and the
built ``libftc_hip.so`` a call raises.

Numeric modes (``precision`` argument, or env ``FTC_PRECISION``):
``"fp32"`` (default; exact-f32 MFMA, the parity mode -- what the reference computes on CUDA/CPU) and
``"bf16"`` (bf16 MFMA with fp32 accumulation, fp32 residual trunk and fp32 outputs -- the speed mode
BASELINE.json's config 2 names).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from . import _lib as L
from . import plan as P
from .schema import decoder_schema, detector_schema, feature_dim


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, is_buffer: bool) -> None:
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, nn.Module())
        m = m._modules[p]
    if is_buffer:
        m.register_buffer(parts[-1], tensor)
    else:
        m.register_parameter(parts[-1], nn.Parameter(tensor))


def _init_like_reference(shape, kind: str) -> torch.Tensor:
    """Random init in the spirit of torchvision's EfficientNet / nn defaults (values are not
    meant to match a particular RNG stream: untrained weights carry no meaning)."""
    if kind in ("conv", "conv_proj", "conv_dw", "se_w1", "se_w2"):
        fan_out = shape[0] * shape[2] * shape[3]
        return torch.randn(shape) * math.sqrt(2.0 / fan_out)
    if kind in ("conv_top", "linear"):
        fan_in = int(torch.tensor(shape[1:]).prod())
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape) * 2 - 1) * bound
    if kind in ("bias_top", "linear_bias"):
        return (torch.rand(shape) * 2 - 1) * 0.02
    if kind in ("bn_weight", "bn_var"):
        return torch.ones(shape)
    if kind == "bn_count":
        return torch.tensor(0, dtype=torch.long)
    return torch.zeros(shape)


def _populate(root: nn.Module, schema) -> None:
    for name, (shape, kind) in schema.items():
        _attach(root, name, _init_like_reference(shape, kind), kind in ("bn_mean", "bn_var", "bn_count"))


class _HipEngine:
    """Weights blob + per-shape plans + workspace for one CenterNetDetection instance."""

    def __init__(self, precision: str, model_size: str):
        self.precision = precision
        self.model_size = model_size
        self.pw: Optional[P.PackedWeights] = None
        self.wdev: Optional[torch.Tensor] = None
        self.plans: Dict[Tuple, P.Plan] = {}
        self.workspace: Optional[torch.Tensor] = None

    def invalidate(self) -> None:
        self.pw, self.wdev = None, None
        self.release_plans()

    def release_plans(self) -> None:
        if self.plans:
            lib = L.load()
            for pl in self.plans.values():
                if pl.handle:
                    lib.ftc_plan_destroy(pl.handle)
                    pl.handle = None
        self.plans = {}

    def __del__(self):
        try:
            self.release_plans()
        except Exception:
            pass

    def ensure_weights(self, state_dict_fn, device) -> None:
        if self.pw is None:                       # only then: enumerating the 2444-tensor state dict costs ~8 ms of host time
            self.pw = P.pack_weights(state_dict_fn(), self.precision, self.model_size)
            self.wdev = None
        if self.wdev is None or self.wdev.device != device:
            self.wdev = torch.from_numpy(self.pw.blob).to(device)

    def get_plan(self, B, H, W, nchw) -> P.Plan:
        key = (B, H, W, nchw)
        pl = self.plans.get(key)
        if pl is None:
            pl = P.build_plan(self.pw, B, H, W, nchw)
            P.create_handle(pl, self.pw.nbytes)
            self.plans[key] = pl
        return pl

    def run(self, x: torch.Tensor, state_dict_fn, with_nms: bool = True):
        if not x.is_cuda:
            raise RuntimeError("findtextcenternet_amd: the detector runs on MI355X (gfx950) only -- move the module and "
                               "the input to 'cuda' (there is no CPU fallback)")
        lib = L.load()
        if x.dtype != torch.float32:
            x = x.float()
        B, Cc, H, W = x.shape
        if Cc != 3:
            raise ValueError("expected [B,3,H,W] input")
        if x.permute(0, 2, 3, 1).is_contiguous():
            nchw = False              # the callers' convention: NHWC memory behind an NCHW view
        elif x.is_contiguous():
            nchw = True
        else:
            x, nchw = x.contiguous(memory_format=torch.channels_last), False
        dev = x.device
        with torch.cuda.device(dev):
            self.ensure_weights(state_dict_fn, dev)
            pl = self.get_plan(B, H, W, nchw)
            if self.workspace is None or self.workspace.device != dev or self.workspace.numel() < pl.workspace_bytes:
                self.workspace = None
                self.workspace = torch.empty(pl.workspace_bytes, dtype=torch.uint8, device=dev)
            heat = torch.empty((B, pl.h, pl.w, 10), dtype=torch.float32, device=dev)
            feat = torch.empty((B, pl.h, pl.w, feature_dim), dtype=torch.float32, device=dev)
            bases = (C.c_void_p * L.NUM_BASES)(None, self.workspace.data_ptr(), self.wdev.data_ptr(), x.data_ptr(),
                                               heat.data_ptr(), feat.data_ptr())
            stream = torch.cuda.current_stream(dev).cuda_stream
            last = -1 if with_nms else len(pl.ops) - 2
            L.check(lib.ftc_plan_run(pl.handle, bases, C.c_void_p(stream), 0, last), "ftc_plan_run")
        # keep x alive until the work is enqueued on the same stream: it is (stream-ordered allocator)
        return heat, feat


class CenterNetDetection(nn.Module):
    """models/detector.py:203-230."""

    def __init__(self, pre_weights=True, model_size="xl", precision: Optional[str] = None, **kwargs) -> None:
        super().__init__(**kwargs)
        self.model_size = model_size
        _populate(self, detector_schema(model_size))
        prec = precision or os.environ.get("FTC_PRECISION", "fp32")
        if prec not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self._engine = _HipEngine(prec, model_size)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._engine.invalidate())
        # pre_weights: the reference looks for efficientnetv2-xl-21k.npz next to detector.py and
        # silently continues when it is missing (models/detector.py:34-36, :129-130); use
        # findtextcenternet_amd.weights.load_tf_efficientnetv2_npz() to import one explicitly.

    @property
    def precision(self) -> str:
        return self._engine.precision

    def set_precision(self, precision: str) -> None:
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        if precision != self._engine.precision:
            self._engine.invalidate()
            self._engine.precision = precision

    def _sd(self):
        return {k: v for k, v in self.state_dict().items()}

    def forward_nhwc(self, x, with_nms: bool):
        """(heat[B,h,w,10] fp32, feat[B,h,w,100] fp32) in NHWC memory; channel 1 is the NMS slot."""
        if self.training:
            raise NotImplementedError("findtextcenternet_amd implements the inference path (eval mode) only; call .eval()")
        return self._engine.run(x, self._sd, with_nms)

    def forward(self, x):
        heat, feat = self.forward_nhwc(x, with_nms=False)
        idx = torch.tensor([0, 2, 3, 4, 5, 6, 7, 8, 9], device=heat.device)
        return heat.index_select(3, idx).permute(0, 3, 1, 2), feat.permute(0, 3, 1, 2)


class SimpleDecoder(nn.Module):
    """Parameter container for models/detector.py:232-254 (checkpoint compatibility; the decoder
    belongs to the training / glyph-classification steps, outside this hot path)."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        _populate(self, decoder_schema())

    def forward(self, x):
        raise NotImplementedError("SimpleDecoder is outside the MI355X detector hot path")


class TextDetectorModel(nn.Module):
    """models/detector.py:256-281."""

    def __init__(self, pre_weights=True, model_size="xl", precision: Optional[str] = None, **kwargs) -> None:
        super().__init__(**kwargs)
        self.detector = CenterNetDetection(pre_weights=pre_weights, model_size=model_size, precision=precision)
        self.decoder = SimpleDecoder()

    def forward(self, x, fmask):
        raise NotImplementedError("training forward (train1.py) is outside the MI355X detector hot path")


class CenterNetDetector(nn.Module):
    """models/detector.py:283-296: detector + 3x3 max-pool NMS channel."""

    def __init__(self, detector, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        if not isinstance(detector, CenterNetDetection):
            raise TypeError("CenterNetDetector expects a findtextcenternet_amd CenterNetDetection")
        self.detector = detector
        self.minval = torch.tensor(float("-inf"))

    def forward_nhwc(self, x):
        return self.detector.forward_nhwc(x, with_nms=True)

    def forward(self, x):
        heat, feat = self.detector.forward_nhwc(x, with_nms=True)
        return heat.permute(0, 3, 1, 2), feat.permute(0, 3, 1, 2)
