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
``"fp32"`` (default; exact-f32 MFMA, the parity mode -- what the reference computes on CUDA/CPU),
``"bf16"`` (bf16 MFMA with fp32 accumulation, fp32 residual trunk and fp32 outputs -- the speed mode
BASELINE.json's config 2 names) and ``"fp16"`` (the same plan with IEEE-half operands: same matrix rate,
11-bit significands, so about 8x closer to the fp32 result; 16-bit activations saturate at +-65504).
"""
from __future__ import annotations

import itertools

import ctypes as C
import math
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib as L
from .model import PRECISIONS, TORCH_DTYPE, FtcModel
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


_INIT_LIKE_REFERENCE = _init_like_reference


def _init_normal_np(shape, kind: str, seed: int) -> torch.Tensor:
    """The conv kinds of _init_like_reference from a per-tensor numpy generator (fills release the GIL: _populate runs them on threads)."""
    fan_out = shape[0] * shape[2] * shape[3]
    a = np.random.Generator(np.random.SFC64(seed)).standard_normal(shape, dtype=np.float32)
    a *= np.float32(math.sqrt(2.0 / fan_out))
    return torch.from_numpy(a)


def _populate(root: nn.Module, schema) -> None:
    # 262 M parameters for "xl": torch.randn on one thread takes ~7 s.  The large tensors (the conv kinds) are drawn by per-tensor numpy
    # generators on a few threads instead, seeded from ONE draw of torch's global generator (torch.manual_seed still fixes the init).
    from concurrent.futures import ThreadPoolExecutor
    if _init_like_reference is not _INIT_LIKE_REFERENCE:              # (tests replace the initialiser: keep the plain path)
        for name, (shape, kind) in schema.items():
            _attach(root, name, _init_like_reference(shape, kind), kind in ("bn_mean", "bn_var", "bn_count"))
        return
    base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    items = list(schema.items())
    big = [(i, shape, kind) for i, (_, (shape, kind)) in enumerate(items) if kind in ("conv", "conv_proj", "conv_dw", "se_w1", "se_w2")]
    drawn = {}
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for (i, _, _), t in zip(big, ex.map(lambda a: _init_normal_np(a[1], a[2], base + a[0]), big)):
            drawn[i] = t
    for i, (name, (shape, kind)) in enumerate(items):
        _attach(root, name, drawn[i] if i in drawn else _init_like_reference(shape, kind), kind in ("bn_mean", "bn_var", "bn_count"))


class _HipEngine:
    """The library-side model (``ftc_create``: folded + packed weights, per-shape plans) of one CenterNetDetection instance, its
    weight blob in HBM and the activation workspace.  All network knowledge lives in the C library; this class only owns
    device memory and notices when the module's parameters change."""

    def __init__(self, precision: str, model_size: str, module: nn.Module):
        self.precision = precision
        self.model_size = model_size
        self.module = module                     # whose state_dict is packed: a CenterNetDetection, or the TextDetectorModel that owns it
        self.model: Optional[FtcModel] = None
        self.wdev: Optional[torch.Tensor] = None
        self.workspace: Optional[torch.Tensor] = None
        self.fingerprint = None

    def invalidate(self) -> None:
        """Forces a re-pack on the next forward.  Needed by hand only after writes the fingerprint cannot see: in-place edits through
        ``p.data`` (``p.data.copy_()``, ``p.data.mul_()`` ...: ``.data`` carries its own version counter) or raw-pointer writes."""
        if self.model is not None:
            self.model.close()
        self.model, self.wdev, self.fingerprint = None, None, None
        self._bump()

    def _bump(self) -> None:
        """The packed weights or the arena moved: captured graphs hold their addresses -- drop them (never replay one whose memory may
        have gone back to the allocator) and start a new generation (object ids and device addresses get reused, a counter does not)."""
        self.generation = getattr(self, "generation", 0) + 1
        for ent in self.__dict__.get("_graphs", {}).values():
            ent["graph"] = None

    def __del__(self):
        try:
            self.invalidate()
        except Exception:
            pass

    def _fingerprint(self):
        module = self.module
        # Parameters and buffers can change behind our back: in-place writes on the tensor itself (p.add_(), optimizer.step() --
        # findtextcenternet_amd's AdamWScheduleFree and TrainStep bump the counters of what their kernels wrote --, the schedule-free
        # train()/eval() swap) bump Tensor._version; re-allocations (p.data = ..., .half().float()) move data_ptr; replaced Parameter
        # objects (load_state_dict(assign=True), m.backbone = ..., parametrizations) change id().  The tensors are re-enumerated on
        # every call (~2 ms of host time for the 2400 tensors, hidden behind the previous forward's GPU work).  NOT seen: in-place
        # writes through `p.data` (its own version counter) -- call invalidate() after those.
        # The fingerprint is the exact per-tensor record (object, address, version), compared as a tuple: sums over the tensors (rounds 3-4)
        # let two offsetting changes cancel.
        return tuple((id(t), t.data_ptr(), t._version) for t in itertools.chain(module.parameters(), module.buffers()))

    def ensure_model(self, device) -> None:
        fp = self._fingerprint()
        if self.model is None or fp != self.fingerprint:
            self.invalidate()
            self.model = FtcModel(self.module.state_dict(), self.precision, self.model_size)      # folds + packs in the library (seconds)
            self.fingerprint = self._fingerprint()
        if self.wdev is None or self.wdev.device != device:
            self.wdev = torch.from_numpy(self.model.weights_host()).to(device)              # one H2D copy of the packed blob
            self._bump()

    @property
    def handle(self):
        return self.model.handle if self.model is not None else None

    def plan(self, B, H, W, nchw=False):
        return self.model.plan(B, H, W, nchw)

    def ensure_workspace(self, B, H, W, device) -> None:
        need = self.model.workspace_bytes(B, H, W)
        if self.workspace is None or self.workspace.device != device or self.workspace.numel() < need:
            self.workspace = None
            self.workspace = torch.empty(need, dtype=torch.uint8, device=device)
            self._bump()

    def run(self, x: torch.Tensor, with_nms: bool = True, out=None, workspace: Optional[torch.Tensor] = None):
        """workspace: a caller-owned activation arena (>= model.workspace_bytes(B, H, W) bytes) instead of the engine's own -- what lets
        several batches be in flight on several streams at once (findtextcenternet_amd.lanes.DetectorLanes)."""
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
        if H % 32 or W % 32:
            raise ValueError("H and W must be multiples of 32 (the reference always uses 768)")
        dev = x.device
        # One or two tiles per call and nothing supplied by the caller (the reference's loops: `heatmap, features = detector(images)`
        # tile by tile, process_ocr_torch.py:43-49, test_image1_torch.py): 368 launches of 5-15 us kernels are launch-bound, so the forward
        # is replayed from a HIP graph (see _run_graph).  FTC_NO_GRAPH=1: always eager.
        if (out is None and workspace is None and B * H * W <= 2 * 768 * 768 and os.environ.get("FTC_NO_GRAPH") != "1"
                and not torch.cuda.is_current_stream_capturing()):
            return self._run_graph(x, B, H, W, nchw, with_nms, dev)
        with torch.cuda.device(dev):
            self.ensure_model(dev)
            if workspace is None:
                self.ensure_workspace(B, H, W, dev)
                workspace = self.workspace
            elif workspace.device != dev or workspace.numel() * workspace.element_size() < self.model.workspace_bytes(B, H, W):
                raise ValueError("workspace= is smaller than model.workspace_bytes(B, H, W) or on another device")
            h, w = H // 4, W // 4
            if out is None:
                heat = torch.empty((B, h, w, 10), dtype=torch.float32, device=dev)
                feat = torch.empty((B, h, w, feature_dim), dtype=torch.float32, device=dev)
            else:
                heat, feat = out
                if (tuple(heat.shape) != (B, h, w, 10) or tuple(feat.shape) != (B, h, w, feature_dim) or heat.dtype != torch.float32
                        or feat.dtype != torch.float32 or not heat.is_contiguous() or not feat.is_contiguous() or heat.device != dev):
                    raise ValueError("out=(heat [B,h,w,10], feat [B,h,w,100]) must be contiguous fp32 tensors on the input's device")
            stream = torch.cuda.current_stream(dev).cuda_stream
            L.check(lib.ftc_forward(self.handle, self.wdev.data_ptr(), x.data_ptr(), B, H, W, 1 if nchw else 0, 1 if with_nms else 0,
                                    heat.data_ptr(), feat.data_ptr(), workspace.data_ptr(), C.c_void_p(stream)), "ftc_forward")
        # x stays alive until the work is enqueued on the same stream (stream-ordered allocator)
        return heat, feat


    def _run_graph(self, x, B, H, W, nchw, with_nms, dev):
        """Static input / output buffers per (shape, layout), the forward captured on the second call and replayed afterwards; the
        replay is launched BEFORE the host walk that checks whether a parameter changed (1.8 ms for 1376 tensors: it overlaps the GPU
        work), and if something did change the eager forward that follows overwrites the result on the same stream.  Fresh output
        tensors are returned on every call (two device copies of 16 MB in total: ~10 us), as a module call must."""
        lib = L.load()
        graphs = self.__dict__.setdefault("_graphs", {})
        k = (B, H, W, nchw, with_nms, dev)
        ent = graphs.get(k)
        h, w = H // 4, W // 4
        with torch.cuda.device(dev):
            if ent is None:
                if len(graphs) >= 4:
                    graphs.clear()                                   # (a caller cycling through many shapes: start over)
                xs = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev) if nchw else torch.empty((B, H, W, 3), dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
                ent = {"x": xs, "heat": torch.empty((B, h, w, 10), dtype=torch.float32, device=dev),
                       "feat": torch.empty((B, h, w, feature_dim), dtype=torch.float32, device=dev), "graph": None, "gen": None, "warm": None}
                graphs[k] = ent
            ent["x"].copy_(x)
            replayed = ent["graph"] is not None
            if replayed:
                ent["graph"].replay()
            self.ensure_model(dev)
            self.ensure_workspace(B, H, W, dev)
            gen = self.generation

            def launch():
                stream = torch.cuda.current_stream(dev).cuda_stream
                L.check(lib.ftc_forward(self.handle, self.wdev.data_ptr(), ent["x"].data_ptr(), B, H, W, 1 if nchw else 0, 1 if with_nms else 0,
                                        ent["heat"].data_ptr(), ent["feat"].data_ptr(), self.workspace.data_ptr(), C.c_void_p(stream)), "ftc_forward")
            if not (replayed and ent["gen"] == gen):
                ent["graph"] = None
                launch()
                if ent["warm"] == gen:                                 # second call in this state: every kernel has run once -> capture
                    try:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g):
                            launch()
                        ent["graph"], ent["gen"] = g, gen
                    except Exception:
                        os.environ["FTC_NO_GRAPH"] = "1"               # (capture unsupported here: stay eager)
                ent["warm"] = gen
            return ent["heat"].clone(), ent["feat"].clone()


class CenterNetDetection(nn.Module):
    """models/detector.py:203-230."""

    def __init__(self, pre_weights=True, model_size="xl", precision: Optional[str] = None, **kwargs) -> None:
        super().__init__(**kwargs)
        self.model_size = model_size
        _populate(self, detector_schema(model_size))
        prec = precision or os.environ.get("FTC_PRECISION", "fp32")
        if prec not in PRECISIONS:
            raise ValueError("precision must be 'fp32', 'fp16x3', 'bf16' or 'fp16'")
        object.__setattr__(self, "_engine", _HipEngine(prec, model_size, self))
        # pre_weights: the reference looks for efficientnetv2-xl-21k.npz next to detector.py and
        # silently continues when it is missing (models/detector.py:34-36, :129-130); use
        # findtextcenternet_amd.weights.load_tf_efficientnetv2_npz() to import one explicitly.

    @property
    def precision(self) -> str:
        return self._engine.precision

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self._engine.invalidate()                        # a new checkpoint always re-packs (also with assign=True)
        return r

    def set_precision(self, precision: str) -> None:
        if precision not in PRECISIONS:
            raise ValueError("precision must be 'fp32', 'fp16x3', 'bf16' or 'fp16'")
        if precision != self._engine.precision:
            self._engine.invalidate()
            self._engine.precision = precision

    # the engine holds C handles and device memory: a copy / pickle of the module gets a fresh one
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_engine"] = self._engine.precision           # only the numeric mode survives a pickle
        return st

    def __setstate__(self, st):
        prec = st.pop("_engine", None) or os.environ.get("FTC_PRECISION", "fp32")
        super().__setstate__(st)
        object.__setattr__(self, "_engine", _HipEngine(prec, self.model_size, self))

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_engine":
                new.__dict__[k] = copy.deepcopy(v, memo)
        object.__setattr__(new, "_engine", _HipEngine(self._engine.precision, self.model_size, new))
        return new

    def forward_nhwc(self, x, with_nms: bool, out=None, workspace=None):
        """(heat[B,h,w,10] fp32, feat[B,h,w,100] fp32) in NHWC memory; channel 1 is the NMS slot.  ``out=(heat, feat)``
        writes into caller-owned tensors instead of allocating; ``workspace`` = a caller-owned activation arena."""
        if self.training:
            raise NotImplementedError("findtextcenternet_amd implements the inference path (eval mode) only; call .eval()")
        return self._engine.run(x, with_nms, out, workspace)

    def forward(self, x):
        heat, feat = self.forward_nhwc(x, with_nms=False)
        idx = torch.tensor([0, 2, 3, 4, 5, 6, 7, 8, 9], device=heat.device)
        return heat.index_select(3, idx).permute(0, 3, 1, 2), feat.permute(0, 3, 1, 2)


class SimpleDecoder(nn.Module):
    """models/detector.py:232-254: three MLPs 100 -> 2048 -> 2048 -> {1091, 1093, 1097} (BatchNorm1d + GELU between the Linear
    layers) that classify a glyph's 100-d feature into the CRT residues of its code point.  Parameter container with the
    reference's keys (``blocks.<i>.{0,3,6}.weight`` ...); ``forward`` (eval mode) runs inside the library: BatchNorm1d folded into the
    Linear before it, the Linear layers as 1x1 implicit GEMMs on the MFMA conv kernel with fused bias + exact GELU
    (``ftc_decoder_forward``).  Called through ``TextDetectorModel`` (which owns the packed weights)."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        _populate(self, decoder_schema())
        object.__setattr__(self, "_owner", None)

    def forward(self, x):
        owner = self.__dict__.get("_owner")
        if owner is None:
            raise NotImplementedError("SimpleDecoder runs as part of a TextDetectorModel (it shares the model's packed weight blob)")
        return owner._decode_rows(x)


class TextDetectorModel(nn.Module):
    """models/detector.py:256-281.  ``forward(x, fmask)`` and ``get_fmask`` implement the reference's eval-mode (validation) step
    on the GPU; training-mode forward (batch-statistics BatchNorm, stochastic depth) and backward are not implemented."""

    def __init__(self, pre_weights=True, model_size="xl", precision: Optional[str] = None, **kwargs) -> None:
        super().__init__(**kwargs)
        self.detector = CenterNetDetection(pre_weights=pre_weights, model_size=model_size, precision=precision)
        self.decoder = SimpleDecoder()
        self._bind()

    def _bind(self) -> None:
        # ONE engine for the whole model: detector and decoder weights live in one packed blob (ftc_create accepts the
        # "detector." / "decoder." prefixed keys of model.pt as they are); the detector module shares it.
        eng = _HipEngine(self.detector.precision, self.detector.model_size, self)
        old = self.detector.__dict__.get("_engine")
        if old is not None:
            old.invalidate()
        object.__setattr__(self.detector, "_engine", eng)
        object.__setattr__(self, "_engine", eng)
        object.__setattr__(self.decoder, "_owner", self)

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self._engine.invalidate()                        # a new checkpoint always re-packs (also with assign=True)
        return r

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_engine"] = None
        st.pop("_train_forward", None)
        return st

    def __setstate__(self, st):
        st.pop("_engine", None)
        super().__setstate__(st)
        self._bind()

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in ("_engine", "_train_forward"):
                new.__dict__[k] = copy.deepcopy(v, memo)
        new._bind()
        return new

    def _decode_rows(self, feats: torch.Tensor, rows_ready: bool = False):
        """feats [N,100] fp32 (or, rows_ready, [N,128] already in the compute dtype) -> three [N, modulo] fp32 tensors."""
        if self.training:
            raise NotImplementedError("findtextcenternet_amd implements the eval-mode forward only; call .eval()")
        if not feats.is_cuda:
            raise RuntimeError("findtextcenternet_amd: the decoder runs on MI355X (gfx950) only (there is no CPU fallback)")
        lib = L.load()
        dev = feats.device
        eng = self._engine
        with torch.cuda.device(dev):
            eng.ensure_model(dev)
            n = feats.shape[0]
            cdt = TORCH_DTYPE[eng.precision]
            if rows_ready:
                rows = feats
            else:
                rows = torch.zeros((n, 128), dtype=cdt, device=dev)
                rows[:, :feature_dim] = feats.to(cdt)
            need = int(lib.ftc_decoder_workspace_bytes(eng.handle, n))
            if need < 0:
                L.check(-1, "ftc_decoder_workspace_bytes")
            if eng.workspace is None or eng.workspace.device != dev or eng.workspace.numel() < need:
                eng.workspace = torch.empty(need, dtype=torch.uint8, device=dev)
            outs = [torch.empty((n, m), dtype=torch.float32, device=dev) for m in (1091, 1093, 1097)]
            L.check(lib.ftc_decoder_forward(eng.handle, eng.wdev.data_ptr(), rows.data_ptr(), n, outs[0].data_ptr(), outs[1].data_ptr(),
                                            outs[2].data_ptr(), eng.workspace.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                    "ftc_decoder_forward")
        return outs

    @torch.compiler.disable
    def forward(self, x, fmask):
        """(heatmap [B,9,h,w], [dec0, dec1, dec2]) -- detector forward, boolean-mask gather of the flattened NHWC feature map
        (``features.permute(0,2,3,1).flatten(0,-2)[fmask]``, models/detector.py:265-266) on the GPU, decoder on the gathered rows."""
        if self.training:
            # train() mode: batch-statistics BatchNorm + StochasticDepth, running statistics updated -- forward only (the reference's
            # BN-refresh pass, train1.py:203-211, runs exactly this under torch.no_grad()).  There is no backward pass.
            if torch.is_grad_enabled():
                # The reference's train step exactly as train1.py:125-131 writes it: the SAME static plan TrainStep.forward_backward runs, cut at
                # the loss op -- loss_function(...) runs the loss op, CoVWeightingLoss weights it, loss.backward() runs the plan's backward half
                # and ADDS the parameter gradients into .grad (findtextcenternet_amd.train_step: "the reference's own calling sequence").
                if fmask is None:
                    raise ValueError("train-mode forward with gradients enabled is the train step's forward (train1.py:125-131): pass fmask = "
                                     "model.get_fmask(labelmap, fmask); for the BatchNorm-refresh pass wrap the call in torch.no_grad()")
                from .train_step import TrainStep
                ts = self.__dict__.get("_train_step")
                if ts is None or ts.precision != self.detector.precision or ts.decoder_only != (not self.detector.training):
                    ts = TrainStep(self, self.detector.precision, decoder_only=not self.detector.training)
                    self.__dict__["_train_step"] = ts
                heat, decs = ts.seam_forward(x, fmask)
                heat.requires_grad_(True)                      # (the anchor of the autograd node loss_function attaches; its own gradient is never formed)
                heat._ftc_train_step = ts
                return heat, decs
            from .train_forward import TrainForward
            tf = self.__dict__.get("_train_forward")
            if tf is None or tf.precision != self.detector.precision:
                tf = TrainForward(self, self.detector.precision)
                self.__dict__["_train_forward"] = tf
            return tf.forward(x, fmask, keep=self.__dict__.get("stochastic_depth_keep"))
        from .loss_func import mask_to_index
        lib = L.load()
        heat, feat = self.detector.forward_nhwc(x, with_nms=False)
        dev = heat.device
        sel, cnt = mask_to_index(fmask)
        n = int(cnt.item())                                           # the reference's boolean indexing synchronises here as well
        cdt = TORCH_DTYPE[self.detector.precision]
        rows = torch.empty((max(n, 1), 128), dtype=cdt, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.ftc_gather_rows(feat.data_ptr(), sel.data_ptr(), cnt.data_ptr(), max(n, 1), feature_dim, 128, rows.data_ptr(),
                                        min(PRECISIONS[self.detector.precision], 2) if cdt != torch.float32 else L.F32,
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "ftc_gather_rows")
        idx = torch.tensor([0, 2, 3, 4, 5, 6, 7, 8, 9], device=dev)
        heatmap = heat.index_select(3, idx).permute(0, 3, 1, 2)
        if n == 0:
            return heatmap, [torch.empty((0, m), dtype=torch.float32, device=dev) for m in (1091, 1093, 1097)]
        return heatmap, self._decode_rows(rows[:n], rows_ready=True)

    def get_fmask(self, heatmap, mask):
        """models/detector.py:270-281: boolean mask over the flattened [B,h,w] key-label map marking its 1024*B largest entries
        (ties at the boundary: lowest index first, what the reference's stable CPU sort gives).  ``heatmap`` is the LABEL map
        in the reference's training loop (train1.py:174)."""
        if not heatmap.is_cuda:
            raise RuntimeError("findtextcenternet_amd: get_fmask runs on the GPU only (there is no CPU fallback)")
        lib = L.load()
        B = heatmap.shape[0]
        vals = heatmap[:, 0, :, :].to(torch.float32).contiguous().reshape(-1)
        n = vals.numel()
        if mask is None or mask.shape != vals.shape or mask.dtype != torch.bool or mask.device != vals.device:
            mask = torch.zeros(n, dtype=torch.bool, device=vals.device)
        with torch.cuda.device(vals.device):
            L.check(lib.ftc_topk_mask(vals.data_ptr(), n, 1024 * B, mask.data_ptr(), None, None,
                                      C.c_void_p(torch.cuda.current_stream(vals.device).cuda_stream)), "ftc_topk_mask")
        return mask


class CenterNetDetector(nn.Module):
    """models/detector.py:283-296: detector + 3x3 max-pool NMS channel."""

    def __init__(self, detector, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        if not isinstance(detector, CenterNetDetection):
            raise TypeError("CenterNetDetector expects a findtextcenternet_amd CenterNetDetection")
        self.detector = detector
        self.minval = torch.tensor(float("-inf"))

    def forward_nhwc(self, x, out=None, workspace=None):
        return self.detector.forward_nhwc(x, with_nms=True, out=out, workspace=workspace)

    def forward(self, x):
        heat, feat = self.detector.forward_nhwc(x, with_nms=True)
        return heat.permute(0, 3, 1, 2), feat.permute(0, 3, 1, 2)
