"""Thin ctypes wrapper of the library-side model (``ftc_create`` ... ``ftc_destroy``, include/ftc.h).

All knowledge of the network -- graph, BatchNorm folding, weight packing, arena, kernel selection -- is in
``libftc_hip.so``; this module only marshals a ``state_dict`` into ``ftc_tensor`` records and exposes the plan the
library built (for the parity tests, the profiler harness and the tuner).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib as L


# numeric modes: fp32 = parity mode (exact-f32 MFMA); bf16 = speed mode (BASELINE config 2); fp16 = the speed-mode plan with IEEE-half
# operands (same MFMA rate, 3 more mantissa bits: ~8x closer to the fp32 result, activations saturate at +-65504)
# "fp16x3": the fp32 plan (fp32 tensors, weights, epilogues, accumulation) with every product on the 16-bit matrix pipe as three
# fp16 MFMAs of hi / lo split operands (FTC_FLAG_SPLIT16, include/ftc.h): the reference's fp32 tolerance at about twice the fp32 speed
PRECISIONS = {"fp32": L.F32, "bf16": L.BF16, "fp16": L.F16, "fp16x3": 3}
TORCH_DTYPE = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16, "fp16x3": torch.float32}


@dataclass
class OpMeta:
    name: str
    kind: str
    flops: float
    bytes: float


@dataclass
class PlanView:
    handle: int               # const ftc_plan*, borrowed from the model
    ops: "C.Array"            # copies of the ftc_op records
    meta: List[OpMeta]
    info: L.PlanInfo
    B: int
    H: int
    W: int

    @property
    def workspace_bytes(self) -> int:
        return int(self.info.workspace_bytes)

    @property
    def h(self) -> int:
        return int(self.info.map_h)

    @property
    def w(self) -> int:
        return int(self.info.map_w)


class FtcModel:
    """``ftc_create`` on a reference-style ``state_dict`` (``TextDetectorModel`` or ``CenterNetDetection`` keys)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], precision: str = "fp32", model_size: str = "xl"):
        if precision not in PRECISIONS:
            raise ValueError("precision must be 'fp32', 'fp16x3', 'bf16' or 'fp16'")
        lib = L.load()
        keep = []
        arr = (L.Tensor * len(state_dict))()
        for i, (k, v) in enumerate(state_dict.items()):
            t = v.detach()
            t = t.to(device="cpu", dtype=torch.float32).contiguous() if t.is_floating_point() else t.to(device="cpu").contiguous()
            kb = k.encode()
            keep.append((t, kb))
            arr[i].name, arr[i].data = kb, t.data_ptr()
            arr[i].dtype = L.F32 if t.dtype == torch.float32 else -1
            arr[i].ndim = t.dim()
            for j, d in enumerate(t.shape):
                arr[i].shape[j] = d
        h = C.c_void_p()
        L.check(lib.ftc_create(arr, len(state_dict), model_size.encode(), PRECISIONS[precision], C.byref(h)), "ftc_create")
        self.handle: Optional[int] = h.value
        self.precision, self.model_size = precision, model_size

    def close(self) -> None:
        if self.handle:
            L.load().ftc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def weights_bytes(self) -> int:
        return int(L.load().ftc_weights_bytes(self.handle))

    def weights_host(self) -> np.ndarray:
        """uint8 view of the packed blob (owned by the model: copy it or keep the model alive)."""
        lib = L.load()
        return np.ctypeslib.as_array(C.cast(lib.ftc_weights_host(self.handle), C.POINTER(C.c_uint8)), shape=(self.weights_bytes,))

    def offset(self, name: str) -> int:
        off = int(L.load().ftc_weights_offset(self.handle, name.encode()))
        if off < 0:
            raise KeyError(name)
        return off

    def workspace_bytes(self, B: int, H: int, W: int) -> int:
        n = int(L.load().ftc_workspace_bytes(self.handle, B, H, W))
        if n < 0:
            L.check(-1, "ftc_workspace_bytes")
        return n

    def plan(self, B: int, H: int, W: int, nchw: bool = False) -> PlanView:
        lib = L.load()
        info, hp = L.PlanInfo(), C.c_void_p()
        L.check(lib.ftc_model_plan(self.handle, B, H, W, 1 if nchw else 0, C.byref(hp), C.byref(info)), "ftc_model_plan")
        ops = (L.Op * info.n_ops)()
        meta = []
        oi = L.OpInfo()
        for i in range(info.n_ops):
            L.check(lib.ftc_plan_op(hp, i, C.byref(ops[i])), "ftc_plan_op")
            L.check(lib.ftc_model_op_info(self.handle, B, H, W, 1 if nchw else 0, i, C.byref(oi)), "ftc_model_op_info")
            meta.append(OpMeta(oi.name.decode(), oi.kind.decode(), float(oi.flops), float(oi.bytes)))
        return PlanView(hp.value, ops, meta, info, B, H, W)
