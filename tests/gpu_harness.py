"""Helpers for the -m gpu parity tests: run single ftc_op plans through the C ABI on real tensors."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from findtextcenternet_amd import _lib as L

REF_FIELDS = ("in_", "in2", "out", "w", "w2", "bias", "bias2", "scale", "shift", "aux", "out2")


def to_dev_bytes(t: torch.Tensor, dtype: int) -> torch.Tensor:
    """CPU fp32 tensor -> contiguous CPU tensor in the storage dtype."""
    t = t.contiguous()
    return t.to(torch.bfloat16) if dtype == L.BF16 else t.to(torch.float16) if dtype == L.F16 else t.float()


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).float()


def round16(t: torch.Tensor, dtype: int) -> torch.Tensor:
    """Round-trip through the 16-bit storage type (bf16 or IEEE half); identity for fp32."""
    return t.to(torch.bfloat16).float() if dtype == L.BF16 else t.to(torch.float16).float() if dtype == L.F16 else t


class Arena:
    """One device byte buffer holding every operand of a test op (all refs use FTC_BASE_WORKSPACE)."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.items = []      # (offset, cpu tensor or None, nbytes)
        self.size = 0

    def put(self, t: torch.Tensor) -> int:
        t = t.contiguous()
        off = self.size
        self.items.append((off, t))
        self.size = (self.size + t.numel() * t.element_size() + 255) // 256 * 256
        return off

    def reserve(self, nbytes: int) -> int:
        off = self.size
        self.items.append((off, None))
        self.size = (self.size + nbytes + 255) // 256 * 256
        return off

    def materialize(self, fill: int = 0xCD) -> torch.Tensor:
        buf = torch.full((self.size + 256,), fill, dtype=torch.uint8, device=self.device)
        for off, t in self.items:
            if t is not None:
                raw = t.view(torch.uint8).reshape(-1) if t.dtype not in (torch.bfloat16, torch.float16) else t.view(torch.int16).view(torch.uint8).reshape(-1)
                buf[off:off + raw.numel()] = raw.to(self.device)
        self.buf = buf
        return buf

    def read(self, off: int, shape, dtype: torch.dtype) -> torch.Tensor:
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        raw = self.buf[off:off + n].cpu()
        return raw.view(dtype).reshape(shape).clone()


def run_op(fields: Dict, arena: Arena) -> None:
    """fields: ftc_op int fields + ref fields given as arena offsets (or None)."""
    lib = L.load()
    op = (L.Op * 1)()
    for k, v in fields.items():
        if k in REF_FIELDS:
            if v is None:
                continue
            r = getattr(op[0], k)
            r.base, r.offset = L.BASE_WORKSPACE, int(v)
        else:
            setattr(op[0], k, int(v))
    h = C.c_void_p()
    L.check(lib.ftc_plan_create(op, 1, arena.size + 256, 0, C.byref(h)), "ftc_plan_create")
    try:
        bases = (C.c_void_p * L.NUM_BASES)(None, arena.buf.data_ptr(), None, None, None, None)
        stream = torch.cuda.current_stream().cuda_stream
        L.check(lib.ftc_plan_run(h, bases, C.c_void_p(stream), 0, -1), "ftc_plan_run")
        torch.cuda.synchronize()
    finally:
        lib.ftc_plan_destroy(h)


def tdtype(d: int) -> torch.dtype:
    return torch.float32 if d == L.F32 else torch.bfloat16 if d == L.BF16 else torch.float16


def presplit_f16x3(w: torch.Tensor) -> torch.Tensor:
    """fp32 K-major weights -> the storage FTC_FLAG_SPLIT16 expects: every 16-byte chunk of four fp32 values becomes
    [hi x4 | lo x4] IEEE halves (hi = fp16(x), lo = fp16(x - hi)); same byte size, returned as a flat uint8 tensor."""
    f = w.contiguous().float().reshape(-1, 4)
    hi = f.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = (f - hi.float()).to(torch.float16)
    return torch.cat([hi, lo], dim=1).contiguous().view(torch.uint8).reshape(-1)


# ---- shared models (the -m gpu suite) --------------------------------------------------------------------------------------------
# Building one XL model costs ~25 s of HOST time (seeded state_dict 9 s, random init of 262 M parameters 7 s, load 3 s, ftc_create's
# float64 fold + pack 5-9 s); round 5's suite built ~30 of them = 40 % of its 1021 s.  Tests that only READ a model share one per
# (precision, size) for the whole session; tests that edit parameters build their own through `fresh_model` (random init skipped:
# every tensor is overwritten by load_state_dict anyway).
_SHARED = {}


def fresh_model(precision, model_size="xl", seed=0, load=True):
    """A private TextDetectorModel holding deterministic_state_dict(seed); the constructor's random init is replaced by zeros."""
    import torch
    import findtextcenternet_amd.detector as D
    from findtextcenternet_amd import TextDetectorModel, deterministic_state_dict
    orig = D._init_like_reference

    def no_random(shape, kind):
        return torch.zeros(shape) if kind in ("conv", "conv_proj", "conv_dw", "se_w1", "se_w2", "conv_top", "linear") else orig(shape, kind)
    D._init_like_reference = no_random if load else orig
    try:
        m = TextDetectorModel(pre_weights=False, precision=precision, model_size=model_size)
    finally:
        D._init_like_reference = orig
    if load:
        m.load_state_dict(deterministic_state_dict(seed, model_size=model_size))
    return m


def shared_detector(precision, model_size="xl"):
    """(CenterNetDetector on cuda in eval mode, its TextDetectorModel) shared by every test of the session that does not edit it."""
    from findtextcenternet_amd import CenterNetDetector
    key = (precision, model_size)
    if key not in _SHARED:
        m = fresh_model(precision, model_size)
        d = CenterNetDetector(m.detector)
        d.to(device="cuda")
        d.eval()
        _SHARED[key] = (d, m)
    d, m = _SHARED[key]
    assert not d.training
    return d, m
