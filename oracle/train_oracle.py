"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the reference's TRAIN step (forward in train() mode + loss + backward).

Restates ``/root/reference/train1.py:125-131, 170-179`` on top of the functional restatements in ``detector_oracle.py`` (train()-mode
forward: batch-statistics BatchNorm, StochasticDepth with a supplied draw) and ``loss_oracle.py`` (``loss_function``,
``/root/reference/loss_func.py:94-177``); the gradients come from torch autograd on the CPU in fp32, which is exactly how the reference
obtains them (``loss.backward()``), minus its bf16 autocast.  The weighted loss is ``sum_i alpha_i * loss_i`` with the alphas as
constants (``CoVWeightingLoss`` detaches them, ``loss_func.py:69-71``).

Pinned by ``tests/golden/g10_train_step.npz``, written by the reference's own modules (``tests/golden/gen_golden.py::gen_train_step``).
Only ``tests/`` and ``__graft_entry__.smoke()`` may import this.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import detector_oracle, loss_oracle

COV_KEYS = ["keymap_loss", "size_loss", "textline_loss", "separator_loss", "id_loss", "code1_loss", "code2_loss", "code4_loss", "code8_loss"]   # train1.py:107-114


def is_parameter(key: str) -> bool:
    return not (key.endswith("running_mean") or key.endswith("running_var") or key.endswith("num_batches_tracked"))


def train_step(sd: Dict[str, torch.Tensor], x: torch.Tensor, labelmap: torch.Tensor, idmap: torch.Tensor, keep: Optional[Dict[str, torch.Tensor]] = None,
               alphas: Optional[Sequence[float]] = None, loss_scale: float = 1.0):
    """sd: TextDetectorModel state dict ("detector." / "decoder." keys); x [B,3,H,W] 0..1.  Returns (loss, raw losses, {param: grad},
    maps [B,9,h,w])."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if is_parameter(k) and v.is_floating_point()}
    full = dict(sd)
    full.update(leaves)
    with torch.enable_grad():
        maps, feats, _ = detector_oracle.detection_forward_train.__wrapped__(full, x, keep)
        fmask = loss_oracle.get_fmask(labelmap)
        rows = feats.permute(0, 2, 3, 1).flatten(0, -2)[fmask]
        dec, _ = detector_oracle.decoder_forward_train.__wrapped__(full, rows)
        raw = loss_oracle.loss_function(fmask, labelmap, idmap, maps, dec)
        a = [1.0 / len(COV_KEYS)] * len(COV_KEYS) if alphas is None else [float(v) for v in alphas]
        loss = sum(ai * raw[k] for ai, k in zip(a, COV_KEYS))
        (loss * loss_scale).backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return loss.detach(), {k: v.detach() for k, v in raw.items()}, grads, maps.detach()


def train_step_decoder_only(sd: Dict[str, torch.Tensor], x: torch.Tensor, labelmap: torch.Tensor, idmap: torch.Tensor,
                            alphas: Optional[Sequence[float]] = None, loss_scale: float = 1.0):
    """The reference's ``decoder_only`` step (``/root/reference/train1.py:98-101, 163-164``): detector parameters frozen, detector in eval
    mode (running statistics, no StochasticDepth), SimpleDecoder in train mode.  Returns (loss, raw losses, {decoder param: grad},
    {decoder running-stat key: new value})."""
    det = {k[len("detector."):]: v for k, v in sd.items() if k.startswith("detector.")}
    with torch.no_grad():
        maps, feats = detector_oracle.detection_forward(det, x)
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("decoder.") and is_parameter(k) and v.is_floating_point()}
    full = dict(sd)
    full.update(leaves)
    with torch.enable_grad():
        fmask = loss_oracle.get_fmask(labelmap)
        rows = feats.permute(0, 2, 3, 1).flatten(0, -2)[fmask]
        dec, new_stats = detector_oracle.decoder_forward_train.__wrapped__(full, rows)
        raw = loss_oracle.loss_function(fmask, labelmap, idmap, maps, dec)
        a = [1.0 / len(COV_KEYS)] * len(COV_KEYS) if alphas is None else [float(v) for v in alphas]
        loss = sum(ai * raw[k] for ai, k in zip(a, COV_KEYS))
        (loss * loss_scale).backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return loss.detach(), {k: v.detach() for k, v in raw.items()}, grads, {k: v.detach() for k, v in new_stats.items()}
