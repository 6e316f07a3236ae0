"""Mirror of the reference's ``loss_func.py`` (``/root/reference/loss_func.py``) for the detector's validation / training step,
forward only, on HIP kernels (``csrc/train_ops.hip``): same names, arguments and result keys.

* ``heatmap_loss(true, logits)``                              -- ``loss_func.py:74-92``
* ``loss_function(fmask, labelmap, idmap, heatmap, decoder_outputs)`` -- ``loss_func.py:94-177``; one fused pass over the maps
  (12 deterministic partial sums per workgroup) + one wave per selected pixel for the three CRT-modulo cross-entropies
* ``CoVWeightingLoss(device=..., losses=[...])``              -- ``loss_func.py:8-72``; state and step on the device, no host sync

No autograd: the values are what the reference prints / logs in its validation loop (``train1.py:218-244``); the backward
pass of the train step is not implemented.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L
from .schema import modulo_list

_KEYS = ["loss", "keymap_loss", "size_loss", "textline_loss", "separator_loss", "id_loss", "code1_loss", "code2_loss", "code4_loss",
         "code8_loss", "correct", "total"]


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _check_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("findtextcenternet_amd.loss_func runs on the GPU only (no CPU fallback)")


def _run_losses(labelmap, idmap, heatmap, decoder_outputs=None, sel_index=None, count=None) -> torch.Tensor:
    _check_cuda(labelmap, idmap, heatmap)
    lib = L.load()
    dev = heatmap.device
    B, nine, h, w = heatmap.shape
    if nine != 9 or tuple(labelmap.shape) != (B, 5, h, w) or tuple(idmap.shape) != (B, 2, h, w):
        raise ValueError("expected heatmap [B,9,h,w], labelmap [B,5,h,w], idmap [B,2,h,w]")
    hm = heatmap if heatmap.dtype == torch.float32 else heatmap.float()
    lab = labelmap.to(torch.float32).contiguous()
    ids = idmap.to(torch.int32).contiguous()                      # the reference carries int64 (train1.py:172); values fit int32
    strides = (C.c_int64 * 4)(*hm.stride())
    out = torch.zeros(16, dtype=torch.float32, device=dev)
    scratch = torch.empty(int(lib.ftc_losses_scratch_bytes()), dtype=torch.uint8, device=dev)
    dec_ptrs = [None, None, None]
    cap = 0
    if decoder_outputs is not None:
        decs = [d.to(torch.float32).contiguous() for d in decoder_outputs]
        if len(decs) != 3 or any(d.shape[1] != m for d, m in zip(decs, modulo_list)):
            raise ValueError("decoder_outputs must be three tensors [N, 1091 | 1093 | 1097]")
        cap = decs[0].shape[0]
        dec_ptrs = [d.data_ptr() for d in decs]
    with torch.cuda.device(dev):
        L.check(lib.ftc_losses(hm.data_ptr(), strides, lab.data_ptr(), ids.data_ptr(), B, h, w, dec_ptrs[0], dec_ptrs[1], dec_ptrs[2],
                               sel_index.data_ptr() if sel_index is not None else None, count.data_ptr() if count is not None else None,
                               cap, out.data_ptr(), scratch.data_ptr(), _stream(dev)), "ftc_losses")
    return out


def mask_to_index(fmask: torch.Tensor):
    """(ascending indices of the set entries [n] int32, count [1] int32) of a flat boolean mask, on the GPU."""
    _check_cuda(fmask)
    lib = L.load()
    m = fmask.reshape(-1).to(torch.uint8).contiguous()
    n = m.numel()
    sel = torch.empty(n, dtype=torch.int32, device=m.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=m.device)
    with torch.cuda.device(m.device):
        L.check(lib.ftc_mask_compact(m.data_ptr(), n, sel.data_ptr(), n, cnt.data_ptr(), _stream(m.device)), "ftc_mask_compact")
    return sel, cnt


def heatmap_loss(true: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """Penalty-reduced focal loss of the key heat-map (mean over all pixels), ``loss_func.py:74-92``."""
    B, h, w = true.shape
    dev = logits.device
    lab = torch.zeros((B, 5, h, w), dtype=torch.float32, device=dev)
    lab[:, 0] = true
    hm = torch.zeros((B, 9, h, w), dtype=torch.float32, device=dev)
    hm[:, 0] = logits
    out = _run_losses(lab, torch.zeros((B, 2, h, w), dtype=torch.int32, device=dev), hm)
    return out[1] / 10.0                                           # loss_function multiplies the mean by 10 (:114)


class _TrainLosses(torch.autograd.Function):
    """The loss op of the train plan as an autograd node (``TrainStep.seam_losses`` / ``seam_backward``): forward = the 16 values of FTC_OP_LOSSES
    on the labels given here; backward = the plan's whole backward half with d(objective)/d(loss_i) as the nine loss weights -- the parameter
    gradients are ADDED to ``.grad`` by the kernels themselves, nothing flows further back through autograd."""

    @staticmethod
    def forward(ctx, anchor, ts, fmask, labelmap, idmap):
        ctx.ts = ts
        return ts.seam_losses(fmask, labelmap, idmap)

    @staticmethod
    def backward(ctx, g):
        # 'loss' (entry 0) is the plain sum of the nine (loss_func.py:162-164): its gradient spreads over them; 'correct' / 'total' are counts
        w = g[1:10] + g[0]
        ctx.ts.seam_backward(w.detach())
        return None, None, None, None, None


class _WeightedSum(torch.autograd.Function):
    """value = what ftc_cov_weighting_step computed (sum_i alpha_i loss_i); d value / d loss_i = alpha_i (the alphas are detached weights,
    loss_func.py:69-71)."""

    @staticmethod
    def forward(ctx, vals, alphas, value):
        ctx.save_for_backward(alphas)
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        (alphas,) = ctx.saved_tensors
        return g * alphas, None, None


@torch.compiler.disable
def loss_function(fmask: torch.Tensor, labelmap: torch.Tensor, idmap: torch.Tensor, heatmap: torch.Tensor,
                  decoder_outputs: Sequence[torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``loss_func.py:94-177``: dict with the reference's keys; values are 0-d fp32 tensors on the device (``correct`` / ``total``
    are counts, as in the reference).  On the outputs of a train-mode ``model(image, fmask)`` call with gradients enabled (train1.py:125-131)
    the values carry the autograd node that runs the backward half of the train plan."""
    ts = getattr(heatmap, "_ftc_train_step", None)
    if ts is not None and torch.is_grad_enabled():
        out = _TrainLosses.apply(heatmap, ts, fmask, labelmap, idmap)
        return {k: out[i] for i, k in enumerate(_KEYS)}
    sel, cnt = mask_to_index(fmask)
    n_rows = decoder_outputs[0].shape[0]
    out = _run_losses(labelmap, idmap, heatmap, decoder_outputs, sel[:max(1, n_rows)].contiguous(), cnt)
    return {k: out[i] for i, k in enumerate(_KEYS)}


class CoVWeightingLoss(torch.nn.Module):
    """Multi-loss weighting with coefficient of variations, ``loss_func.py:8-72`` (state: Welford running statistics of the loss
    values and of their ratios to the running mean).  Note the reference's ``if not self.train:`` tests a bound method, so its
    weighted branch is taken in validation too; mirrored here."""

    def __init__(self, *args, **kwargs) -> None:
        self.device = kwargs.pop("device", "cuda")
        self.losses: List[str] = kwargs.pop("losses", [])
        self.num_losses = len(self.losses)
        super().__init__(*args, **kwargs)
        if not 0 < self.num_losses <= 16:
            raise ValueError("CoVWeightingLoss handles 1..16 losses")
        self.current_iter = -1
        self._state = torch.zeros(80, dtype=torch.float32, device=self.device)

    @property
    def alphas(self) -> torch.Tensor:
        return self._state[64:64 + self.num_losses]

    @property
    def running_mean_L(self) -> torch.Tensor:
        return self._state[0:self.num_losses]

    @property
    def running_mean_l(self) -> torch.Tensor:
        return self._state[16:16 + self.num_losses]

    @property
    def running_std_l(self) -> Optional[torch.Tensor]:
        return self._state[48:48 + self.num_losses] if self.current_iter >= 0 else None

    @torch.compiler.disable
    def forward(self, losses: Dict[str, torch.Tensor]) -> torch.Tensor:
        if torch.is_grad_enabled() and any(losses[k].requires_grad for k in self.losses):
            stacked = torch.stack([losses[k].to(torch.float32).reshape(()) for k in self.losses])
            with torch.no_grad():
                value = self.forward({k: losses[k].detach() for k in self.losses})
            return _WeightedSum.apply(stacked, self.alphas.detach().clone(), value)
        lib = L.load()
        vals = torch.stack([losses[k].detach().to(torch.float32).reshape(()) for k in self.losses]).to(self._state.device).contiguous()
        _check_cuda(vals)
        self.current_iter += 1
        out = torch.empty(1, dtype=torch.float32, device=vals.device)
        with torch.cuda.device(vals.device):
            L.check(lib.ftc_cov_weighting_step(vals.data_ptr(), self.num_losses, self.current_iter, self._state.data_ptr(), out.data_ptr(),
                                               _stream(vals.device)), "ftc_cov_weighting_step")
        return out[0]
