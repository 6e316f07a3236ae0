"""Schedule-Free AdamW on the MI355X path (BASELINE config 5).

Mirror of ``AdamWScheduleFree`` of the reference (``/root/reference/models/adamw_schedulefree.py``; the reference vendors
it from facebookresearch/schedule_free): same constructor arguments, ``param_groups`` keys, per-parameter state
(``z``, ``exp_avg_sq``) and the ``train()`` / ``eval()`` protocol, so ``train1.py:96-101, 186-211`` can construct and drive
it unchanged.  ``step()`` is one HIP launch over all parameters of a group (``ftc_adamw_schedulefree_step``) instead of ten
``torch._foreach_*`` passes; parameters and gradients must be fp32 CUDA tensors (there is no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _lib as L

CHUNK = 4096          # elements per workgroup of the update kernel


def bump_versions(tensors) -> None:
    """Marks tensors that a HIP kernel wrote through raw pointers as modified (Tensor._version + 1), as an in-place ATen op would:
    the detector engine re-packs its weight blob when the counters of the module's parameters / buffers move."""
    ts = list(tensors)
    if ts:
        torch._C._autograd._unsafe_set_version_counter(ts, [t._version + 1 for t in ts])


def step_scalars(group: dict) -> Dict[str, float]:
    """The per-step scalars of one parameter group, in float64 exactly as the reference derives them
    (adamw_schedulefree.py:124-147), and the group's bookkeeping updated in place (scheduled_lr, lr_max, weight_sum)."""
    k = group["k"]
    beta1, beta2 = group["betas"]
    warm = group["warmup_steps"]
    lr = group["lr"] * ((k + 1) / warm if k < warm else 1.0)
    group["scheduled_lr"] = lr
    lr_max = group["lr_max"] = max(lr, group["lr_max"])
    weight = ((k + 1) ** group["r"]) * (lr_max ** group["weight_lr_power"])
    weight_sum = group["weight_sum"] = group["weight_sum"] + weight
    ckp1 = weight / weight_sum if weight_sum != 0 else 0
    return {"beta2": beta2, "one_minus_beta2": 1 - beta2, "bias_correction2": 1 - beta2 ** (k + 1), "eps": group["eps"],
            "weight_decay": group["weight_decay"], "ckp1": ckp1, "y_alpha": lr * (beta1 * (1 - ckp1) - 1), "lr": lr}


class AdamWScheduleFree(torch.optim.Optimizer):
    def __init__(self, params, lr=0.0025, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0,
                 warmup_steps: int = 0, r: float = 0.0, weight_lr_power: float = 2.0, foreach: Optional[bool] = True):
        defaults = dict(lr=lr, betas=betas, eps=eps, r=r, k=0, warmup_steps=warmup_steps, train_mode=False, weight_sum=0.0,
                        lr_max=-1.0, scheduled_lr=0.0, weight_lr_power=weight_lr_power, weight_decay=weight_decay, foreach=foreach)
        super().__init__(params, defaults)

    # x = the averaged iterate (evaluation / checkpoints), y = where gradients are taken (training): p holds one or the other
    @torch.no_grad()
    def _swap(self, to_train: bool) -> None:
        for group in self.param_groups:
            if group["train_mode"] == to_train:
                continue
            beta1 = group["betas"][0]
            w = 1 - beta1 if to_train else 1 - 1 / beta1
            for p in group["params"]:
                st = self.state[p]
                if "z" in st:
                    p.lerp_(end=st["z"].to(p.device), weight=w)
            group["train_mode"] = to_train

    def eval(self):
        self._swap(False)

    def train(self):
        self._swap(True)

    def _chunk_table(self, gi: int, active: List[torch.Tensor]) -> Tuple[torch.Tensor, int]:
        # The table holds raw device pointers of the parameter, its gradient AND its two state tensors, so all four are in the
        # cache key: load_state_dict(), state.clear() or a moved state tensor would otherwise leave the kernel writing through
        # stale pointers.  `_tables` is created lazily: Optimizer.__setstate__ (unpickle, deepcopy) does not restore it.
        tables = self.__dict__.setdefault("_tables", {})
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["z"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel())
                    for p in active)
        hit = tables.get(gi)
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        rows = []
        for p in active:
            st = self.state[p]
            n = p.numel()
            for name, t in (("parameter", p), ("gradient", p.grad), ("z", st["z"]), ("exp_avg_sq", st["exp_avg_sq"])):
                if t.data_ptr() % 16 or t.device != p.device or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
                    raise RuntimeError(f"findtextcenternet_amd.AdamWScheduleFree: {name} of a parameter is not a 16-byte aligned, "
                                       "contiguous fp32 tensor on the parameter's device (view parameters / bucket-view gradients "
                                       "are not supported by the 16-byte-lane kernel)")
            for off in range(0, n, CHUNK):
                rows.append((p.data_ptr() + 4 * off, p.grad.data_ptr() + 4 * off, st["exp_avg_sq"].data_ptr() + 4 * off,
                             st["z"].data_ptr() + 4 * off, min(CHUNK, n - off)))
        arr = (L.MtChunk * len(rows))()
        for i, (y, g, v, z, n) in enumerate(rows):
            arr[i].y, arr[i].g, arr[i].v, arr[i].z, arr[i].n = y, g, v, z, n
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        dev = host.to(active[0].device)
        tables[gi] = (key, dev, len(rows))
        return dev, len(rows)

    @torch.no_grad()
    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:
        if not self.param_groups[0]["train_mode"]:
            raise Exception("Optimizer was not in train mode when step is called. Please insert .train() and .eval() calls on the "
                            "optimizer. See documentation for details.")
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        for gi, group in enumerate(self.param_groups):
            sc = step_scalars(group)
            active = [p for p in group["params"] if p.grad is not None]
            for p in active:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous() and p.grad.dtype == torch.float32):
                    raise RuntimeError("findtextcenternet_amd.AdamWScheduleFree: parameters and gradients must be contiguous fp32 "
                                       "CUDA tensors (no CPU fallback)")
                st = self.state[p]
                if "z" not in st:
                    st["z"] = torch.clone(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if active:
                dev = active[0].device
                with torch.cuda.device(dev):
                    table, n = self._chunk_table(gi, active)
                    f = C.c_float
                    L.check(lib.ftc_adamw_schedulefree_step(table.data_ptr(), n, f(sc["beta2"]), f(sc["one_minus_beta2"]), f(sc["bias_correction2"]),
                                                            f(sc["eps"]), f(sc["weight_decay"]), f(sc["ckp1"]), f(sc["y_alpha"]), f(sc["lr"]), 1,
                                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "ftc_adamw_schedulefree_step")
                bump_versions(active)                # the kernel wrote through raw pointers: tell whoever fingerprints the parameters
            group["k"] = group["k"] + 1
        return loss
