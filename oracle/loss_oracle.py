"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's validation-step arithmetic:

* ``get_fmask``        -- ``TextDetectorModel.get_fmask``, /root/reference/models/detector.py:270-281
* ``simple_decoder``   -- ``SimpleDecoder.forward`` in eval mode, /root/reference/models/detector.py:232-254
* ``heatmap_loss`` / ``loss_function`` -- /root/reference/loss_func.py:74-92, :94-177
* ``CoVWeighting``     -- ``CoVWeightingLoss.forward``, /root/reference/loss_func.py:24-72

Plain torch-CPU / numpy, fp32 as the reference.  Pinned by tests/golden/g7_validation_step.npz, which
tests/golden/gen_golden.py wrote by running the reference's own ``models/detector.py`` and ``loss_func.py``
(tests/test_oracle.py::test_validation_step_oracle_matches_reference).  Only tests/ may import this.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

MODULO = [1091, 1093, 1097]        # util_func.py:5


def get_fmask(labelmap: torch.Tensor) -> torch.Tensor:
    """models/detector.py:270-281 (stable descending sort: ties keep ascending index order, as torch's CPU sort does)."""
    b = labelmap.shape[0]
    flat = labelmap[:, 0].flatten()
    idx = torch.sort(flat, descending=True, stable=True).indices
    mask = torch.zeros_like(flat, dtype=torch.bool)
    mask[idx[:1024 * b]] = True
    return mask


def simple_decoder(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> List[torch.Tensor]:
    """``sd``: state dict with ``decoder.blocks.<i>...`` keys; x [N,100]."""
    out = []
    for i in range(3):
        p = f"decoder.blocks.{i}"
        y = F.linear(x, sd[p + ".0.weight"])
        y = F.gelu(F.batch_norm(y, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"], False, 0.0, 1e-5))
        y = F.linear(y, sd[p + ".3.weight"])
        y = F.gelu(F.batch_norm(y, sd[p + ".4.running_mean"], sd[p + ".4.running_var"], sd[p + ".4.weight"], sd[p + ".4.bias"], False, 0.0, 1e-5))
        out.append(F.linear(y, sd[p + ".6.weight"], sd[p + ".6.bias"]))
    return out


def heatmap_loss(true: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """loss_func.py:74-92."""
    x = logits.float()
    pr = torch.sigmoid(x)
    pos = (true >= 1.0).float()
    neg = (true < 1.0).float()
    pos_loss = -F.logsigmoid(x) * (1 - pr) ** 2 * pos
    neg_loss = (x + F.softplus(-x)) * pr ** 2 * (1.0 - true) ** 4 * neg
    return (pos_loss + neg_loss).mean()


def loss_function(fmask, labelmap, idmap, heatmap, decoder_outputs: Sequence[torch.Tensor]) -> Dict[str, torch.Tensor]:
    """loss_func.py:94-177."""
    key = labelmap[:, 0]
    m1 = key > 0.85
    kf, idf = key.flatten()[fmask], idmap[:, 0].flatten()[fmask]
    m3 = (kf > 0.99) & (idf > 0)
    m4 = (kf == 1) & (idf > 0)
    w1 = (torch.clamp(key - 0.85, min=0) / (1 - 0.85))[m1]
    w1c = torch.clamp(w1.sum(), min=1.0)
    w2 = torch.clamp(key - 0.85, min=0) / (1 - 0.85)
    w3 = (torch.clamp(kf - 0.99, min=0) / (1 - 0.99))[m3]
    w3c = torch.clamp(w3.sum(), min=1.0)
    out = {"keymap_loss": heatmap_loss(key, heatmap[:, 0]) * 10.0}
    hub = torch.nn.HuberLoss(reduction="none")
    size = (hub(heatmap[:, 1][m1], labelmap[:, 1][m1]) + hub(heatmap[:, 2][m1], labelmap[:, 2][m1])) * w1
    out["size_loss"] = size.sum() / w1c
    out["textline_loss"] = F.binary_cross_entropy_with_logits(heatmap[:, 3], labelmap[:, 3])
    out["separator_loss"] = F.binary_cross_entropy_with_logits(heatmap[:, 4], labelmap[:, 4])
    for i in range(4):
        bit = ((idmap[:, 1] & (1 << i)) > 0).float()
        out[f"code{1 << i}_loss"] = F.binary_cross_entropy_with_logits(heatmap[:, 5 + i], bit, weight=torch.ones_like(bit) + bit * w2 + w2)
    id_loss = torch.zeros(())
    right = torch.zeros(int(m4.sum()), dtype=torch.long)
    for m, d in zip(MODULO, decoder_outputs):
        ce = F.cross_entropy(d[m3], (idf % m)[m3], reduction="none")
        id_loss = id_loss + (ce * w3).sum() / w3c
        right += (torch.argmax(d[m4], dim=-1) == (idf % m)[m4]).long()
    out["id_loss"] = id_loss
    out["correct"] = (right == 3).sum()
    out["total"] = torch.tensor(int(m4.sum()))
    out["loss"] = sum(out[k] for k in ("keymap_loss", "size_loss", "textline_loss", "separator_loss", "id_loss", "code1_loss", "code2_loss", "code4_loss", "code8_loss"))
    return out


class CoVWeighting:
    """loss_func.py:8-72 in numpy float32 (the reference's branch for validation is never taken, :29)."""

    def __init__(self, n: int):
        self.n, self.it = n, -1
        self.mean_L = np.zeros(n, np.float32)
        self.mean_l = np.zeros(n, np.float32)
        self.S_l = np.zeros(n, np.float32)
        self.std_l = None
        self.alphas = np.zeros(n, np.float32)

    def __call__(self, L: np.ndarray) -> float:
        L = L.astype(np.float32)
        self.it += 1
        L0 = L.copy() if self.it == 0 else self.mean_L
        l = L / L0
        if self.it <= 1:
            self.alphas = np.ones(self.n, np.float32) / np.float32(self.n)
        else:
            ls = self.std_l / self.mean_l
            self.alphas = ls / ls.sum(dtype=np.float32)
        mp = 0.0 if self.it == 0 else (1.0 - 1 / (self.it + 1))
        new_mean = np.float32(mp) * self.mean_l + np.float32(1 - mp) * l
        self.S_l = self.S_l + (l - self.mean_l) * (l - new_mean)
        self.mean_l = new_mean
        self.std_l = np.sqrt(np.maximum(self.S_l / np.float32(self.it + 1), np.float32(1e-16)))
        self.mean_L = np.float32(mp) * self.mean_L + np.float32(1 - mp) * L
        acc = np.float32(0)
        for a, v in zip(self.alphas, L):
            acc = np.float32(acc + a * v)
        return float(acc)
