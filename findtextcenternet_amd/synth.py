"""Seeded synthetic inputs shared by the golden generator, the parity tests and the bench.

Everything here is regenerable from a seed, so only the reference's *outputs* need to be committed
as fixtures.
"""
from __future__ import annotations

import numpy as np


def noise_images(seed: int, b: int, h: int, w: int) -> np.ndarray:
    """[B,H,W,3] float32 in 0..1 (PCG64 uniform noise)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random((b, h, w, 3), dtype=np.float32)


def page_images(seed: int, b: int, h: int, w: int) -> np.ndarray:
    """[B,H,W,3] float32 in 0..1: white page with seeded dark rectangles ("glyphs") + light noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = np.ones((b, h, w, 3), np.float32)
    for bi in range(b):
        for _ in range(h * w // 600):
            rh, rw = int(rng.integers(4, 28)), int(rng.integers(4, 28))
            y0, x0 = int(rng.integers(0, h - rh)), int(rng.integers(0, w - rw))
            x[bi, y0:y0 + rh, x0:x0 + rw, :] = rng.random(3, dtype=np.float32) * 0.5
    x += (rng.random(x.shape, dtype=np.float32) - 0.5) * 0.1
    return np.clip(x, 0, 1)


def page_uint8(seed: int, h: int, w: int) -> np.ndarray:
    """[H,W,3] uint8 page used by the decode fixtures (the contrast filter of the page-level merge,
    process_ocr_base.py:652-693, looks at the original pixels)."""
    return np.round(page_images(seed, 1, h, w)[0] * 255).astype(np.uint8)


def _smooth(rng, h, w, k):
    a = rng.standard_normal((h + 2 * k, w + 2 * k)).astype(np.float32)
    ker = np.ones(2 * k + 1, np.float32) / (2 * k + 1)
    a = np.apply_along_axis(lambda r: np.convolve(r, ker, mode="valid"), 1, a)
    a = np.apply_along_axis(lambda r: np.convolve(r, ker, mode="valid"), 0, a)
    return a.astype(np.float32)


def detector_maps(seed: int, b: int = 1, h: int = 192, w: int = 192, density: float = 1.0):
    """Synthetic detector outputs with the statistics the decode cares about.

    Returns (heatmap[B,10,h,w] f32 with channel 1 = NMS of channel 0, features[B,100,h,w] f32).
    Channel 0: smooth field + noise (several hundred local maxima above the cut-off);
    2-3: log-size maps giving boxes of ~8..60 px; 4-5: line / separator logits; 6-9: code logits.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    hm = np.empty((b, 10, h, w), np.float32)
    for bi in range(b):
        key = _smooth(rng, h, w, 2) * 6.0 * density + rng.standard_normal((h, w)).astype(np.float32) * 0.8 - 1.5
        hm[bi, 0] = key
        pad = np.pad(key, 1, constant_values=-np.inf)
        win = np.stack([pad[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)]).max(0)
        hm[bi, 1] = np.where(key < win, -np.inf, key)
        hm[bi, 2] = np.log(rng.uniform(8, 60, (h, w)).astype(np.float32) / 1024) + 3
        hm[bi, 3] = np.log(rng.uniform(8, 60, (h, w)).astype(np.float32) / 1024) + 3
        hm[bi, 4] = _smooth(rng, h, w, 3) * 8.0
        hm[bi, 5] = _smooth(rng, h, w, 3) * 8.0 - 2.0
        for k in range(4):
            hm[bi, 6 + k] = rng.standard_normal((h, w)).astype(np.float32) * 2.0
    feat = rng.standard_normal((b, 100, h, w)).astype(np.float32) * 4.0
    return hm, feat


# ---- Schedule-Free AdamW cases (g6): seeded parameters and per-step gradients ---------------------------------------------
ADAMW_CASES = [
    {"kwargs": dict(lr=0.0025, weight_decay=0.01, warmup_steps=3), "shapes": [(5000,), (37, 5), (3, 3, 3, 3), (1,)], "steps": 6},
    {"kwargs": dict(lr=0.01, betas=(0.95, 0.99), eps=1e-6, weight_decay=0, warmup_steps=0, r=1.0, weight_lr_power=1.0),
     "shapes": [(4097,), (8, 8)], "steps": 4},
]


def adamw_case(ci: int):
    """(params0: list of fp32 arrays, grads: per step a list of fp32 arrays) of ADAMW_CASES[ci]."""
    cfg = ADAMW_CASES[ci]
    rng = np.random.Generator(np.random.PCG64(4000 + ci))
    params0 = [rng.standard_normal(s).astype(np.float32) for s in cfg["shapes"]]
    grads = [[(rng.standard_normal(s) * 10.0 ** rng.uniform(-3, 1)).astype(np.float32) for s in cfg["shapes"]] for _ in range(cfg["steps"])]
    return params0, grads
