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


# ---- training / validation labels (config 5: train1.py data layout) --------------------------------------------------------
def train_labels(seed: int, b: int, h: int, w: int, n_glyphs: int = 0):
    """Synthetic label maps with the semantics of the reference's sample synthesiser (dataset/processer.pyx:133-202):

    labelmap [B,5,h,w] f32 -- 0: Gaussian centre map (max of per-glyph kernels, exactly 1.0 at a glyph centre),
                              1-2: log-size maps ``log(px/1024)+3`` inside the glyph ellipse, 3: text-line map, 4: separator map (0..1);
    idmap    [B,2,h,w] int32 -- 0: glyph id (unicode-like code point) inside the ellipse, 1: 4-bit code flags.
    All values come from the seed; there are NO ties among the centre-map values other than the exact 0 background and the exact
    1.0 centres (so that the top-k of get_fmask is well defined away from those two plateaus)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    label = np.zeros((b, 5, h, w), np.float32)
    idmap = np.zeros((b, 2, h, w), np.int32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    n_glyphs = n_glyphs or max(4, h * w // 96)
    for bi in range(b):
        label[bi, 1:3] = 0.0
        for _ in range(n_glyphs):
            cy, cx = float(rng.uniform(1, h - 2)), float(rng.uniform(1, w - 2))
            gw, gh = float(rng.uniform(6, 40)), float(rng.uniform(6, 40))          # glyph size in input pixels
            fw, fh = max(gw / 4 / 2, 1.0), max(gh / 4 / 2, 1.0)                    # center_map: half sizes in map pixels
            yi, xi = int(round(cy)), int(round(cx))
            g = np.exp(-(((xx - xi) / (fw / 4)) ** 2 + ((yy - yi) / (fh / 4)) ** 2) / 2).astype(np.float32)
            g *= (np.abs(xx - xi) <= max(fw, fh) * 1.5) & (np.abs(yy - yi) <= max(fw, fh) * 1.5)
            g[yi, xi] = 1.0
            label[bi, 0] = np.maximum(label[bi, 0], g)
            ex, ey = max(gw / 10, 4.0), max(gh / 10, 4.0)                          # box_map / id_map ellipse, input pixels
            inside = ((xx * 4 - cx * 4) / ex) ** 2 + ((yy * 4 - cy * 4) / ey) ** 2 < 1
            label[bi, 1][inside] = np.log(gw / 1024) + 3
            label[bi, 2][inside] = np.log(gh / 1024) + 3
            idmap[bi, 0][inside] = int(rng.integers(0x20, 0x2FFFF))
            idmap[bi, 1][inside] = int(rng.integers(0, 16))
        # tiny seeded jitter on the non-plateau values: no accidental ties for the top-k selection
        jit = rng.random((h, w), dtype=np.float32) * np.float32(1e-4)
        k = label[bi, 0]
        label[bi, 0] = np.where((k > 0) & (k < 1), np.clip(k + jit, 1e-6, 1 - 1e-6), k)
        label[bi, 3] = np.clip(_smooth(rng, h, w, 2) * 2 + 0.3, 0, 1)
        label[bi, 4] = np.clip(_smooth(rng, h, w, 2) * 2 - 0.2, 0, 1)
    return label, idmap


def cov_loss_sequence(seed: int, steps: int = 6):
    """Per-step raw loss values fed to CoVWeightingLoss (train1.py:106-113 key order)."""
    keys = ["keymap_loss", "size_loss", "textline_loss", "separator_loss", "id_loss", "code1_loss", "code2_loss", "code4_loss", "code8_loss"]
    rng = np.random.Generator(np.random.PCG64(seed))
    base = rng.uniform(0.2, 3.0, len(keys))
    return keys, [(base * (0.9 ** s) * rng.uniform(0.8, 1.2, len(keys))).astype(np.float32) for s in range(steps)]


def loss_case(seed: int, b: int, h: int, w: int, target_ids: np.ndarray, modulo=(1091, 1093, 1097)):
    """Synthetic network outputs for the loss kernels: heatmap [B,9,h,w] f32 and three decoder-logit arrays [N, modulo] whose rows
    predict their target id about half of the time (so `correct` is neither 0 nor `total`)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    hm = (rng.standard_normal((b, 9, h, w)) * 2).astype(np.float32)
    dec = [(rng.standard_normal((len(target_ids), m)) * 3).astype(np.float32) for m in modulo]
    for j, m in enumerate(modulo):
        rows = np.nonzero(rng.random(len(target_ids)) < 0.5)[0]
        dec[j][rows, target_ids[rows] % m] += 25.0
    return hm, dec
