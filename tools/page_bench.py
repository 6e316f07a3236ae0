#!/usr/bin/env python3
"""Whole-page timing of PageDetector.detect_page (tiling, batched forward, GPU decode, GPU paste, GPU page merge) on a synthetic
A4 page at 300 dpi, plus the page-level selection alone: GPU (ftc_box_hists + ftc_page_merge) vs the NumPy oracle on the same boxes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402
from findtextcenternet_amd import CenterNetDetector, PageDetector, TextDetectorModel, deterministic_state_dict, page_merge_gpu  # noqa: E402
from oracle import decode_oracle  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
m = TextDetectorModel(pre_weights=False, precision=prec)
m.load_state_dict(deterministic_state_dict(0))
det = CenterNetDetector(m.detector).to("cuda").eval()
pd = PageDetector(det, batch=8, max_boxes=4096)
img = synth.page_uint8(7, 3508, 2480)
pd.detect_page(img)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    loc, gf, lines, seps = pd.detect_page(img)
    ts.append(time.perf_counter() - t0)
ph, pw = lines.shape[0] * 4, lines.shape[1] * 4
ntiles = len(__import__("findtextcenternet_amd").page.tile_origins(ph, pw, pd.stepx, pd.stepy))
print(f"detect_page {img.shape[1]}x{img.shape[0]} ({ntiles} tiles, {prec}): {min(ts) * 1e3:.1f} ms/page, {len(loc)} boxes kept")

# page-level selection alone, on a dense synthetic box set
rng = np.random.Generator(np.random.PCG64(5))
n = 6000
page = synth.page_uint8(9, 3508, 2480).astype(np.float32)
mh, mw = 3508 // 4, 2480 // 4
centres = rng.uniform([0, 0], [2480, 3508], size=(n // 6, 2))
cx = (centres[rng.integers(0, len(centres), n), 0] + rng.normal(0, 14, n)).astype(np.float32)
cy = (centres[rng.integers(0, len(centres), n), 1] + rng.normal(0, 14, n)).astype(np.float32)
w = np.exp(rng.uniform(np.log(10), np.log(70), n)).astype(np.float32)
h = np.exp(rng.uniform(np.log(10), np.log(70), n)).astype(np.float32)
pr = rng.uniform(0.3, 1.0, n).astype(np.float32)
loc32 = np.stack([pr, cx, cy, w, h, *rng.uniform(0, 1, (4, n)).astype(np.float32)], 1)
feats = rng.standard_normal((n, 100)).astype(np.float32)
seps = (rng.uniform(0, 1, (mh, mw)) ** 4).astype(np.float32)
codes = [rng.uniform(0, 1, (mh, mw)).astype(np.float32) for _ in range(4)]
t0 = time.perf_counter()
ref_loc, ref_gf = decode_oracle.page_merge(loc32.astype(np.float64), feats.copy(), page, seps, codes, 0.4)
t_cpu = time.perf_counter() - t0
dev = torch.device("cuda")
canv = torch.zeros((7, mh, mw), device=dev)
canv[2] = torch.from_numpy(seps).to(dev)
for k in range(4):
    canv[3 + k] = torch.from_numpy(codes[k]).to(dev)
args = (torch.from_numpy(loc32).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(page).to(dev), canv, 0.4)
page_merge_gpu(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
g_loc, g_gf = page_merge_gpu(*args)
torch.cuda.synchronize()
t_gpu = time.perf_counter() - t0
print(f"page selection, {n} boxes -> {len(ref_loc)} kept: NumPy oracle {t_cpu * 1e3:.0f} ms, GPU {t_gpu * 1e3:.1f} ms, identical: "
      f"{np.array_equal(g_loc.cpu().numpy(), ref_loc) and np.array_equal(g_gf.cpu().numpy(), ref_gf)}")
