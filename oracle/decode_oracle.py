"""TEST INFRASTRUCTURE ONLY -- numpy oracle for the host-side peak decode of the reference.

Restates, function by function:

* ``util_func.sigmoid`` (``/root/reference/util_func.py:14-15``): ``(tanh(x/2)+1)/2`` in the input dtype;
* the per-tile block of ``OCR_Processer.run_detector`` (``process_ocr_base.py:487-538``), which is
  identical to ``test_image1_torch.py:88-143`` except for the border-mask margins
  (``(1-step_ratio)/2`` vs 1/8) -- see ``tile_keep_rect``;
* the page-level part of ``run_detector`` (``process_ocr_base.py:540-650``): contrast filter
  (``imageHist`` ``:652-693``), greedy suppression, separator filter, 3x3 code max.

Pinned by ``tests/golden/decode_*.npz``: produced by calling the reference's own
``OCR_Processer.run_detector`` in the build container (``tests/golden/gen_golden.py``).

Ordering note: the reference sorts with ``np.argsort(-peak.ravel())`` (``process_ocr_base.py:519``),
an unstable introsort, so the order among exactly equal scores is unspecified there.  This oracle
(and the HIP decode) use the total order (score descending, flat pixel index ascending).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np

width = 768      # util_func.py:6-9
height = 768
scale = 4
feature_dim = 100


def sigmoid(x):
    return (np.tanh(x / 2) + 1) / 2


def tile_keep_rect(x_i: int, y_i: int, page_w: int, page_h: int, step_ratio: float = None) -> Tuple[int, int, int, int]:
    """Inner region of a tile that is trusted (``process_ocr_base.py:498-503``; with
    ``step_ratio=None`` the 1/8 margins of ``test_image1_torch.py:103-108``).  Returns
    (x_min, x_max, y_min, y_max) in map pixels, max exclusive."""
    x_s, y_s = width // scale, height // scale
    if step_ratio is None:
        lo_x, hi_x = int(x_s * 1 / 8), int(x_s * 7 / 8) + 1
        lo_y, hi_y = int(y_s * 1 / 8), int(y_s * 7 / 8) + 1
    else:
        lo_x, hi_x = int(x_s * (1 - step_ratio) / 2), int(x_s * (1 - (1 - step_ratio) / 2)) + 1
        lo_y, hi_y = int(y_s * (1 - step_ratio) / 2), int(y_s * (1 - (1 - step_ratio) / 2)) + 1
    x_min = lo_x if x_i > 0 else 0
    x_max = hi_x if x_i + width < page_w else x_s
    y_min = lo_y if y_i > 0 else 0
    y_max = hi_y if y_i + height < page_h else y_s
    return x_min, x_max, y_min, y_max


def decode_tile(heatmap: np.ndarray, features: np.ndarray, x_i: int, y_i: int, page_w: int, page_h: int,
                cut_off: float, keep_rect: Tuple[int, int, int, int]):
    """Per-tile peak decode (``process_ocr_base.py:498-538``).

    heatmap [1,10,h,w] f32, features [1,100,h,w] f32 ->
    (locations [N,9] f64: p, ix, iy, w, h, code1, code2, code4, code8;  glyphfeatures [N,100] f32;
     flat pixel indices [N] int64).  Rows are in (score desc, index asc) order."""
    hm = heatmap[0]
    y_s, x_s = hm.shape[1:]
    x_min, x_max, y_min, y_max = keep_rect
    mask = np.zeros([y_s, x_s], dtype=bool)
    mask[y_min:y_max, x_min:x_max] = True
    code_p = [sigmoid(hm[6 + k]) for k in range(4)]
    peak = sigmoid(hm[1]) * mask
    order = np.argsort(-peak.ravel(), kind="stable")
    locs, feats, idxs = [], [], []
    for flat in order:
        y, x = divmod(int(flat), x_s)
        if peak[y, x] < cut_off:
            break
        w = np.exp(hm[2, y, x] - 3) * 1024
        h = np.exp(hm[3, y, x] - 3) * 1024
        if w <= 0 or h <= 0:
            continue
        if w > page_w or h > page_h:
            continue
        ix = x * scale + x_i
        iy = y * scale + y_i
        locs.append(np.array([peak[y, x], ix, iy, w, h] + [code_p[k][y, x] for k in range(4)]))
        feats.append(features[0, :, y, x])
        idxs.append(flat)
    if not locs:
        return np.zeros([0, 9]), np.zeros([0, feature_dim], np.float32), np.zeros([0], np.int64)
    return np.array(locs), np.array(feats), np.array(idxs, dtype=np.int64)


def paste_maps(page_maps: Sequence[np.ndarray], heatmap: np.ndarray, x_i: int, y_i: int,
               keep_rect: Tuple[int, int, int, int]) -> None:
    """Sigmoid maps merged into page canvases with np.maximum (``process_ocr_base.py:505-516``).
    page_maps = [keymap_all, lines_all, seps_all, code1..code8] updated in place."""
    hm = heatmap[0]
    y_s, x_s = hm.shape[1:]
    x_min, x_max, y_min, y_max = keep_rect
    mask = np.zeros([y_s, x_s], dtype=bool)
    mask[y_min:y_max, x_min:x_max] = True
    x_is, y_is = x_i // scale, y_i // scale
    for canvas, ch in zip(page_maps, [0, 4, 5, 6, 7, 8, 9]):
        sl = canvas[y_is:y_is + y_s, x_is:x_is + x_s]
        sl[...] = np.maximum(sigmoid(hm[ch]) * mask, sl)


def image_hist(im: np.ndarray) -> float:
    """``OCR_Processer.imageHist`` (``process_ocr_base.py:652-693``): per colour channel, distance
    between the two 1-D k-means cluster centres of the 256-bin histogram; max over channels."""
    def cluster_dist(hist):
        tot = np.sum(hist)
        if tot == 0:
            return 0
        i = np.arange(hist.shape[0])
        v = hist * i
        cut = int(np.sum(v) / tot + 0.5)
        s1, s2 = np.sum(hist[:cut]), np.sum(hist[cut:])
        if s1 == 0 or s2 == 0:
            return 0
        k1, k2 = np.sum(v[:cut]) / s1, np.sum(v[cut:]) / s2
        prev, cur = 256.0, abs(k1 - k2)
        while prev != cur:
            prev = cur
            near1 = np.abs(i - k1) < np.abs(i - k2)
            s1, s2 = np.sum(hist[near1]), np.sum(hist[~near1])
            if s1 == 0 or s2 == 0:
                return 0
            k1, k2 = np.sum(v[near1]) / s1, np.sum(v[~near1]) / s2
            cur = abs(k1 - k2)
        return prev
    best = -1
    for c in range(3):
        best = max(best, cluster_dist(np.histogram(im[:, :, c], bins=256, range=(0, 256))[0]))
    return best


def page_merge(locations: np.ndarray, glyphfeatures: np.ndarray, org_img: np.ndarray, seps_all: np.ndarray,
               code_all: Sequence[np.ndarray], cut_off: float, variant: str = "production"):
    """Page-level selection (``process_ocr_base.py:540-650``).  ``locations`` carries the reference's
    leading all-zero dummy row (``:478-479``).

    ``variant="demo"``: the same block of the demo script's ``eval()`` (``test_image1_torch.py:152-240``), which differs in
    three places: no ``imageHist`` contrast filter; the coverage image ``fill_map`` is filled with ``p2x`` WITHOUT the ``+1``
    and ``p1y`` WITH a ``+1`` (``:196-200``); the result stays float64 (no ``astype(np.float32)``).  Its two-pass seed rows
    (``:150-151``) are the caller's business: ``eval_demo`` appends them."""
    demo = variant == "demo"
    assert variant in ("production", "demo")
    page_h, page_w = org_img.shape[:2]
    hists = []
    for i in range(locations.shape[0] if not demo else 0):
        p, cx, cy, w, h = locations[i, :5]
        if p < cut_off:
            continue
        hists.append(image_hist(org_img[int(cy - h / 2) - 1:int(cy + h / 2) + 2, int(cx - w / 2) - 1:int(cx + w / 2) + 2, :]))
    th_hist = np.median(hists) / 5 if not demo else None

    done = np.zeros([0, 4])
    keep: List[int] = []
    for i in np.argsort(-locations[:, 0], kind="stable"):
        p, cx, cy, w, h = locations[i, :5]
        if p < cut_off:
            break
        if not demo:
            bx0, bx1 = max(0, int(cx - w / 2)), min(page_w - 1, int(cx + w / 2) + 1)
            by0, by1 = max(0, int(cy - h / 2)), min(page_h - 1, int(cy + h / 2) + 1)
            if image_hist(org_img[by0:by1, bx0:bx1, :]) < th_hist:
                continue
        a0 = w * h
        fill = np.zeros([int(w), int(h)], dtype=bool)
        if done.size > 0:
            a1 = done[:, 2] * done[:, 3]
            ix0 = np.maximum(cx - w / 2, done[:, 0] - done[:, 2] / 2)
            iy0 = np.maximum(cy - h / 2, done[:, 1] - done[:, 3] / 2)
            ix1 = np.minimum(cx + w / 2, done[:, 0] + done[:, 2] / 2)
            iy1 = np.minimum(cy + h / 2, done[:, 1] + done[:, 3] / 2)
            inter = np.maximum(ix1 - ix0, 0.) * np.maximum(iy1 - iy0, 0.)
            union = a0 + a1 - inter
            iou = np.where(union > 0., inter / union, 0.)
            if iou.max() > 0.5:
                continue
            if inter.max() > a0 * 0.75:
                continue
            for j in np.where(iou > 0)[0]:
                cx1, cy1, w1, h1 = done[j]
                p1x = int(max(cx1 - w1 / 2, cx - w / 2) - (cx - w / 2))
                p2x = int(min(cx1 + w1 / 2, cx + w / 2) - (cx - w / 2)) + (0 if demo else 1)
                p1y = int(max(cy1 - h1 / 2, cy - h / 2) - (cy - h / 2)) + (1 if demo else 0)
                p2y = int(min(cy1 + h1 / 2, cy + h / 2) - (cy - h / 2)) + 1
                fill[p1x:p2x, p1y:p2y] = True
            if np.mean(fill) > 0.5:
                continue
        done = np.vstack([done, np.array([cx, cy, w, h])])
        keep.append(i)

    keep2 = []
    mh, mw = page_h // scale, page_w // scale
    for i in keep:
        x, y = int(locations[i, 1] / scale), int(locations[i, 2] / scale)
        if 0 <= x < mw and 0 <= y < mh and seps_all[y, x] > 0.5:
            continue
        keep2.append(i)
    if keep2:
        sel = np.array(keep2)
        locations, glyphfeatures = locations[sel, :], glyphfeatures[sel, :]
    else:
        locations, glyphfeatures = np.zeros([0, 9]), np.zeros([0, feature_dim], dtype=np.float32)

    for i in range(locations.shape[0]):
        cx, cy = locations[i, 1], locations[i, 2]
        x, y = int(cx / scale), int(cy / scale)
        if 0 <= x < mw and 0 <= y < mh:
            x0, y0 = max(0, int(cx / scale - 1)), max(0, int(cy / scale - 1))
            x1, y1 = min(mw, int(cx / scale + 1) + 1), min(mh, int(cy / scale + 1) + 1)
            for k in range(4):
                locations[i, 5 + k] = max(np.max(code_all[k][y0:y1, x0:x1]), locations[i, 5 + k])
    return (locations if demo else locations.astype(np.float32)), glyphfeatures


def run_detector(ds: Sequence[dict], org_img: np.ndarray, call_detector: Callable, step_ratio: float = 0.6,
                 cut_off: float = 0.4):
    """``OCR_Processer.run_detector`` (``process_ocr_base.py:474-650``) end to end.
    Returns (locations f32 [M,9], glyphfeatures [M,100], lines_all, seps_all, raw) where ``raw`` is
    the per-tile decode before page_merge (list of (locations, features, flat indices))."""
    page_h, page_w = org_img.shape[:2]
    canv = [np.zeros([page_h // scale, page_w // scale], dtype=np.float32) for _ in range(7)]
    locs = [np.zeros([1, 9])]
    feats = [np.zeros([1, feature_dim], dtype=np.float32)]
    raw = []
    for inputs in ds:
        x_i, y_i = inputs["offsetx"], inputs["offsety"]
        heatmap, features = call_detector(inputs["input"])
        rect = tile_keep_rect(x_i, y_i, page_w, page_h, step_ratio)
        paste_maps(canv, heatmap, x_i, y_i, rect)
        l, f, idx = decode_tile(heatmap, features, x_i, y_i, page_w, page_h, cut_off, rect)
        raw.append((l, f, idx))
        locs.append(l)
        feats.append(f)
    locations = np.concatenate(locs, axis=0)
    glyphfeatures = np.concatenate(feats, axis=0)
    locations, glyphfeatures = page_merge(locations, glyphfeatures, org_img, canv[2], canv[3:], cut_off)
    return locations, glyphfeatures, canv[1], canv[2], raw


def eval_demo(ds: Sequence[dict], org_img: np.ndarray, call_detector: Callable, cut_off: float = 0.5,
              locations0: np.ndarray = None, glyphfeatures0: np.ndarray = None, tile: int = width, return_candidates: bool = False):
    """``eval()`` of the demo script (``/root/reference/test_image1_torch.py:75-240``; the plotting tail ``:242-266`` left out) end
    to end: per-tile block with the 1/8 border margins (``:103-108``), float64 page canvases (``:81-87``), the seed rows of a coarse
    first pass appended BEHIND the tile rows (``:147-151``), then the demo variant of the page-level selection.  ``tile`` = the
    script's ``width`` = ``height`` (768; the golden fixture runs the script's own source with a smaller value to stay small).
    Returns (locations f64 [M,9], glyphfeatures f32 [M,C], keymap_all, lines_all, seps_all, code_all)."""
    page_h, page_w = org_img.shape[:2]
    canv = [np.zeros([page_h // scale, page_w // scale]) for _ in range(7)]            # float64, as np.zeros gives
    nfeat = None
    locs = [np.zeros([1, 9])]
    feats = []
    x_s = y_s = tile // scale
    for inputs in ds:
        x_i, y_i = inputs["offsetx"], inputs["offsety"]
        heatmap, features = call_detector(inputs["input"])
        if nfeat is None:
            nfeat = features.shape[1]
            feats.append(np.zeros([1, nfeat], dtype=np.float32))
        x_min = int(x_s * 1 / 8) if x_i > 0 else 0
        x_max = int(x_s * 7 / 8) + 1 if x_i + tile < page_w else x_s
        y_min = int(y_s * 1 / 8) if y_i > 0 else 0
        y_max = int(y_s * 7 / 8) + 1 if y_i + tile < page_h else y_s
        rect = (x_min, x_max, y_min, y_max)
        paste_maps(canv, heatmap, x_i, y_i, rect)
        l, f, _ = decode_tile(heatmap, features, x_i, y_i, page_w, page_h, cut_off, rect)
        if len(l):
            locs.append(l)
            feats.append(f)
    locations = np.concatenate(locs, axis=0)
    glyphfeatures = np.concatenate(feats, axis=0) if feats else np.zeros([1, feature_dim], np.float32)
    if locations0 is not None:
        locations = np.concatenate([locations, locations0])
    if glyphfeatures0 is not None:
        glyphfeatures = np.concatenate([glyphfeatures, glyphfeatures0])
    if return_candidates:                 # (tests: what the selection sees -- rows in eval()'s concatenation order, seed rows last -- and the canvases)
        return locations, glyphfeatures, canv
    locations, glyphfeatures = page_merge(locations, glyphfeatures, org_img, canv[2], canv[3:], cut_off, variant="demo")
    return locations, glyphfeatures, canv[0], canv[1], canv[2], canv[3:]
