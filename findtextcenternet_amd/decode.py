"""Host side of the GPU peak decode (``ftc_decode``) and the ``call_detector`` plug-in mirror.

Reference: the per-tile block of ``OCR_Processer.run_detector``
(``/root/reference/process_ocr_base.py:487-538``, same code in ``test_image1_torch.py:88-143``) and
``OCR_torch_Processer.call_detector`` (``/root/reference/process_ocr_torch.py:43-49``).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .schema import feature_dim, height, scale, width


def _ref_sigmoid_f32(x: np.ndarray) -> np.ndarray:
    return (np.tanh(x / 2) + 1) / 2          # util_func.py:14-15, float32 in -> float32 out


def exact_logit_cut(cut_off: float) -> float:
    """Smallest float32 logit v with sigmoid_f32(v) >= float32(cut_off): the reference keeps a
    peak iff ``sigmoid(logit) >= cut_off`` evaluated in float32 (``process_ocr_base.py:521-523``), so
    thresholding the raw logit at this value selects exactly the same pixels without computing a
    transcendental per pixel.  Found by bisection over the float32 bit pattern with the same numpy
    expression the reference uses (monotone in practice)."""
    c = np.float32(cut_off)
    if not (0.0 < float(c) < 1.0):
        raise ValueError("cut_off must be in (0,1)")
    lo, hi = np.float32(-40.0), np.float32(40.0)
    ilo, ihi = _f2o(lo), _f2o(hi)
    while ihi - ilo > 1:
        mid = (ilo + ihi) // 2
        if _ref_sigmoid_f32(np.array([_o2f(mid)], np.float32))[0] >= c:
            ihi = mid
        else:
            ilo = mid
    return float(_o2f(ihi))


def _f2o(f: np.float32) -> int:
    u = int(np.array([f], np.float32).view(np.uint32)[0])
    return (~u & 0xFFFFFFFF) if u & 0x80000000 else (u | 0x80000000)


def _o2f(o: int) -> np.float32:
    u = (o & 0x7FFFFFFF) if o & 0x80000000 else (~o & 0xFFFFFFFF)
    return np.array([u], np.uint32).view(np.float32)[0]


def tile_keep_rect(x_i: int, y_i: int, page_w: int, page_h: int, step_ratio: Optional[float] = 0.6,
                   tile_w: int = width, tile_h: int = height) -> Tuple[int, int, int, int]:
    """Trusted inner rectangle of a tile in map pixels, (x_min, x_max, y_min, y_max), max exclusive.
    ``step_ratio`` float: production margins (``process_ocr_base.py:498-503``); ``None``: the 1/8
    margins of the demo script (``test_image1_torch.py:103-108``)."""
    x_s, y_s = tile_w // scale, tile_h // scale
    if step_ratio is None:
        lx, hx, ly, hy = int(x_s * 1 / 8), int(x_s * 7 / 8) + 1, int(y_s * 1 / 8), int(y_s * 7 / 8) + 1
    else:
        lx, hx = int(x_s * (1 - step_ratio) / 2), int(x_s * (1 - (1 - step_ratio) / 2)) + 1
        ly, hy = int(y_s * (1 - step_ratio) / 2), int(y_s * (1 - (1 - step_ratio) / 2)) + 1
    return (lx if x_i > 0 else 0, hx if x_i + tile_w < page_w else x_s,
            ly if y_i > 0 else 0, hy if y_i + tile_h < page_h else y_s)


@dataclass
class TileGeom:
    offset_x: int
    offset_y: int
    page_w: int
    page_h: int
    rect: Tuple[int, int, int, int]      # x_min, x_max, y_min, y_max


REC_W = 112          # floats per record row: box (9) + 3 pad + feature row (100, 16-byte aligned)
REC_FEAT0 = 12


@dataclass
class Decoded:
    boxes: torch.Tensor      # [B, max_boxes, 9] f32 view: p, ix, iy, w, h, code1, code2, code4, code8
    feats: torch.Tensor      # [B, max_boxes, C] f32 view
    index: torch.Tensor      # [B, max_boxes] int32 flat map index y*w+x
    counts: torch.Tensor     # [B] int32 number of peaks found (may exceed max_boxes)
    records: Optional[torch.Tensor] = None   # [B, max_boxes, 112] f32: the block `boxes` / `feats` are views of


class DecodeWorkspace:
    """Output block + scratch of ``decode_peaks`` for a fixed (B, h, w, C, max_boxes), allocated once and reused:
    nothing is allocated or cleared per call, so rows at and beyond ``counts[b]`` keep whatever an earlier call left there --
    consumers of a REUSED workspace must slice by ``counts`` (``page_merge_gpu`` and the box gather treat rows with p < cut_off as
    inert padding, which only holds for the zero-initialised block of a fresh workspace: ``PageDetector`` uses a fresh one per batch)."""

    def __init__(self, B: int, h: int, w: int, C_: int, max_boxes: int, device):
        if C_ + REC_FEAT0 > REC_W:
            raise ValueError("record layout is sized for feature rows of at most 100 floats")
        lib = L.load()
        self.key = (B, h, w, C_, max_boxes)
        self.rec_w = REC_W
        self.records = torch.zeros((B, max_boxes, self.rec_w), dtype=torch.float32, device=device)
        self.index = torch.full((B, max_boxes), -1, dtype=torch.int32, device=device)
        self.counts = torch.zeros((B,), dtype=torch.int32, device=device)
        self.scratch = torch.empty(int(lib.ftc_decode_scratch_bytes(B, h, w)), dtype=torch.uint8, device=device)

    def decoded(self) -> Decoded:
        C_ = self.key[3]
        return Decoded(self.records[:, :, 0:9], self.records[:, :, REC_FEAT0:REC_FEAT0 + C_], self.index, self.counts, self.records)


def tiles_to_device(tiles: Sequence[TileGeom], device, h: int, w: int) -> torch.Tensor:
    """[B,8] int32 ftc_tile records on the GPU (upload once when the geometry repeats)."""
    tl = np.array([[t.offset_x, t.offset_y, t.page_w, t.page_h, *t.rect] for t in tiles], np.int32)
    for r in tl:
        if not (0 <= r[4] <= r[5] <= w and 0 <= r[6] <= r[7] <= h):
            raise ValueError("tile rectangle outside the map")
    return torch.from_numpy(tl).to(device)


def decode_peaks(heat_nhwc: torch.Tensor, feat_nhwc: torch.Tensor, tiles, cut_off: float = 0.4,
                 max_boxes: int = 4096, logit_cut: Optional[float] = None, workspace: Optional[DecodeWorkspace] = None) -> Decoded:
    """heat_nhwc [B,h,w,10] fp32, feat_nhwc [B,h,w,C] fp32 (NHWC memory, on the GPU); ``tiles`` is a
    sequence of TileGeom or the [B,8] int32 device tensor from ``tiles_to_device``.  The kernel writes box and feature row
    of a peak into ONE record row (``Decoded.records`` [B,max_boxes,112]); ``boxes`` / ``feats`` are views of it.  Without a
    ``workspace`` a fresh zero-filled block is allocated per call."""
    if not (heat_nhwc.is_cuda and feat_nhwc.is_cuda):
        raise RuntimeError("decode_peaks runs on the GPU only (no CPU fallback)")
    lib = L.load()
    heat_nhwc, feat_nhwc = heat_nhwc.contiguous(), feat_nhwc.contiguous()
    B, h, w, ch = heat_nhwc.shape
    Cf = feat_nhwc.shape[3]
    assert ch == 10 and heat_nhwc.dtype == torch.float32 and feat_nhwc.dtype == torch.float32
    assert len(tiles) == B and feat_nhwc.shape[:3] == heat_nhwc.shape[:3]
    dev = heat_nhwc.device
    with torch.cuda.device(dev):
        if isinstance(tiles, torch.Tensor):
            tl_dev = tiles
            assert tl_dev.shape == (B, 8) and tl_dev.dtype == torch.int32 and tl_dev.device == dev and tl_dev.is_contiguous()
        else:
            tl_dev = tiles_to_device(tiles, dev, h, w)
        ws = workspace
        if ws is None:
            ws = DecodeWorkspace(B, h, w, Cf, max_boxes, dev)
        elif ws.key != (B, h, w, Cf, max_boxes) or ws.records.device != dev:
            raise ValueError(f"DecodeWorkspace was built for {ws.key}, not {(B, h, w, Cf, max_boxes)}")
        lc = exact_logit_cut(cut_off) if logit_cut is None else logit_cut
        stream = torch.cuda.current_stream(dev).cuda_stream
        L.check(lib.ftc_decode(heat_nhwc.data_ptr(), feat_nhwc.data_ptr(), B, h, w, Cf, tl_dev.data_ptr(), C.c_float(lc),
                               scale, max_boxes, ws.records.data_ptr(), ws.rec_w, ws.records.data_ptr() + 4 * REC_FEAT0, ws.rec_w,
                               ws.index.data_ptr(), ws.counts.data_ptr(), ws.scratch.data_ptr(), C.c_void_p(stream)), "ftc_decode")
    return ws.decoded()


class HipDetectorBackend:
    """Mirror of the detector half of ``OCR_torch_Processer`` (``process_ocr_torch.py:7-49``):
    ``call_detector(image_input[1,768,768,3] float32 0..255) -> (heatmap[1,10,192,192], features[1,100,192,192])``
    as numpy arrays, plus ``detect_tiles`` which keeps the maps on the GPU and returns only the
    decoded peaks (what makes multi-tile batches feasible, SURVEY.md section 7 "Output volume")."""

    def __init__(self, detector, device: str = "cuda"):
        self.device = torch.device(device)
        detector.to(device=self.device)
        detector.eval()
        self.detector = detector

    def call_detector(self, image_input: np.ndarray):
        """The reference's calling convention: one tile, synchronous, host arrays in and out.  (A one-tile forward is launch-bound; the
        module replays it from a HIP graph -- ``_HipEngine._run_graph`` -- so this is 5.7 ms per tile instead of 7.7 in bf16.)"""
        images = torch.from_numpy(np.asarray(image_input, dtype=np.float32) / np.float32(255.)).permute(0, 3, 1, 2).to(device=self.device)
        with torch.no_grad():
            heatmap, features = self.detector(images)
            heatmap = heatmap.cpu().numpy()
            features = features.cpu().numpy()
        return heatmap, features

    def detect_tiles(self, images_u8_or_f32: np.ndarray, tiles: Sequence[TileGeom], cut_off: float = 0.4,
                     max_boxes: int = 4096):
        """images [B,768,768,3] (0..255) -> (Decoded, heat_nhwc, feat_nhwc) all on the GPU."""
        x = torch.from_numpy(np.asarray(images_u8_or_f32, dtype=np.float32) / np.float32(255.)).to(self.device).permute(0, 3, 1, 2)
        with torch.no_grad():
            heat, feat = self.detector.forward_nhwc(x)
        return decode_peaks(heat, feat, tiles, cut_off, max_boxes), heat, feat
