"""Page-level detector pipeline on the MI355X path: tiling front-end, batched tiles through the HIP
detector, GPU peak decode, GPU paste of the page maps, host-side page merge.

Drop-in for ``OCR_Processer.run_detector`` (``/root/reference/process_ocr_base.py:474-650``) and for the
tiling part of ``call_OCR`` (``:57-78``): same arguments, same four return values
``(locations float32 [M,9], glyphfeatures float32 [M,100], lines_all, seps_all)``.

What runs where
* GPU (HIP): tile gather from the uint8 page (``ftc_tile_gather``), detector forward + NMS for batches of
  tiles (the reference runs them one by one), peak decode + feature gather (``ftc_decode``), the
  ``np.maximum`` paste of the masked sigmoid maps (``ftc_paste_maps``).  Only the decoded peaks and the
  page-sized canvases come back over PCIe -- not 16 MB of maps per tile.
  The page-level selection -- 2-means contrast filter, greedy overlap suppression with the coverage rule,
  separator filter, 3x3 code maximum (``process_ocr_base.py:540-648``) -- runs on the GPU
  too (``ftc_box_hists`` + ``ftc_page_merge``: float64, sequential semantics kept, bit-identical results);
  only the selected boxes, their feature rows and the two page canvases come back over PCIe.
  (There is no host implementation of the selection in this package: the NumPy restatement the GPU kernels are checked
  against is test infrastructure, ``oracle/decode_oracle.py``.)
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .decode import TileGeom, decode_peaks, tile_keep_rect, tiles_to_device
from .schema import feature_dim, height, scale, width


# ------------------------------------------------------------------------------------------------
# tiling (process_ocr_base.py:63-76)
# ------------------------------------------------------------------------------------------------
def padded_page_size(h: int, w: int, stepx: int, stepy: int) -> Tuple[int, int]:
    padx = max(0, (width - w) % stepx, width - w)
    pady = max(0, (height - h) % stepy, height - h)
    return h + pady, w + padx


def tile_origins(page_h: int, page_w: int, stepx: int, stepy: int) -> List[Tuple[int, int]]:
    return [(y, x) for y in range(0, page_h - height + 1, stepy) for x in range(0, page_w - width + 1, stepx)]


# ------------------------------------------------------------------------------------------------
# page-level selection on the GPU (process_ocr_base.py:540-648)
# ------------------------------------------------------------------------------------------------
def page_merge_gpu(boxes: torch.Tensor, feats: torch.Tensor, page, canv: torch.Tensor, cut_off: float, variant: str = "production",
                   seed_start: int = -1, seed_scale: float = 1.0):
    """GPU page-level selection.  boxes [N,9] fp32 (rows with p < cut_off are inert, e.g. the zero padding of
    ``decode_peaks``), feats [N,C] fp32, page [H,W,3] fp32 0..255, canv [7,mh,mw] fp32 (``ftc_paste_maps``), all on the GPU.
    Returns device tensors (locations [M,9] fp32, glyphfeatures [M,C] fp32); one host sync (the kept count).

    ``variant="demo"``: the selection of the demo script's ``eval()`` (``/root/reference/test_image1_torch.py:152-240``: no contrast filter --
    ``page`` may be a ``(page_h, page_w)`` tuple --, its ``fill_map`` offsets) with rows ``[seed_start, N)`` = the UNSCALED boxes of a coarse
    first pass, multiplied by ``seed_scale`` in float64 inside the kernels (``:313-332``); returns (locations float64 [M,9] numpy -- eval()'s own
    result rows: seed rows scaled, ``max(code maximum, code)`` in float64 --, glyphfeatures [M,C] device tensor)."""
    lib = L.load()
    dev = boxes.device
    boxes = boxes.contiguous()
    N = boxes.shape[0]
    demo = variant == "demo"
    if variant not in ("production", "demo"):
        raise ValueError("variant must be 'production' or 'demo'")
    ph, pw = (page if demo and isinstance(page, tuple) else page.shape[:2])
    mh, mw = canv.shape[1:]
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if demo:
            hist = torch.zeros((2, N), dtype=torch.float64, device=dev)            # (no contrast filter: ftc_page_order only ranks)
        else:
            hist = torch.empty((2, N), dtype=torch.float64, device=dev)
            L.check(lib.ftc_box_hists(boxes.data_ptr(), N, page.data_ptr(), ph, pw, C.c_float(cut_off), hist.data_ptr(), stream), "ftc_box_hists")
        # order = the rows with p >= cut_off in stable score order (then the rest), threshold = np.median(hists) / 5 over the rows with p >= cut_off -- in-tree kernels (rank by
        # counting, radix select), results identical to torch.sort / np.median
        order = torch.empty((N,), dtype=torch.int32, device=dev)
        th = torch.empty((1,), dtype=torch.float64, device=dev)
        ob = int(lib.ftc_page_order_scratch_bytes(N))
        osc = torch.empty(ob, dtype=torch.uint8, device=dev)
        L.check(lib.ftc_page_order(boxes.data_ptr(), N, hist[0].data_ptr(), C.c_float(cut_off), order.data_ptr(), th.data_ptr(), osc.data_ptr(), ob, stream),
                "ftc_page_order")
        out_loc = torch.empty((N, 9), dtype=torch.float32, device=dev)
        out_idx = torch.empty((N,), dtype=torch.int32, device=dev)
        out_n = torch.zeros((1,), dtype=torch.int32, device=dev)
        nbytes = int(lib.ftc_page_merge_scratch_bytes(N, ph, pw))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        codes = canv[3:7].contiguous()
        if not demo:
            L.check(lib.ftc_page_merge(boxes.data_ptr(), order.data_ptr(), N, hist[1].data_ptr(), th.data_ptr(), C.c_float(cut_off),
                                       canv[2].data_ptr(), codes.data_ptr(), mh, mw, scale, ph, pw, out_loc.data_ptr(), out_idx.data_ptr(),
                                       out_n.data_ptr(), scratch.data_ptr(), nbytes, stream), "ftc_page_merge")
        else:
            out_cm = torch.empty((N, 4), dtype=torch.float32, device=dev)
            L.check(lib.ftc_page_merge_variant(boxes.data_ptr(), order.data_ptr(), N, None, None, C.c_float(cut_off), canv[2].data_ptr(), codes.data_ptr(),
                                               mh, mw, scale, ph, pw, 1, int(seed_start), C.c_double(seed_scale), out_loc.data_ptr(), out_idx.data_ptr(),
                                               out_cm.data_ptr(), out_n.data_ptr(), scratch.data_ptr(), nbytes, stream), "ftc_page_merge_variant")
        n = int(out_n.item())
        if n < 0:
            raise RuntimeError("ftc_page_merge: a box is larger than the page-sized coverage bitmap")
        sel = out_idx[:n].long()
        if not demo:
            return out_loc[:n], feats.index_select(0, sel)
        # eval()'s float64 rows (test_image1_torch.py:147-151, :233): the selected INPUT rows, seed rows scaled in float64, codes = max(3x3 maximum, code)
        rows = boxes.index_select(0, sel).cpu().numpy().astype(np.float64)
        src = sel.cpu().numpy()
        if 0 <= seed_start < N:
            seeded = src >= seed_start
            rows[seeded, 1:] = rows[seeded, 1:] * float(seed_scale)
        cm = out_cm[:n].cpu().numpy().astype(np.float64)
        upd = np.isfinite(cm[:, 0])
        rows[upd, 5:9] = np.where(rows[upd, 5:9] > cm[upd], rows[upd, 5:9], cm[upd])          # max(a, b) of Python: b only if b > a
        return rows, feats.index_select(0, sel)


# ------------------------------------------------------------------------------------------------
# the pipeline
# ------------------------------------------------------------------------------------------------
class PageDetector:
    """``run_detector`` / tiling of ``OCR_Processer`` on top of a HIP ``CenterNetDetector``."""

    def __init__(self, detector, step_ratio: float = 0.6, cut_off: float = 0.4, batch: int = 8, max_boxes: int = 4096,
                 device: str = "cuda", group=None, shard: bool = True, lanes: int = 2, variant: str = "production", twopass: bool = False):
        """variant="demo": the tiling, border margins and page-level selection of the demo script instead of the production class --
        ``/root/reference/test_image1_torch.py``: step = 3/4 of a tile (``:298-299``), 1/8 margins (``:103-108``), ``eval()``'s selection
        (``:152-240``; float64 result rows); ``twopass`` (its command-line switch, ``:313-332``): a page of more than 2 steps is first run
        shrunk onto ONE tile and the boxes found there join the candidates of the full-resolution pass, scaled back."""
        if variant not in ("production", "demo"):
            raise ValueError("variant must be 'production' or 'demo'")
        if twopass and variant != "demo":
            raise ValueError("twopass is the demo script's mode: variant='demo'")
        self.variant, self.twopass = variant, bool(twopass)
        self.device = torch.device(device)
        detector.to(device=self.device)
        detector.eval()
        self.detector = detector
        self.step_ratio, self.cut_off, self.batch, self.max_boxes = step_ratio, cut_off, batch, max_boxes
        # With torch.distributed initialised the tiles OF ONE PAGE are split over the ranks (dist.shard_range: contiguous blocks), the
        # record rows are gathered (one collective, no host sync) and the page canvases merged with an all-reduce(MAX): every rank
        # ends up with the whole page's boxes.  shard=False: every rank runs the whole page (replicas).
        self.group, self.shard = group, shard
        # The batches OF ONE PAGE alternate over `lanes` HIP streams (see lanes.py: batch k+1's backbone under batch k's FPN heads), each
        # with its own activation arena; the page canvas is shared (ftc_paste_maps merges with atomicMax: order-independent).  Results do
        # not depend on `lanes`.
        self.lanes = max(1, int(lanes))
        self._lane_streams, self._lane_ws = None, {}
        self._row_hint = None                                  # rows per tile the last page needed (multi-GPU gather of large blocks)
        self.stepx, self.stepy = int(width * step_ratio), int(height * step_ratio)      # process_ocr_base.py:43-45
        if variant == "demo":
            self.stepx, self.stepy = width * 3 // 4, height * 3 // 4                     # test_image1_torch.py:298-299

    # -- reference signature -----------------------------------------------------------------
    def run_detector(self, ds: Sequence[dict], org_img: np.ndarray):
        """ds: list of {'input': [1,768,768,3] float32 0..255, 'offsetx', 'offsety'}; org_img float32 page."""
        tiles = np.concatenate([np.asarray(d["input"], dtype=np.float32) for d in ds], axis=0)
        origins = [(d["offsety"], d["offsetx"]) for d in ds]
        x_all = torch.from_numpy(tiles / np.float32(255.)).to(self.device)
        org = np.ascontiguousarray(org_img)
        return self._run(lambda lo, hi: x_all[lo:hi], origins, org.shape[:2],
                         lambda: torch.from_numpy(org).to(self.device).float() if org.dtype == np.uint8 else torch.from_numpy(np.asarray(org, np.float32)).to(self.device))

    def detect_page(self, im_u8: np.ndarray):
        """uint8 RGB page [H,W,3] -> same outputs; pads with white and tiles like call_OCR (:63-76)."""
        lib = L.load()
        h0, w0 = im_u8.shape[:2]
        ph, pw = padded_page_size(h0, w0, self.stepx, self.stepy)
        origins = tile_origins(ph, pw, self.stepx, self.stepy)
        # One upload of the raw page; the white padding (:63-65) exists only on the GPU: the tile gather reads beyond the page as 255 and the
        # padded fp32 page the contrast filter looks at is formed there too (round 4: the host-side pad + second upload cost 6 ms per page).
        page_dev = torch.from_numpy(np.ascontiguousarray(im_u8[:, :, :3])).to(self.device)
        o_all = torch.tensor(origins, dtype=torch.int32, device=self.device).reshape(-1, 2)      # ONE upload: nothing inside the batch loop may wait for the GPU

        def gather(lo, hi):
            out = torch.empty((hi - lo, height, width, 3), dtype=torch.float32, device=self.device)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            L.check(lib.ftc_tile_gather(page_dev.data_ptr(), h0, w0, o_all[lo:hi].data_ptr(), hi - lo, height, width, out.data_ptr(),
                                        C.c_void_p(stream)), "ftc_tile_gather")
            return out

        def padded_page():
            full = torch.full((ph, pw, 3), 255.0, dtype=torch.float32, device=self.device)
            full[:h0, :w0] = page_dev.float()
            return full
        seeds = None
        if self.twopass and (pw / self.stepx > 2 or ph / self.stepy > 2):
            seeds = self._coarse_pass(im_u8, ph, pw)
        return self._run(gather, origins, (ph, pw), padded_page, seeds=seeds)

    def _coarse_pass(self, im_u8: np.ndarray, ph: int, pw: int):
        """The first pass of the demo script's two-pass mode (test_image1_torch.py:313-332): the white-padded page shrunk by
        s = max(W, H) / 768 with PIL's bilinear filter (the reference's own resampler: host code there as well), padded to one tile, through the
        same detector and the demo selection (cut_off 0.4).  Returns (locations0 UNSCALED fp32 [K,9] on the GPU, glyphfeatures0 [K,C], s)."""
        from PIL import Image
        im0 = np.full((ph, pw, 3), 255, np.uint8)
        im0[:im_u8.shape[0], :im_u8.shape[1]] = im_u8[:, :, :3]
        s_ = max(im0.shape[1], im0.shape[0]) / max(width, height)
        im1 = np.asarray(Image.fromarray(im0).resize((int(im0.shape[1] / s_), int(im0.shape[0] / s_)), resample=Image.BILINEAR))
        px, py = max(0, width - im1.shape[1]), max(0, height - im1.shape[0])
        im1 = np.pad(im1, [[0, py], [0, px], [0, 0]], "constant", constant_values=255)
        x1 = torch.from_numpy((im1.astype(np.float32) / np.float32(255.))[None]).to(self.device)
        loc0, gf0, _, _ = self._run(lambda lo, hi: x1[lo:hi], [(0, 0)], im1.shape[:2], None, cut_off=0.4, world1=True)
        if loc0.shape[0] and not np.array_equal(loc0.astype(np.float32).astype(np.float64), loc0):
            raise RuntimeError("coarse-pass rows are not fp32 values")              # (cannot happen: no seed rows in the coarse pass)
        return torch.from_numpy(loc0.astype(np.float32)).to(self.device), torch.from_numpy(gf0).to(self.device), float(s_)

    # -- shared ------------------------------------------------------------------------------------
    def _run(self, get_tiles, origins, page_hw, page_f32, seeds=None, cut_off=None, world1=False):
        """get_tiles(lo, hi) -> [n,768,768,3] fp32 0..1 on the GPU; page_hw = (padded) page size; page_f32() -> the padded fp32 page on the GPU.
        seeds (demo variant) = (locations0 fp32 [K,9] unscaled, glyphfeatures0 [K,C], scale) of a coarse pass; world1: no sharding (the coarse pass)."""
        lib = L.load()
        cut = self.cut_off if cut_off is None else cut_off
        demo = self.variant == "demo"
        page_h, page_w = page_hw
        mh, mw = page_h // scale, page_w // scale
        canv = torch.zeros((7, mh, mw), dtype=torch.float32, device=self.device)
        parts = []
        import torch.distributed as tdist
        from .dist import all_gather_boxes_static, shard_range
        world = tdist.get_world_size(self.group) if (self.shard and not world1 and tdist.is_available() and tdist.is_initialized()) else 1
        first, last = shard_range(len(origins), tdist.get_rank(self.group), world) if world > 1 else (0, len(origins))
        n_batches = (last - first + self.batch - 1) // self.batch if last > first else 0
        n_lanes = min(self.lanes, max(1, n_batches))
        with torch.cuda.device(self.device), torch.no_grad():
            main = torch.cuda.current_stream(self.device)
            if n_lanes > 1:
                if self._lane_streams is None or len(self._lane_streams) < n_lanes:
                    self._lane_streams = [torch.cuda.Stream(device=self.device) for _ in range(n_lanes)]
                streams = self._lane_streams[:n_lanes]
                eng = self.detector.detector._engine
                eng.ensure_model(self.device)
                need = eng.model.workspace_bytes(self.batch, height, width)
                for i in range(n_lanes):
                    if i not in self._lane_ws or self._lane_ws[i].numel() < need:
                        self._lane_ws[i] = torch.empty(need, dtype=torch.uint8, device=self.device)
            else:
                streams = [main]
            # every tile's geometry record in ONE upload, before any forward is enqueued: a host-to-device copy from pageable memory waits for
            # the stream it is issued on, and inside the loop that was the forward just enqueued (round 4: 2-4 ms of GPU idle per batch)
            geoms = [TileGeom(ox, oy, page_w, page_h, tile_keep_rect(ox, oy, page_w, page_h, None if demo else self.step_ratio)) for (oy, ox) in origins[first:last]]
            tl_all = tiles_to_device(geoms, self.device, height // scale, width // scale) if geoms else None
            if n_lanes > 1:
                for s_ in streams:
                    s_.wait_stream(main)
            for k, lo in enumerate(range(first, last, self.batch)):
                hi = min(last, lo + self.batch)
                lane = k % n_lanes
                with torch.cuda.stream(streams[lane]):
                    x = get_tiles(lo, hi).permute(0, 3, 1, 2)
                    heat, feat = self.detector.forward_nhwc(x, workspace=self._lane_ws[lane] if n_lanes > 1 else None)
                    tl = tl_all[lo - first:hi - first]
                    stream = torch.cuda.current_stream(self.device).cuda_stream
                    L.check(lib.ftc_paste_maps(heat.data_ptr(), tl.data_ptr(), hi - lo, heat.shape[1], heat.shape[2], scale, canv.data_ptr(),
                                               mh, mw, C.c_void_p(stream)), "ftc_paste_maps")
                    dec = decode_peaks(heat, feat, tl, cut_off=cut, max_boxes=self.max_boxes)
                    parts.append((dec.counts, dec.boxes, dec.feats, dec.records))
            if n_lanes > 1:
                for s_ in streams:
                    main.wait_stream(s_)
            # every tile's rows in tile order; rows past a tile's count are zeros (p = 0 < cut_off): inert padding
            if world > 1:
                n_feat = parts[0][2].shape[-1] if parts else 100
                cnt_l = torch.cat([c for c, _, _, _ in parts]) if parts else torch.zeros(0, dtype=torch.int32, device=self.device)
                rec_l = torch.cat([r for _, _, _, r in parts]) if parts else torch.zeros((0, self.max_boxes, 112), dtype=torch.float32, device=self.device)
                # small blocks travel whole; above STATIC_GATHER_BYTES only the rows the last page needed (+25 %), and the device
                # overflow flag -- identical on every rank, it is computed from the gathered counts -- sends the whole block after all
                # (the choice is made inside all_gather_boxes_static from the GLOBAL tile count: shards are uneven, a choice on the local block
                # size would let two ranks disagree about the message size -- round-4 advisor finding)
                g = all_gather_boxes_static(cnt_l, rec_l, len(origins), group=self.group, row_hint=self._row_hint)
                if g.records.shape[1] < self.max_boxes and bool(g.overflow):
                    g = all_gather_boxes_static(cnt_l, rec_l, len(origins), group=self.group)
                tdist.all_reduce(canv, op=tdist.ReduceOp.MAX, group=self.group)
                counts = g.counts
                boxes = g.records[:, :, :9].reshape(-1, 9)
                fts = g.records[:, :, g.feat0:g.feat0 + n_feat].reshape(-1, n_feat)
            else:
                counts = torch.cat([c for c, _, _, _ in parts])
                boxes = torch.cat([b.reshape(-1, 9) for _, b, _, _ in parts])
                fts = torch.cat([f.reshape(-1, f.shape[-1]) for _, _, f, _ in parts])
            if demo:
                n_tile_rows = boxes.shape[0]
                if seeds is not None and seeds[0].shape[0]:
                    boxes, fts = torch.cat([boxes, seeds[0]]), torch.cat([fts, seeds[1]])
                loc_d, glyph_d = page_merge_gpu(boxes, fts, (page_h, page_w), canv, cut, variant="demo", seed_start=n_tile_rows,
                                                seed_scale=seeds[2] if seeds is not None else 1.0)
            else:
                loc_d, glyph_d = page_merge_gpu(boxes, fts, page_f32(), canv, cut)
            cmax = int(counts.max().item())
            if cmax > self.max_boxes:
                raise RuntimeError(f"a tile produced {cmax} peaks > max_boxes={self.max_boxes}; raise max_boxes")
            if world > 1 or not world1:
                # next page's gather: rows sent when the block is large.  Every rank must hold the SAME value (it sizes an RCCL message): only
                # counts that came out of a gather -- or a run in which every rank saw every tile -- may set it, never the unsharded coarse
                # pass of the two-pass mode, whose per-rank results are only equal if the GPUs agree bit for bit.
                self._row_hint = min(self.max_boxes, (cmax + cmax // 4 + 64) // 64 * 64)
            canv_h = canv[1:3].cpu().numpy()
            return (loc_d if demo else loc_d.cpu().numpy()), glyph_d.cpu().numpy(), canv_h[0], canv_h[1]


def linedetect_request(locations: np.ndarray, lines: np.ndarray, seps: np.ndarray) -> bytes:
    """Binary request of the reference's ``linedetect`` CLI (``process_ocr_base.py:80-88``; parser
    ``textline_detect/src/main.cpp:100-180``): u32 mode=0, u32 w, u32 h, f32 lines[h*w], f32 seps[h*w],
    u32 n, n x 8 f32 (cx, cy, w, h, code1, code2, code4, code8)."""
    h, w = lines.shape
    out = int(0).to_bytes(4, "little") + int(w).to_bytes(4, "little") + int(h).to_bytes(4, "little")
    out += np.ascontiguousarray(lines, dtype=np.float32).tobytes() + np.ascontiguousarray(seps, dtype=np.float32).tobytes()
    out += int(locations.shape[0]).to_bytes(4, "little")
    out += np.ascontiguousarray(locations[:, 1:], dtype=np.float32).tobytes()
    return out


def linedetect_parse(result: bytes):
    """i32 n, n x 7 i32 (id, block, idx, subidx, subtype, page, section) -- process_ocr_base.py:91-112."""
    n = int.from_bytes(result[:4], "little")
    a = np.frombuffer(result, dtype="<i4", count=7 * n, offset=4).reshape(n, 7)
    return [tuple(int(v) for v in row) for row in a]
