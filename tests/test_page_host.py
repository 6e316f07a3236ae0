"""CPU: the oracle's page merge (oracle/decode_oracle.py, the checker of the GPU page kernels) and the tiling rule / linedetect
wire format of findtextcenternet_amd.page against the reference's own outputs (tests/golden/g3_decode_*.npz, written by
OCR_Processer.run_detector)."""
import os

import numpy as np

import synth
from findtextcenternet_amd import page
from oracle import decode_oracle

G = os.path.join(os.path.dirname(__file__), "golden")


def _tiles_2x2():
    step = int(768 * 0.6)
    img = synth.page_uint8(32, 768 + step, 768 + step).astype(np.float32)
    origins = [(0, 0), (0, step), (step, 0), (step, step)]
    maps = [synth.detector_maps(200 + n) for n in range(4)]
    return img, origins, maps


def test_page_merge_matches_reference_run_detector():
    g = np.load(os.path.join(G, "g3_decode_2x2.npz"))
    img, origins, maps = _tiles_2x2()
    ph, pw = img.shape[:2]
    canv = [np.zeros([ph // 4, pw // 4], np.float32) for _ in range(7)]
    locs, feats = [np.zeros([1, 9])], [np.zeros([1, 100], np.float32)]
    for (y, x), (hm, ft) in zip(origins, maps):
        rect = decode_oracle.tile_keep_rect(x, y, pw, ph, 0.6)
        decode_oracle.paste_maps(canv, hm, x, y, rect)                 # per-tile stages from the pinned oracle
        l, f, _ = decode_oracle.decode_tile(hm, ft, x, y, pw, ph, 0.4, rect)
        locs.append(l)
        feats.append(f)
    loc, gf = decode_oracle.page_merge(np.concatenate(locs), np.concatenate(feats), img, canv[2], canv[3:], 0.4)
    assert loc.shape == g["locations"].shape and loc.shape[0] > 100
    assert np.array_equal(loc, g["locations"]) and np.array_equal(gf, g["glyphfeatures"])


def test_page_merge_single_tile_and_empty():
    g = np.load(os.path.join(G, "g3_decode_single.npz"))
    img = synth.page_uint8(31, 768, 768).astype(np.float32)
    hm, ft = synth.detector_maps(101)
    rect = decode_oracle.tile_keep_rect(0, 0, 768, 768, 0.6)
    canv = [np.zeros([192, 192], np.float32) for _ in range(7)]
    decode_oracle.paste_maps(canv, hm, 0, 0, rect)
    l, f, _ = decode_oracle.decode_tile(hm, ft, 0, 0, 768, 768, 0.4, rect)
    loc, gf = decode_oracle.page_merge(np.concatenate([np.zeros([1, 9]), l]), np.concatenate([np.zeros([1, 100], np.float32), f]), img,
                              canv[2], canv[3:], 0.4)
    assert np.array_equal(loc, g["locations"]) and np.array_equal(gf, g["glyphfeatures"])
    loc, gf = decode_oracle.page_merge(np.zeros([1, 9]), np.zeros([1, 100], np.float32), img, canv[2], canv[3:], 0.4)
    assert loc.shape == (0, 9) and gf.shape == (0, 100)


def test_tiling_matches_reference_rule():
    # img/test1.png is 533x640 -> one 768x768 tile; 2358x1030 -> 8 tiles at stride 576, 10 at 460
    assert page.padded_page_size(640, 533, 460, 460) == (768, 768)
    for (h, w, step, n) in [(1030, 2358, 576, 8), (1030, 2358, 460, 10), (640, 533, 460, 1)]:
        ph, pw = page.padded_page_size(h, w, step, step)
        assert (ph - 768) % step == 0 and (pw - 768) % step == 0 and ph >= h and pw >= w
        assert len(page.tile_origins(ph, pw, step, step)) == n


def test_linedetect_wire_format_roundtrip():
    g = np.load(os.path.join(G, "g3_decode_single.npz"))
    req = page.linedetect_request(g["locations"], g["lines"], g["seps"])
    h, w = g["lines"].shape
    n = g["locations"].shape[0]
    assert len(req) == 12 + 2 * 4 * h * w + 4 + 8 * 4 * n
    assert int.from_bytes(req[:4], "little") == 0 and int.from_bytes(req[4:8], "little") == w and int.from_bytes(req[8:12], "little") == h
    off = 12 + 2 * 4 * h * w
    assert int.from_bytes(req[off:off + 4], "little") == n
    assert np.array_equal(np.frombuffer(req, np.float32, 8 * n, off + 4).reshape(n, 8), g["locations"][:, 1:])
    reply = (3).to_bytes(4, "little") + np.arange(21, dtype="<i4").tobytes()
    assert page.linedetect_parse(reply) == [tuple(range(0, 7)), tuple(range(7, 14)), tuple(range(14, 21))]
