"""-m gpu: ftc_decode (select + rank + gather) against the numpy oracle of the reference's host
loop (oracle/decode_oracle.py, pinned by tests/golden/g3_*.npz) -- index sets and order bit-exact,
box values within 2e-6 relative (GPU tanhf/expf vs numpy)."""
import os

import numpy as np
import pytest
import torch

import synth
from findtextcenternet_amd import TileGeom, decode_peaks, exact_logit_cut, tile_keep_rect
from oracle import decode_oracle

pytestmark = pytest.mark.gpu


def _nhwc(a):
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 3, 1))).cuda()


def _check_against_oracle(hm, ft, geoms, cut_off, max_boxes=4096):
    dec = decode_peaks(_nhwc(hm), _nhwc(ft), geoms, cut_off=cut_off, max_boxes=max_boxes)
    counts = dec.counts.cpu().numpy()
    for b, gm in enumerate(geoms):
        loc, gf, idx = decode_oracle.decode_tile(hm[b:b + 1], ft[b:b + 1], gm.offset_x, gm.offset_y, gm.page_w, gm.page_h, cut_off, gm.rect)
        assert counts[b] == len(idx), (b, counts[b], len(idx))
        n = min(len(idx), max_boxes)
        gi = dec.index[b, :n].cpu().numpy()
        assert set(gi.tolist()) == set(idx[:n].tolist()) or n < len(idx)
        # order: GPU = (logit desc, index asc); oracle = (sigmoid desc, index asc).  Identical wherever
        # the float32 sigmoid is strictly decreasing; inside a sigmoid tie group only the set must agree.
        p = loc[:n, 0]
        start = 0
        while start < n:
            end = start
            while end + 1 < n and p[end + 1] == p[start]:
                end += 1
            assert sorted(gi[start:end + 1].tolist()) == sorted(idx[start:end + 1].tolist())
            start = end + 1
        gb = dec.boxes[b, :n].cpu().numpy()
        gfe = dec.feats[b, :n].cpu().numpy()
        order = {int(v): k for k, v in enumerate(idx[:n])}
        perm = np.array([order[int(v)] for v in gi], dtype=np.int64)
        np.testing.assert_allclose(gb, loc[perm].astype(np.float32), rtol=2e-6, atol=1e-7)
        assert np.array_equal(gfe, gf[perm])                      # gathered rows are copies: exact
        assert (dec.index[b, n:] == -1).all()
    return dec


def test_single_tile_vs_oracle():
    hm, ft = synth.detector_maps(101)
    g = TileGeom(0, 0, 768, 768, tile_keep_rect(0, 0, 768, 768, 0.6))
    dec = _check_against_oracle(hm, ft, [g], 0.4)
    assert int(dec.counts[0]) > 100


def test_batch_of_tiles_with_margins_and_offsets():
    step = int(768 * 0.6)
    page = 768 + step
    maps = [synth.detector_maps(200 + n) for n in range(4)]
    hm = np.concatenate([m[0] for m in maps])
    ft = np.concatenate([m[1] for m in maps])
    geoms = [TileGeom(x, y, page, page, tile_keep_rect(x, y, page, page, 0.6)) for (y, x) in [(0, 0), (0, step), (step, 0), (step, step)]]
    _check_against_oracle(hm, ft, geoms, 0.4)
    # demo-script margins (1/8) and its cut-off variants
    geoms = [TileGeom(x, y, page, page, tile_keep_rect(x, y, page, page, None)) for (y, x) in [(0, 0), (0, step), (step, 0), (step, step)]]
    _check_against_oracle(hm, ft, geoms, 0.35)


def test_sparse_golden_fixture(golden_dir):
    """Against the reference's own run_detector output on the fixture where its page-level
    suppression removes nothing (tests/golden/g3_decode_sparse.npz)."""
    g = np.load(os.path.join(golden_dir, "g3_decode_sparse.npz"))
    hm = g["heatmap"]
    feat = np.random.Generator(np.random.PCG64(303)).standard_normal((1, 100, 192, 192)).astype(np.float32)
    geom = TileGeom(0, 0, 768, 768, tile_keep_rect(0, 0, 768, 768, 0.6))
    dec = decode_peaks(_nhwc(hm), _nhwc(feat), [geom], cut_off=0.4)
    n = int(dec.counts[0])
    ref = g["locations"]
    assert n == ref.shape[0]
    np.testing.assert_allclose(dec.boxes[0, :n, :5].cpu().numpy(), ref[:, :5], rtol=2e-6, atol=1e-7)
    assert np.array_equal(dec.feats[0, :n].cpu().numpy(), g["glyphfeatures"])
    idx = dec.index[0, :n].cpu().numpy().tolist()
    assert 6 * 192 + 6 in idx and 6 * 192 + 18 not in idx            # 1e-3 above / below the cut-off
    assert 30 * 192 + 30 not in idx and 42 * 192 + 42 not in idx     # w > page, h underflow


def test_truncation_empty_and_ties():
    hm, ft = synth.detector_maps(7)
    g = TileGeom(0, 0, 768, 768, (0, 192, 0, 192))
    full = decode_peaks(_nhwc(hm), _nhwc(ft), [g], cut_off=0.4)
    n = int(full.counts[0])
    small = decode_peaks(_nhwc(hm), _nhwc(ft), [g], cut_off=0.4, max_boxes=64)
    assert int(small.counts[0]) == n > 64                               # total is reported, rows are top-64
    assert torch.equal(small.index[0], full.index[0, :64]) and torch.equal(small.boxes[0], full.boxes[0, :64])
    empty = hm.copy()
    empty[:, 1] = -np.inf
    d = decode_peaks(_nhwc(empty), _nhwc(ft), [g], cut_off=0.4)
    assert int(d.counts[0]) == 0 and (d.index == -1).all() and (d.boxes == 0).all() and (d.records == 0).all()
    tie = np.full((1, 10, 192, 192), -5.0, np.float32)
    tie[0, 1] = -np.inf
    for (y, x) in [(100, 7), (3, 150), (3, 20), (50, 50)]:
        tie[0, 1, y, x] = 2.5                                           # equal logits: index ascending
    tie[0, 1, 60, 60] = 3.0
    d = decode_peaks(_nhwc(tie), _nhwc(ft), [g], cut_off=0.4)
    assert d.index[0, :5].cpu().tolist() == [60 * 192 + 60, 3 * 192 + 20, 3 * 192 + 150, 50 * 192 + 50, 100 * 192 + 7]


def test_cut_off_boundary_is_exact():
    """A logit exactly at exact_logit_cut() is kept, its float32 predecessor is not -- the same
    decision numpy's float32 sigmoid makes against float32(cut_off)."""
    for cut in (0.4, 0.35, 0.5):
        t = np.float32(exact_logit_cut(cut))
        below = np.nextafter(t, np.float32(-np.inf))
        assert decode_oracle.sigmoid(np.array([t], np.float32))[0] >= np.float32(cut) > decode_oracle.sigmoid(np.array([below], np.float32))[0]
        hm = np.full((1, 10, 8, 8), 0.0, np.float32)
        hm[0, 1] = -np.inf
        hm[0, 1, 2, 2], hm[0, 1, 5, 5] = t, below
        ft = np.zeros((1, 100, 8, 8), np.float32)
        d = decode_peaks(_nhwc(hm), _nhwc(ft), [TileGeom(0, 0, 1000, 1000, (0, 8, 0, 8))], cut_off=cut)
        assert int(d.counts[0]) == 1 and int(d.index[0, 0]) == 2 * 8 + 2


def test_record_block_and_workspace_reuse():
    """boxes / feats are views of ONE [B, max, 112] record block (the multi-GPU gather message, written by the kernel itself),
    and a preallocated DecodeWorkspace gives the same rows as a fresh zero-filled one up to counts[b]."""
    from findtextcenternet_amd.decode import REC_FEAT0, REC_W, DecodeWorkspace
    hm = np.concatenate([synth.detector_maps(31)[0], synth.detector_maps(32)[0]])
    ft = np.concatenate([synth.detector_maps(31)[1], synth.detector_maps(32)[1]])
    g = [TileGeom(0, 0, 768, 768, tile_keep_rect(0, 0, 768, 768, 0.6))] * 2
    fresh = decode_peaks(_nhwc(hm), _nhwc(ft), g, cut_off=0.4, max_boxes=2048)
    assert fresh.records.shape == (2, 2048, REC_W) and fresh.boxes.data_ptr() == fresh.records.data_ptr()
    assert fresh.feats.data_ptr() == fresh.records.data_ptr() + 4 * REC_FEAT0 and (fresh.records[:, :, 9:REC_FEAT0] == 0).all()
    ws = DecodeWorkspace(2, 192, 192, 100, 2048, "cuda")
    ws.records.fill_(7.0)                                                # stale contents of an earlier call
    for _ in range(2):
        d = decode_peaks(_nhwc(hm), _nhwc(ft), g, cut_off=0.4, max_boxes=2048, workspace=ws)
        assert torch.equal(d.counts, fresh.counts)
        for b in range(2):
            n = int(d.counts[b])
            assert torch.equal(d.boxes[b, :n], fresh.boxes[b, :n]) and torch.equal(d.feats[b, :n], fresh.feats[b, :n])
            assert torch.equal(d.index[b, :n], fresh.index[b, :n])
    with pytest.raises(ValueError):
        decode_peaks(_nhwc(hm[:1]), _nhwc(ft[:1]), g[:1], cut_off=0.4, max_boxes=2048, workspace=ws)


@pytest.mark.parametrize("hw", [(240, 272), (5, 7), (193, 191), (64, 512)], ids=["240x272_two_passes", "5x7", "193x191", "64x512"])
def test_other_map_sizes_vs_oracle(hw):
    """Round 6 rewrote the select pass (16 workgroups per image, 12 positions per thread and pass, one atomic per workgroup): maps larger than
    192 x 192 take more than one pass per workgroup, small and odd-sized maps leave workgroups with ragged or empty ranges.  Whole-map and inner
    rectangles, against the oracle."""
    h, w = hw
    hm, ft = synth.detector_maps(400 + h, b=2, h=h, w=w)
    H, W = 4 * h, 4 * w
    geoms = [TileGeom(0, 0, W, H, (0, w, 0, h)), TileGeom(0, 0, 2 * W, 2 * H, tile_keep_rect(W // 2, H // 2, 2 * W, 2 * H, 0.6, tile_w=W, tile_h=H))]
    dec = _check_against_oracle(hm, ft, geoms, 0.4, max_boxes=8192)
    assert int(dec.counts[0]) > 0 or h * w < 100
