"""-m gpu: successive batches on several HIP streams (findtextcenternet_amd.lanes.DetectorLanes) give, batch for batch, exactly the
bytes the one-stream path gives -- maps, NMS slot, box counts and box records -- while several batches are in flight at once."""
import numpy as np
import pytest
import torch

import synth
from findtextcenternet_amd import DetectorLanes, TileGeom, decode_peaks, exact_logit_cut, tile_keep_rect, tiles_to_device
from gpu_harness import shared_detector

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision,lanes", [("bf16", 2), ("bf16", 3), ("fp16x3", 2)])
def test_lanes_are_bit_identical_to_one_stream(precision, lanes):
    dev = torch.device("cuda")
    det, _ = shared_detector(precision)
    B, S, n_batches = 2, 256, 7
    h = S // 4
    rect = tile_keep_rect(0, 0, S, S, 0.6, tile_w=S, tile_h=S)
    tiles = tiles_to_device([TileGeom(0, 0, S, S, rect) for _ in range(B)], dev, h, h)
    lcut = exact_logit_cut(0.4)
    xs = [torch.from_numpy(synth.noise_images(50 + k, B, S, S)).to(dev).permute(0, 3, 1, 2) for k in range(n_batches)]

    want = []
    for x in xs:                                                   # one stream, the engine's own arena
        with torch.no_grad():
            heat, feat = det.forward_nhwc(x)
        dec = decode_peaks(heat, feat, tiles, cut_off=0.4, max_boxes=512, logit_cut=lcut)
        n = dec.counts.cpu().numpy().copy()
        want.append((heat.cpu().numpy().copy(), feat.cpu().numpy().copy(), n, dec.records.cpu().numpy().copy(), dec.index.cpu().numpy().copy()))
    assert sum(int(w[2].sum()) for w in want) > 0

    ln = DetectorLanes(det, B, S, S, lanes=lanes, max_boxes=512, device=dev)
    got = [None] * n_batches
    pending = []

    def collect(k, lane):
        ln.wait(lane)
        torch.cuda.current_stream().synchronize()
        got[k] = (ln.heat[lane].cpu().numpy().copy(), ln.feat[lane].cpu().numpy().copy(), ln.dws[lane].counts.cpu().numpy().copy(),
                  ln.dws[lane].records.cpu().numpy().copy(), ln.dws[lane].index.cpu().numpy().copy())

    for k, x in enumerate(xs):                                     # `lanes` batches in flight; a lane is read before it is reused
        if len(pending) == lanes:
            collect(*pending.pop(0))
        lane, _ = ln.submit(x, tiles, cut_off=0.4, logit_cut=lcut)
        pending.append((k, lane))
    while pending:
        collect(*pending.pop(0))

    for k in range(n_batches):
        wh, wf, wn, wr, wi = want[k]
        gh, gf, gn, gr, gi = got[k]
        assert np.array_equal(wh, gh, equal_nan=True) and np.array_equal(wf, gf, equal_nan=True), f"batch {k}: maps differ"
        assert np.array_equal(wn, gn), f"batch {k}: counts differ"
        for b in range(B):
            ow, og = np.argsort(wi[b, :wn[b]], kind="stable"), np.argsort(gi[b, :gn[b]], kind="stable")
            assert np.array_equal(wi[b, :wn[b]][ow], gi[b, :gn[b]][og])
            assert np.array_equal(wr[b, :wn[b]][ow].view(np.uint32), gr[b, :gn[b]][og].view(np.uint32)), f"batch {k} tile {b}: records differ"


def test_lane_workspace_is_checked():
    dev = torch.device("cuda")
    det, _ = shared_detector("bf16")
    x = torch.zeros((1, 3, 128, 128), device=dev)
    with pytest.raises(ValueError):
        det.forward_nhwc(x, workspace=torch.empty(1024, dtype=torch.uint8, device=dev))
