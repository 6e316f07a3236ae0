"""-m gpu: the full detector forward + NMS (+ decode) on MI355X through the drop-in modules,
against the golden vectors written by the reference's own code and against the CPU oracle.

Tolerance (BASELINE.json north_star): fp32 mode within 1e-3 absolute of the reference CPU
detector on maps of O(1-10) magnitude, -inf (suppressed) positions identical; bf16 mode has its
own relative gate (the reference itself under CPU bf16 autocast is off by ~3 % of range and shares
only ~93 % of its peaks with its fp32 run -- SURVEY.md Appendix C).
"""
import os

import numpy as np
import pytest
import torch

import synth
from findtextcenternet_amd import (CenterNetDetector, HipDetectorBackend, TextDetectorModel, TileGeom, decode_peaks,
                                   deterministic_state_dict, tile_keep_rect)
from oracle import decode_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _log(msg):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/test_detector.log", "a") as f:
        f.write(msg + "\n")
    print(msg)


@pytest.fixture(scope="module")
def sd():
    return deterministic_state_dict(0)


@pytest.fixture(scope="module")
def det_fp32(sd):
    m = TextDetectorModel(pre_weights=False, precision="fp32")
    m.load_state_dict(sd)
    d = CenterNetDetector(m.detector)
    d.to(device="cuda")
    d.eval()
    return d


@pytest.fixture(scope="module")
def det_bf16(sd):
    m = TextDetectorModel(pre_weights=False, precision="bf16")
    m.load_state_dict(sd)
    d = CenterNetDetector(m.detector)
    d.to(device="cuda")
    d.eval()
    return d


def _nms_margin(hm_ref, b, y, x):
    """|key - best other neighbour| in the reference map: how close the keep/suppress call was."""
    k = hm_ref[b, 0]
    h, w = k.shape
    win = [k[yy, xx] for yy in range(max(0, y - 1), min(h, y + 2)) for xx in range(max(0, x - 1), min(w, x + 2)) if (yy, xx) != (y, x)]
    return abs(float(k[y, x]) - max(win))


def _compare_maps(tag, hm, ft, g_hm, g_ft=None, tol=TOL):
    """L-inf on finite entries < tol; the -inf (suppressed) pattern of the NMS channel must be
    identical wherever the reference's own keep/suppress call had a margin above MARGIN = 1e-4
    (20x the fp32 summation-order noise we measure, 10x below tol): in flat background regions
    (white page padding) neighbouring key logits differ by ~1e-6, so which of them "wins" the 3x3
    window is decided by the last fp32 bit in the reference itself (3e-5
    between 4 and 8 CPU threads)."""
    MARGIN = 1e-4
    fin_ref, fin = np.isfinite(g_hm), np.isfinite(hm)
    both = fin & fin_ref
    e_hm = float(np.abs(hm[both] - g_hm[both]).max())
    mism = np.argwhere(fin != fin_ref)
    assert (mism[:, 1] == 1).all() if len(mism) else True          # only the NMS channel can hold -inf
    margins = np.array([_nms_margin(g_hm, b, y, x) for b, c, y, x in mism])
    above = sum(1 for b, c, y, x in mism if g_hm[b, 0, y, x] > np.log(0.4 / 0.6))
    e_ft = float(np.abs(ft - g_ft).max()) if g_ft is not None else None
    _log(f"{tag}: heatmap Linf {e_hm:.3e}  features Linf {e_ft}  NMS-mask flips {len(mism)} of {int(fin_ref[:, 1].size)} px "
         f"(max ref margin {margins.max() if len(mism) else 0:.1e}, {above} of them above the 0.4 cut-off)  "
         f"hm range [{g_hm[fin_ref].min():.2f},{g_hm[fin_ref].max():.2f}]")
    assert e_hm < tol
    assert len(mism) == 0 or margins.max() < MARGIN
    return e_hm


def test_forward_128_fp32_golden(det_fp32, golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_fwd128.npz"))
    x = np.concatenate([synth.noise_images(1234, 1, 128, 128), synth.page_images(77, 1, 128, 128)])
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).to("cuda")          # NHWC memory behind an NCHW view
    with torch.no_grad():
        hm, ft = det_fp32(xt)
    assert hm.shape == (2, 10, 32, 32) and ft.shape == (2, 100, 32, 32) and hm.dtype == torch.float32
    hm, ft = hm.cpu().numpy(), ft.cpu().numpy()
    _compare_maps("g1 128x128 fp32", hm, ft, g["heatmap"], g["features"])
    assert float(np.abs(ft - g["features"]).max()) < TOL


@pytest.mark.parametrize("name", ["test1", "page"])
def test_forward_768_fp32_golden_and_decode(det_fp32, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"g2_fwd768_{name}.npz"))
    if name == "test1":
        from PIL import Image
        im = np.asarray(Image.open(os.path.join(golden_dir, "test1_padded.png")).convert("RGB")).astype(np.float32)
        inp = im[None]
    else:
        inp = synth.page_images(4242, 1, 768, 768) * np.float32(255.)
    be = HipDetectorBackend(det_fp32)
    if name == "test1":
        hm, ft = be.call_detector(inp)                                # the reference plug-in signature
    else:
        with torch.no_grad():
            h_, f_ = det_fp32(torch.from_numpy(synth.page_images(4242, 1, 768, 768)).permute(0, 3, 1, 2).to("cuda"))
        hm, ft = h_.cpu().numpy(), f_.cpu().numpy()
    assert hm.shape == (1, 10, 192, 192) and ft.shape == (1, 100, 192, 192)
    _compare_maps(f"g2 768 {name} fp32", hm, None, g["heatmap"])
    e_ft = float(np.abs(ft[0].reshape(100, -1)[:, g["feat_pos"]] - g["feat_at"]).max())
    _log(f"g2 768 {name}: features@1024 Linf {e_ft:.3e}")
    assert e_ft < TOL
    # decode on the GPU maps vs the oracle decode on the REFERENCE maps: same peak set
    rect = tile_keep_rect(0, 0, 768, 768, 0.6)
    heat_nhwc = torch.from_numpy(hm).permute(0, 2, 3, 1).contiguous().cuda()
    feat_nhwc = torch.from_numpy(ft).permute(0, 2, 3, 1).contiguous().cuda()
    dec = decode_peaks(heat_nhwc, feat_nhwc, [TileGeom(0, 0, 768, 768, rect)], cut_off=0.4)
    n = int(dec.counts[0])
    idx_gpu = dec.index[0, :n].cpu().numpy()
    # oracle on the golden heat-map (features are not needed for the index set)
    loc, _, idx_ref = decode_oracle.decode_tile(g["heatmap"], np.zeros((1, 100, 192, 192), np.float32), 0, 0, 768, 768, 0.4, rect)
    only_gpu, only_ref = set(idx_gpu) - set(idx_ref), set(idx_ref) - set(idx_gpu)
    _log(f"g2 768 {name}: peaks gpu {n} ref {len(idx_ref)} only_gpu {len(only_gpu)} only_ref {len(only_ref)}")
    # a difference is tolerated only where the reference's own decision was an fp32 tie: score within
    # 2*TOL of the cut-off, or 3x3 keep/suppress margin below 1e-4 (test1.png is mostly white padding:
    # its flat background sits above the cut-off with random weights and neighbouring logits differ
    # by ~1e-6, so the reference's own peak list there depends on its thread count)
    hard = []
    for i in only_gpu | only_ref:
        y, x = divmod(int(i), 192)
        near_cut = abs(float(g["heatmap"][0, 0, y, x]) - np.log(0.4 / 0.6)) < 2 * TOL
        if not (near_cut or _nms_margin(g["heatmap"], 0, y, x) < 1e-4):
            hard.append((y, x))
    _log(f"g2 768 {name}: differing peaks that are NOT reference-side ties: {len(hard)}")
    assert not hard and n > 20
    if name == "page":
        assert not only_gpu and not only_ref                           # realistic page: index set bit-exact


def test_matches_oracle_on_fresh_input(det_fp32, sd):
    """Same seeded input through the CPU oracle (not a stored fixture), batch 3 at 256x192."""
    from oracle import detector_oracle
    x = np.concatenate([synth.page_images(9, 2, 256, 192), synth.noise_images(10, 1, 256, 192)])
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    o_hm, o_ft = detector_oracle.detector_forward(sd, xt)
    with torch.no_grad():
        hm, ft = det_fp32(xt.to("cuda"))
    _compare_maps("oracle 256x192 B3 fp32", hm.cpu().numpy(), ft.cpu().numpy(), o_hm.numpy(), o_ft.numpy())
    assert float((ft.cpu() - o_ft).abs().max()) < TOL


def test_input_layouts_and_9ch_forward(det_fp32):
    x = torch.from_numpy(synth.page_images(21, 1, 128, 160))
    a = x.permute(0, 3, 1, 2).to("cuda")                       # channels_last strides
    b = x.permute(0, 3, 1, 2).contiguous().to("cuda")          # plain NCHW
    with torch.no_grad():
        h1, f1 = det_fp32(a)
        h2, f2 = det_fp32(b)
        maps, f3 = det_fp32.detector(a)
    assert torch.equal(h1, h2) and torch.equal(f1, f2)
    assert maps.shape == (1, 9, 32, 40)
    assert torch.equal(maps[:, 0], h1[:, 0]) and torch.equal(maps[:, 1:], h1[:, 2:]) and torch.equal(f3, f1)


def test_deterministic_and_batch_invariant(det_fp32):
    x = torch.from_numpy(synth.page_images(5, 3, 128, 128)).permute(0, 3, 1, 2).to("cuda")
    with torch.no_grad():
        h1, f1 = det_fp32(x)
        h2, f2 = det_fp32(x)
        h3, f3 = det_fp32(x[1:2])
    assert torch.equal(h1, h2) and torch.equal(f1, f2)                      # run-to-run bit identical
    fin = torch.isfinite(h1[1:2])
    assert torch.equal(fin, torch.isfinite(h3))
    assert float((h1[1:2][fin] - h3[fin]).abs().max()) < 1e-4               # tiles are independent units
    assert float((f1[1:2] - f3).abs().max()) < 1e-4


def test_training_mode_and_cpu_input_fail_loudly(det_fp32):
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(RuntimeError):
        det_fp32(x)                                                         # CPU tensor: no fallback
    det_fp32.train()
    with pytest.raises(NotImplementedError):
        det_fp32(x.cuda())
    det_fp32.eval()


def test_forward_768_bf16_speed_mode(det_bf16, golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_fwd768_page.npz"))
    x = torch.from_numpy(synth.page_images(4242, 1, 768, 768)).permute(0, 3, 1, 2).to("cuda")
    with torch.no_grad():
        hm, ft = det_bf16(x)
    hm, ft = hm.cpu().numpy(), ft.cpu().numpy()
    gh = g["heatmap"]
    both = np.isfinite(hm) & np.isfinite(gh)
    rng = float(gh[np.isfinite(gh)].max() - gh[np.isfinite(gh)].min())
    e = float(np.abs(hm[both] - gh[both]).max())
    e_ft = float(np.abs(ft[0].reshape(100, -1)[:, g["feat_pos"]] - g["feat_at"]).max())
    frng = float(g["feat_at"].max() - g["feat_at"].min())
    rect = tile_keep_rect(0, 0, 768, 768, 0.6)
    _, _, idx_ref = decode_oracle.decode_tile(gh, np.zeros((1, 100, 192, 192), np.float32), 0, 0, 768, 768, 0.4, rect)
    _, _, idx_bf = decode_oracle.decode_tile(hm, np.zeros((1, 100, 192, 192), np.float32), 0, 0, 768, 768, 0.4, rect)
    inter = len(set(idx_ref) & set(idx_bf))
    jac = inter / max(1, len(set(idx_ref) | set(idx_bf)))
    _log(f"bf16 768 page: heatmap Linf {e:.3e} ({100 * e / rng:.2f}% of range {rng:.1f})  features Linf {e_ft:.3e} "
         f"({100 * e_ft / frng:.2f}% of range)  peaks ref {len(idx_ref)} bf16 {len(idx_bf)} common {inter} jaccard {jac:.3f}")
    assert e / rng < 0.05 and e_ft / frng < 0.05 and jac > 0.85


@pytest.mark.parametrize("shape", [(3, 256, 192), (1, 320, 544), (5, 128, 128), (2, 448, 768)], ids=lambda s: "x".join(map(str, s)))
def test_bf16_mode_tracks_fp32_mode_on_other_geometries(det_fp32, det_bf16, shape):
    """The bf16 plan differs structurally from the fp32 one (fused top convolutions, in-loader upsample, per-image project weights,
    grouped heads): on map sizes that are not multiples of the 16x16 pixel tiles and on odd batch sizes both must describe the same
    network -- bf16 within its rounding noise of the fp32 (parity-mode) result, -inf NMS slots in the same places up to near-ties."""
    B, H, W = shape
    x = torch.from_numpy(synth.page_images(777 + H, B, H, W)).permute(0, 3, 1, 2).to("cuda")
    with torch.no_grad():
        h32, f32_ = det_fp32(x)
        h16, f16 = det_bf16(x)
    h32, f32_, h16, f16 = h32.cpu().numpy(), f32_.cpu().numpy(), h16.cpu().numpy(), f16.cpu().numpy()
    fin = np.isfinite(h32) & np.isfinite(h16)
    rng = float(h32[np.isfinite(h32)].max() - h32[np.isfinite(h32)].min())
    e = float(np.abs(h32[fin] - h16[fin]).max())
    frng = float(f32_.max() - f32_.min())
    ef = float(np.abs(f32_ - f16).max())
    flips = int((np.isfinite(h32[:, 1]) != np.isfinite(h16[:, 1])).sum())
    _log(f"bf16 vs fp32 {shape}: heatmap Linf {100 * e / rng:.2f}% of range, features {100 * ef / frng:.2f}%, NMS flips {flips} of {h32[:, 1].size}")
    assert e / rng < 0.03 and ef / frng < 0.03
    assert flips < 0.02 * h32[:, 1].size
