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
from gpu_harness import fresh_model, shared_detector
from oracle import decode_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-3
# peak-set agreement of the bf16 speed mode with the fp32 reference golden; the reference itself under CPU bf16 autocast reaches
# 0.85 Jaccard / 0.93 recall against its own fp32 run (1590 common of 1702 vs 1765)
# Gates = just below what is MEASURED on the 768x768 page fixture (gpurun_out/test_detector.log over rounds 2-3: Jaccard 0.893 - 0.898,
# recall 0.950 - 0.959), so that a regression of the bf16 kernels fails instead of hiding in the slack.
BF16_JACCARD_GATE = 0.885
BF16_RECALL_GATE = 0.945
# fp16 (the recommended 16-bit mode, same speed): measured Jaccard 0.997 / recall 0.999 on the same fixture
FP16_JACCARD_GATE = 0.985
FP16_RECALL_GATE = 0.99


def _log(msg):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/test_detector.log", "a") as f:
        f.write(msg + "\n")
    print(msg)


@pytest.fixture(scope="module")
def sd():
    return deterministic_state_dict(0)


# Every test written for the fp32 parity mode also runs on "fp16x3" (fp32 tensors / weights / accumulation, each product as three
# fp16 MFMAs of hi / lo split operands): ONE program that has to meet the reference's 1e-3 tolerance, the exact NMS mask and the
# exact peak set AND run above 125 images/s per GPU (the fp32-MFMA mode: 117).
@pytest.fixture(scope="module", params=["fp32", "fp16x3"])
def det_fp32(request):
    return shared_detector(request.param)[0]


@pytest.fixture(scope="module")
def det_bf16():
    return shared_detector("bf16")[0]


@pytest.fixture(scope="module")
def det_fp16():
    return shared_detector("fp16")[0]


def _load_unstable(golden_dir, tag):
    """Reference-side decision-stability data written by tests/golden/gen_golden.py::gen_instability: which keep/suppress and
    above/below-cut decisions the REFERENCE itself takes differently between its fp32 runs at 1/2/4/8 threads and its float64
    run, plus the float64 key map and 3x3 margins."""
    u = np.load(os.path.join(golden_dir, f"g2_{tag}_unstable.npz"))
    shape = tuple(int(v) for v in u["shape"])
    n = int(np.prod(shape))
    return {"nms": np.unpackbits(u["nms_unstable"])[:n].reshape(shape).astype(bool),
            "cut": np.unpackbits(u["cut_unstable"])[:n].reshape(shape).astype(bool),
            "noise": float(u["noise"]), "key64": u["key64"], "margin64": u["margin64"]}


def _oracle_unstable(o_hm):
    """Same envelope for inputs that have no reference-written fixture: margins from the CPU oracle's map, noise = the bound to
    which tests/test_oracle.py pins the oracle against the reference's goldens (1e-4)."""
    k = o_hm[:, 0].astype(np.float64)
    h, w = k.shape[-2:]
    pad = np.pad(k, [(0, 0), (1, 1), (1, 1)], constant_values=-np.inf)
    nb = np.stack([pad[:, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3) if (dy, dx) != (1, 1)]).max(0)
    z = np.zeros(k.shape, bool)
    return {"nms": z, "cut": z, "noise": 1e-4, "key64": k.astype(np.float32), "margin64": np.abs(k - nb).astype(np.float32)}


def _compare_maps(tag, hm, ft, g_hm, g_ft=None, tol=TOL, unstable=None):
    """L-inf on finite entries < tol.  The -inf (suppressed) pattern of the NMS channel must be IDENTICAL to the reference's
    outside the reference-derived instability mask: pixels whose keep/suppress decision the reference itself takes differently
    among its own fp32 (1/2/4/8 threads) and float64 runs, or whose float64 margin to the best neighbour is within the
    reference's fp32 rounding envelope widened by twice the key-map error measured HERE (a result that is e away from the
    golden can only flip decisions whose exact margin is below noise + 2e).  Without a mask (oracle-generated cases) no flip
    is allowed at all."""
    fin_ref, fin = np.isfinite(g_hm), np.isfinite(hm)
    both = fin & fin_ref
    e_hm = float(np.abs(hm[both] - g_hm[both]).max())
    e_key = float(np.abs(hm[:, 0] - g_hm[:, 0]).max())
    mism = fin != fin_ref
    assert not mism[:, [0] + list(range(2, 10))].any()              # only the NMS channel can hold -inf
    flips = mism[:, 1]
    e_ft = float(np.abs(ft - g_ft).max()) if g_ft is not None else None
    if unstable is not None:
        allowed = unstable["nms"] | (unstable["margin64"] <= unstable["noise"] + 2 * e_key)
        n_allowed = int(allowed.sum())
    else:
        allowed, n_allowed = np.zeros_like(flips), 0
    outside = int((flips & ~allowed).sum())
    _log(f"{tag}: heatmap Linf {e_hm:.3e} (key {e_key:.2e})  features Linf {e_ft}  NMS-mask flips {int(flips.sum())} of "
         f"{flips.size} px, {outside} outside the reference-derived instability mask ({n_allowed} px)  "
         f"hm range [{g_hm[fin_ref].min():.2f},{g_hm[fin_ref].max():.2f}]")
    assert e_hm < tol
    assert outside == 0
    return e_hm, e_key


def test_forward_128_fp32_golden(det_fp32, golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_fwd128.npz"))
    x = np.concatenate([synth.noise_images(1234, 1, 128, 128), synth.page_images(77, 1, 128, 128)])
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).to("cuda")          # NHWC memory behind an NCHW view
    with torch.no_grad():
        hm, ft = det_fp32(xt)
    assert hm.shape == (2, 10, 32, 32) and ft.shape == (2, 100, 32, 32) and hm.dtype == torch.float32
    hm, ft = hm.cpu().numpy(), ft.cpu().numpy()
    _compare_maps("g1 128x128 fp32", hm, ft, g["heatmap"], g["features"], unstable=_load_unstable(golden_dir, "fwd128"))
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
    un = _load_unstable(golden_dir, f"fwd768_{name}")
    _, e_key = _compare_maps(f"g2 768 {name} fp32", hm, None, g["heatmap"], unstable=un)
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
    # A peak may differ from the reference's list only where the reference's OWN decision is unstable (see _load_unstable):
    # its NMS call (plateaus along the white padding of test1.png: the reference flips 1136 of those decisions between its own
    # thread counts) or its comparison with the cut-off, each envelope widened by twice the error measured in this run.
    allowed = (un["nms"] | (un["margin64"] <= un["noise"] + 2 * e_key) | un["cut"]
               | (np.abs(un["key64"] - np.float32(np.log(0.4 / 0.6))) <= un["noise"] + 2 * e_key))[0].reshape(-1)
    hard = sorted(int(i) for i in only_gpu | only_ref if not allowed[int(i)])
    n_ref_stable = sum(1 for i in idx_ref if not allowed[int(i)])
    _log(f"g2 768 {name}: peaks gpu {n} ref {len(idx_ref)} only_gpu {len(only_gpu)} only_ref {len(only_ref)}; outside the "
         f"reference-derived instability mask: {len(hard)} differ, {n_ref_stable} reference peaks are stable and all found")
    assert not hard and n > 20
    # bounded inside the mask as well: no more differing peaks than unstable pixels above the cut-off
    assert len(only_gpu | only_ref) <= int((allowed & (un["key64"][0].reshape(-1) > np.log(0.4 / 0.6) - 1e-3)).sum())
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
    _compare_maps("oracle 256x192 B3 fp32", hm.cpu().numpy(), ft.cpu().numpy(), o_hm.numpy(), o_ft.numpy(),
                  unstable=_oracle_unstable(o_hm.numpy()))
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
    recall = inter / max(1, len(set(idx_ref)))
    _log(f"bf16 768 page: heatmap Linf {e:.3e} ({100 * e / rng:.2f}% of range {rng:.1f})  features Linf {e_ft:.3e} "
         f"({100 * e_ft / frng:.2f}% of range)  peaks ref {len(idx_ref)} bf16 {len(idx_bf)} common {inter} jaccard {jac:.3f} recall {recall:.3f} "
         f"(the reference under its own CPU bf16 autocast: jaccard 0.847, recall 0.934 -- SURVEY.md Appendix C)")
    assert e / rng < 0.05 and e_ft / frng < 0.05 and jac >= BF16_JACCARD_GATE and recall >= BF16_RECALL_GATE


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


def _peak_sets(hm):
    rect = tile_keep_rect(0, 0, 768, 768, 0.6)
    z = np.zeros((1, 100, 192, 192), np.float32)
    return [set(decode_oracle.decode_tile(hm[b:b + 1], z, 0, 0, 768, 768, 0.4, rect)[2].tolist()) for b in range(hm.shape[0])]


@pytest.mark.parametrize("prec,B", [("bf16", 8), ("bf16", 32), ("fp16", 8), ("fp16", 32)])
def test_bench_plans_b8_b32_16bit_match_their_b1_results_and_the_golden(det_bf16, det_fp16, golden_dir, prec, B):
    """BASELINE configs[1] (batch 8, what bench.py times) and configs[3] (batch 32, forward + NMS + decode + gather): the plan is
    built per batch size and the measured kernel table is keyed by it, so the big-batch plans get their own check -- every image
    against the SAME image run alone (tiles are independent units), image 0 against the reference golden, and the GPU decode of
    the whole batch against the oracle decode of the same maps."""
    det_bf16 = det_bf16 if prec == "bf16" else det_fp16            # (the body below is written for "the 16-bit detector under test")
    # plan-vs-plan and plan-vs-golden gates, just below the measurements (bf16: B-vs-1 Jaccard 0.880 - 0.883 while both plans ran the same
    # kernel families; round 4: the batch-8 / batch-32 plans run stages 4-7 through FTC_OP_MBHEAD, the batch-1 plan keeps the three-kernel
    # form -- same rounding points, more sums that associate differently: 0.865; against the golden nothing moved)
    # (fp16: 0.977 / 0.987 measured in rounds 4-5; round 6, stage 2 through FTC_OP_FMBCONV in both plans: 0.969 / 0.98x -- the B = 8 plan's other tile choices move
    #  ~300 of 295 k NMS decisions on the noise images either way)
    jmin_gate, lin_gate, jac_gate, rec_gate = (0.85, 0.02, BF16_JACCARD_GATE, BF16_RECALL_GATE) if prec == "bf16" else (0.96, 0.004, FP16_JACCARD_GATE, FP16_RECALL_GATE)
    g = np.load(os.path.join(golden_dir, "g2_fwd768_page.npz"))
    imgs = np.concatenate([synth.page_images(4242, 1, 768, 768)] + [synth.noise_images(900 + i, 1, 768, 768) if i % 2 else
                                                                     synth.page_images(900 + i, 1, 768, 768) for i in range(1, B)])
    x = torch.from_numpy(imgs).permute(0, 3, 1, 2).to("cuda")
    with torch.no_grad():
        heat, feat = det_bf16.forward_nhwc(x)
        rect = tile_keep_rect(0, 0, 768, 768, 0.6)
        dec = decode_peaks(heat, feat, [TileGeom(0, 0, 768, 768, rect)] * B, cut_off=0.4, max_boxes=4096)
        hm = heat.permute(0, 3, 1, 2).cpu().numpy()
        ft = feat.permute(0, 3, 1, 2).cpu().numpy()
        worst, worst_f, flips, npx, jmin = 0.0, 0.0, 0, 0, 1.0
        sets_b = _peak_sets(hm)
        for b in ([0, 1, 2, B // 2, B - 1] if B > 8 else range(B)):
            h1, f1 = det_bf16.forward_nhwc(x[b:b + 1])
            h1, f1 = h1.permute(0, 3, 1, 2).cpu().numpy(), f1.permute(0, 3, 1, 2).cpu().numpy()
            fin = np.isfinite(h1) & np.isfinite(hm[b:b + 1])
            rng_b = float(h1[np.isfinite(h1)].max() - h1[np.isfinite(h1)].min())
            worst = max(worst, float(np.abs(h1[fin] - hm[b:b + 1][fin]).max()) / rng_b)
            worst_f = max(worst_f, float(np.abs(f1 - ft[b:b + 1]).max()) / float(f1.max() - f1.min()))
            flips += int((np.isfinite(h1[:, 1]) != np.isfinite(hm[b:b + 1, 1])).sum())
            npx += h1[:, 1].size
            s1 = _peak_sets(h1)[0]
            jmin = min(jmin, len(s1 & sets_b[b]) / max(1, len(s1 | sets_b[b])))
    # Same network and the same bf16 rounding POINTS in both plans, but the batch-N plan picks other tile shapes / split-K variants
    # (measured per shape): their fp32 sums associate differently, which moves a fraction of the bf16 roundings of the activations
    # by one ulp.  Two bf16 plans therefore agree to bf16 noise (as bf16 vs fp32 does), not to fp32 noise.
    _log(f"{prec} B={B} vs B=1 plans: heatmap Linf {100 * worst:.2f}% of range, features {100 * worst_f:.2f}%, NMS flips {flips} of {npx}, "
         f"min peak jaccard {jmin:.3f}")
    # (two independently rounded bf16 results differ from each other by ~sqrt(2) x what each differs from fp32)
    assert worst < lin_gate and worst_f < lin_gate and flips < lin_gate * npx and jmin >= jmin_gate
    gh = g["heatmap"]
    both = np.isfinite(hm[:1]) & np.isfinite(gh)
    rng = float(gh[np.isfinite(gh)].max() - gh[np.isfinite(gh)].min())
    e = float(np.abs(hm[:1][both] - gh[both]).max())
    ref, got = _peak_sets(gh)[0], sets_b[0]
    jac = len(ref & got) / max(1, len(ref | got))
    recall = len(ref & got) / max(1, len(ref))
    _log(f"{prec} B={B} image 0 vs reference golden: heatmap Linf {e:.3e} ({100 * e / rng:.2f}% of range), peak jaccard {jac:.3f} recall {recall:.3f}")
    assert e / rng < (0.03 if prec == "bf16" else 0.0015) and jac >= jac_gate and recall >= rec_gate
    # GPU decode + gather of the whole batch == oracle decode of the same maps (bit-exact index sets, copied feature rows)
    for b in range(B):
        n = int(dec.counts[b])
        assert n == len(sets_b[b]) and set(dec.index[b, :n].cpu().tolist()) == sets_b[b]
        k = min(n, 16)
        idx = dec.index[b, :k].cpu().numpy()
        assert np.array_equal(dec.feats[b, :k].cpu().numpy(), ft[b].reshape(100, -1)[:, idx].T)


def _plan_labels(m, B, H, W):
    """Kernel labels (ftc_op_kernel_label) of the plan the library runs for this shape -- the plan ftc_forward itself selects."""
    import ctypes as C
    from findtextcenternet_amd import _lib as L
    lib = L.load()
    pl = m.detector._engine.model.plan(B, H, W)
    buf = C.create_string_buffer(160)
    out = []
    for i in range(len(pl.ops)):
        L.check(lib.ftc_op_kernel_label(C.byref(pl.ops[i]), buf, 160), "ftc_op_kernel_label")
        out.append(buf.value.decode())
    return out


@pytest.mark.parametrize("prec,B", [("fp16x3", 8), ("fp16x3", 32), ("fp32", 8)])
def test_bench_plans_b8_b32_contract_grade_modes_meet_the_reference_tolerance(golden_dir, prec, B):
    """The plans that produce bench.py's contract-grade figure (`config.contract_mode` / `north_star_value`: fp16x3 at batch 8 and 32; fp32 beside
    it) checked at FULL size against the reference's golden -- round 5's full-model golden tests only reached these modes at B <= 3, where
    model.hip selects no fused MBConv head (fewer than 128 workgroups) and the 144-pixel project tiles are not the batch-8 choices.
    BASELINE.json north_star: maps within 1e-3, peak indices bit-exact.  Image 0 is golden g2 `page`: heat-map within TOL with the NMS mask
    identical outside the reference-derived instability mask, features at the 1024 stored positions within TOL, peak index set identical
    to the oracle decode of the REFERENCE's map; every image (a sample at B = 32) against the same image run alone within 1e-4 and with the
    same NMS mask (tiles are independent units); GPU decode + gather of the batch == oracle decode of the same maps."""
    det, m = shared_detector(prec)
    g = np.load(os.path.join(golden_dir, "g2_fwd768_page.npz"))
    imgs = np.concatenate([synth.page_images(4242, 1, 768, 768)] + [synth.noise_images(900 + i, 1, 768, 768) if i % 2 else
                                                                     synth.page_images(900 + i, 1, 768, 768) for i in range(1, B)])
    x = torch.from_numpy(imgs).permute(0, 3, 1, 2).to("cuda")
    with torch.no_grad():
        heat, feat = det.forward_nhwc(x)
        rect = tile_keep_rect(0, 0, 768, 768, 0.6)
        dec = decode_peaks(heat, feat, [TileGeom(0, 0, 768, 768, rect)] * B, cut_off=0.4, max_boxes=4096)
        hm = heat.permute(0, 3, 1, 2).cpu().numpy()
        ft = feat.permute(0, 3, 1, 2).cpu().numpy()
    labels = _plan_labels(m, B, 768, 768)
    if prec == "fp16x3":                                     # the kernels round 5's fp16x3 gain came from must be IN the plan under test
        n_head = sum(l.startswith("mbconv_slice<f16x3") for l in labels)
        n_px = sum(l.startswith("conv1x1_px144<f16x3") for l in labels)
        _log(f"{prec} B={B} plan: {len(labels)} ops, {n_head} mbconv_slice<f16x3>, {n_px} conv1x1_px144<f16x3>")
        assert n_head >= 70 and n_px >= 50, (n_head, n_px)       # (B = 8: 78 fused heads, 54 of the 78 project convolutions on 144-pixel tiles)
    # (a) image 0 against the reference's golden
    un = _load_unstable(golden_dir, "fwd768_page")
    _compare_maps(f"{prec} B={B} image 0 vs golden g2 page", hm[:1], None, g["heatmap"], unstable=un)
    e_ft = float(np.abs(ft[0].reshape(100, -1)[:, g["feat_pos"]] - g["feat_at"]).max())
    sets_b = _peak_sets(hm)
    ref = _peak_sets(g["heatmap"])[0]
    _log(f"{prec} B={B} image 0: features@1024 Linf {e_ft:.3e}; peaks {len(sets_b[0])} vs reference {len(ref)}, identical: {sets_b[0] == ref}")
    assert e_ft < TOL and sets_b[0] == ref and len(ref) > 20
    # (b) every image against the same image run alone.  The two plans sum in different orders (other tiles, fused heads): a keep /
    # suppress or cut-off decision may differ only where its margin is below twice the difference measured between the two runs
    worst, worst_f, n_flip = 0.0, 0.0, 0
    with torch.no_grad():
        for b in ([0, 1, 2, B // 2, B - 1] if B > 8 else range(B)):
            h1, f1 = det.forward_nhwc(x[b:b + 1])
            h1, f1 = h1.permute(0, 3, 1, 2).cpu().numpy(), f1.permute(0, 3, 1, 2).cpu().numpy()
            fin = np.isfinite(h1) & np.isfinite(hm[b:b + 1])
            e_b = float(np.abs(h1[fin] - hm[b:b + 1][fin]).max())
            env = _oracle_unstable(h1)
            near = (env["margin64"] <= 2 * e_b)[0] | (np.abs(env["key64"][0] - np.float32(np.log(0.4 / 0.6))) <= 2 * e_b)
            flips = np.isfinite(h1[0, 1]) != np.isfinite(hm[b, 1])
            assert not (flips & ~near).any(), f"image {b}: NMS mask differs between the batch-{B} and batch-1 plans outside near-ties"
            diff = _peak_sets(h1)[0] ^ sets_b[b]
            assert all(near.reshape(-1)[i] for i in diff), f"image {b}: peak sets differ outside near-ties: {sorted(diff)[:8]}"
            n_flip += int(flips.sum()) + len(diff)
            worst, worst_f = max(worst, e_b), max(worst_f, float(np.abs(f1 - ft[b:b + 1]).max()))
    _log(f"{prec} B={B} vs B=1 plans: heatmap Linf {worst:.2e}, features Linf {worst_f:.2e}, {n_flip} near-tie decisions differ")
    assert worst < 1e-4 and worst_f < 1e-4 and n_flip <= 4
    # (c) GPU decode + gather of the whole batch == oracle decode of the same maps
    for b in range(B):
        n = int(dec.counts[b])
        assert n == len(sets_b[b]) and set(dec.index[b, :n].cpu().tolist()) == sets_b[b]
        k = min(n, 16)
        idx = dec.index[b, :k].cpu().numpy()
        assert np.array_equal(dec.feats[b, :k].cpu().numpy(), ft[b].reshape(100, -1)[:, idx].T)


@pytest.mark.parametrize("prec", ["bf16", "fp16x3"])
def test_forward_stays_inside_its_buffers(prec):
    """Device-side bounds check of the whole forward (the ASan build of the shim only sees host code): workspace, both
    outputs and the input sit between 1 MiB guard bands of a known pattern inside ONE allocation; after forwards at several shapes -- the
    benchmarked batch 8, an odd batch whose plan takes other kernels (no FTC_OP_MBHEAD below 128 workgroups), a non-square tile -- every
    guard byte is unchanged, and a forward into a NaN-poisoned workspace gives the same result as into a zeroed one (nothing reads what
    the plan did not write first)."""
    det, m = shared_detector(prec)
    eng = m.detector._engine
    G = 1 << 20
    for (B, H, W) in [(8, 768, 768), (3, 768, 768), (2, 512, 640)]:
        x_h = torch.from_numpy(synth.page_images(77 + B, B, H, W))
        with torch.no_grad():
            det.forward_nhwc(x_h[:1].permute(0, 3, 1, 2).to("cuda"))                      # (builds the model on the device)
        ws_n = eng.model.workspace_bytes(B, H, W)
        h, w = H // 4, W // 4
        sizes = [B * H * W * 3 * 4, ws_n, B * h * w * 10 * 4, B * h * w * 100 * 4]
        offs, cur = [], G
        for n in sizes:
            offs.append(cur)
            cur += (n + 255) // 256 * 256 + G
        big = torch.full((cur,), 0xA5, dtype=torch.uint8, device="cuda")
        x = big[offs[0]:offs[0] + sizes[0]].view(torch.float32).reshape(B, H, W, 3)
        x.copy_(x_h.to("cuda"))
        ws = big[offs[1]:offs[1] + sizes[1]]
        heat = big[offs[2]:offs[2] + sizes[2]].view(torch.float32).reshape(B, h, w, 10)
        feat = big[offs[3]:offs[3] + sizes[3]].view(torch.float32).reshape(B, h, w, 100)
        results = []
        for fill in (0x00, 0xFF):                                                           # 0xFF = NaN patterns in every fp32 / 16-bit slot
            ws.fill_(fill)
            heat.fill_(float("nan")); feat.fill_(float("nan"))
            with torch.no_grad():
                det.forward_nhwc(x.permute(0, 3, 1, 2), out=(heat, feat), workspace=ws)
            torch.cuda.synchronize()
            results.append((heat.clone(), feat.clone()))
        assert torch.equal(results[0][1], results[1][1]) and bool(torch.isfinite(results[1][1]).all())
        assert torch.equal(torch.nan_to_num(results[0][0], nan=7.0, neginf=-7.0), torch.nan_to_num(results[1][0], nan=7.0, neginf=-7.0))
        bands, lo = [], 0
        for o, n in zip(offs, sizes):
            bands.append((lo, o))
            lo = o + n
        bands.append((lo, cur))
        for a_, b_ in bands:
            assert bool((big[a_:b_] == 0xA5).all()), f"a kernel wrote outside its buffers at B={B} {H}x{W} ({prec}): band [{a_}, {b_})"
        _log(f"{prec} B={B} {H}x{W}: {sum(b_ - a_ for a_, b_ in bands)} guard bytes around input / workspace / outputs intact; NaN-poisoned workspace gives the same maps")
        del big, x, ws, heat, feat, results
        torch.cuda.empty_cache()


def test_parameter_edits_are_noticed(sd):
    """The packed weight blob must follow the module: in-place edits on the parameter (p.add_(), optimizer.step() incl. this repo's
    raw-pointer AdamWScheduleFree), re-allocations, load_state_dict (also assign=True) all change the next forward; a deep copy gets
    its own engine.  Writes through `p.data` are not detectable (separate version counter): they need engine.invalidate()."""
    import copy
    m = fresh_model("bf16")
    d = CenterNetDetector(m.detector).to("cuda").eval()
    x = torch.from_numpy(synth.page_images(3, 1, 128, 128)).permute(0, 3, 1, 2).to("cuda")
    with torch.no_grad():
        h0, _ = d(x)
        p = m.detector.keyheatmap.top_conv._modules["0"].bias
        p.add_(1.0)                                        # what an optimizer step does
        h1, _ = d(x)
        assert float((h1[:, 0] - h0[:, 0] - 1.0).abs().max()) < 1e-5
        p.data = p.data - 1.0                              # re-allocation: same version, new address
        h2, _ = d(x)
        assert torch.equal(h2[:, 0], h0[:, 0])
        d2 = copy.deepcopy(d)
        assert d2.detector._engine is not d.detector._engine and d2.detector._engine.model is None
        h3, _ = d2(x)
        assert torch.equal(h3, h2)
        # in-place writes through `.data` carry their own version counter: NOT seen, documented -> invalidate() by hand
        p.data.add_(1.0)
        m.detector._engine.invalidate()
        h4, _ = d(x)
        assert float((h4[:, 0] - h0[:, 0] - 1.0).abs().max()) < 1e-5
        # load_state_dict(assign=True) replaces the Parameter objects: the post-hook re-packs
        m.load_state_dict(sd, assign=True)
        h5, _ = d(x)
        assert torch.equal(h5[:, 0], h0[:, 0])
        # this repo's optimizer writes parameters through raw device pointers and bumps their version counters
        from findtextcenternet_amd import AdamWScheduleFree
        m.to("cuda")
        q = m.detector.keyheatmap.top_conv._modules["0"].bias
        opt = AdamWScheduleFree([q], lr=0.5)
        opt.train()
        h6, _ = d(x)
        q.grad = torch.ones_like(q)
        opt.step()
        h7, _ = d(x)
        assert float((h7[:, 0] - h6[:, 0]).abs().max()) > 1e-3


def test_fp16_mode_same_plan_much_closer_to_the_reference(golden_dir):
    """precision='fp16': the speed-mode plan with IEEE-half MFMA operands (same matrix rate as bf16, 11-bit significands).
    Against the reference golden it must beat the bf16 mode by a wide margin: heat-map / features within 0.15 % of range (bf16:
    ~0.5 %), peak-set Jaccard >= 0.97."""
    d = shared_detector("fp16")[0]
    g = np.load(os.path.join(golden_dir, "g2_fwd768_page.npz"))
    x = torch.from_numpy(synth.page_images(4242, 1, 768, 768)).permute(0, 3, 1, 2).to("cuda")
    with torch.no_grad():
        hm, ft = d(x)
    hm, ft = hm.cpu().numpy(), ft.cpu().numpy()
    gh = g["heatmap"]
    both = np.isfinite(hm) & np.isfinite(gh)
    rng = float(gh[np.isfinite(gh)].max() - gh[np.isfinite(gh)].min())
    e = float(np.abs(hm[both] - gh[both]).max())
    e_ft = float(np.abs(ft[0].reshape(100, -1)[:, g["feat_pos"]] - g["feat_at"]).max())
    frng = float(g["feat_at"].max() - g["feat_at"].min())
    ref, got = _peak_sets(gh)[0], _peak_sets(hm)[0]
    jac, recall = len(ref & got) / max(1, len(ref | got)), len(ref & got) / max(1, len(ref))
    _log(f"fp16 768 page: heatmap Linf {e:.3e} ({100 * e / rng:.3f}% of range)  features Linf {e_ft:.3e} ({100 * e_ft / frng:.3f}%)  "
         f"peaks ref {len(ref)} fp16 {len(got)} jaccard {jac:.3f} recall {recall:.3f}")
    assert e / rng < 0.0015 and e_ft / frng < 0.0015 and jac >= FP16_JACCARD_GATE and recall >= FP16_RECALL_GATE
    # other geometries / batch sizes run the same code paths as bf16 (shared tuning table): quick agreement check with fp32
    x2 = torch.from_numpy(synth.page_images(778, 3, 256, 192)).permute(0, 3, 1, 2).to("cuda")
    d32 = shared_detector("fp32")[0]
    with torch.no_grad():
        h16, f16 = d(x2)
        h32, f32_ = d32(x2)
    fin = torch.isfinite(h16) & torch.isfinite(h32)
    assert float((h16[fin] - h32[fin]).abs().max()) < 0.03 and float((f16 - f32_).abs().max()) < 0.05


@pytest.mark.parametrize("size", ["s", "m", "l"])
def test_other_model_sizes_match_the_reference(golden_dir, size):
    """TextDetectorModel(model_size='s' | 'm' | 'l') on the GPU against the reference's own outputs for those sizes (g8)."""
    g = np.load(os.path.join(golden_dir, f"g8_fwd128_{size}.npz"))
    x = torch.from_numpy(synth.page_images(int(g["seed"]), 1, 128, 128)).permute(0, 3, 1, 2).to("cuda")
    outs = {}
    for prec in ("fp32", "fp16", "bf16"):
        d = shared_detector(prec, size)[0]
        with torch.no_grad():
            hm, ft = d(x)
        outs[prec] = (hm.cpu().numpy(), ft.cpu().numpy())
    hm, ft = outs["fp32"]
    fin = np.isfinite(g["heatmap"])
    assert np.array_equal(np.isfinite(hm), fin)
    e, ef = float(np.abs(hm[fin] - g["heatmap"][fin]).max()), float(np.abs(ft - g["features"]).max())
    _log(f"model_size={size} 128x128 fp32: heatmap Linf {e:.3e} features Linf {ef:.3e}")
    assert e < TOL and ef < TOL
    rng = float(g["heatmap"][fin].max() - g["heatmap"][fin].min())
    for prec, lim in (("fp16", 0.004), ("bf16", 0.03)):
        h2, f2 = outs[prec]
        both = np.isfinite(h2) & fin
        e2 = float(np.abs(h2[both] - g["heatmap"][both]).max()) / rng
        _log(f"model_size={size} 128x128 {prec}: heatmap Linf {100 * e2:.3f}% of range")
        assert e2 < lim


def test_small_module_calls_replay_a_graph_bit_identical_and_follow_weight_edits(sd, monkeypatch):
    """One-tile module calls (the reference's loops; HipDetectorBackend.call_detector) replay the forward from a HIP graph captured on the
    second call: same bits as the eager path on every call, fresh output tensors per call, and an edited parameter re-packs and
    re-captures instead of replaying the old weights."""
    m = fresh_model("bf16")
    det = CenterNetDetector(m.detector)
    be = HipDetectorBackend(det)
    eng = m.detector._engine
    tiles = [(synth.page_images(700 + i, 1, 768, 768) * np.float32(255.)).astype(np.float32) for i in range(4)]

    def eager(t):
        monkeypatch.setenv("FTC_NO_GRAPH", "1")
        try:
            return be.call_detector(t)
        finally:
            monkeypatch.delenv("FTC_NO_GRAPH")
    for i, t in enumerate(tiles):
        hm, ft = be.call_detector(t)
        hr, fr = eager(t)
        assert hm.shape == (1, 10, 192, 192) and ft.shape == (1, 100, 192, 192)
        assert np.array_equal(hm, hr, equal_nan=True) and np.array_equal(ft, fr, equal_nan=True), i
    ents = list(eng._graphs.values())
    assert len(ents) == 1 and ents[0]["graph"] is not None              # calls 3 and 4 were replays
    x = torch.from_numpy(tiles[0] / np.float32(255.)).permute(0, 3, 1, 2).cuda()
    with torch.no_grad():
        a1, _ = det(x)
        a2, _ = det(x)
    assert a1.data_ptr() != a2.data_ptr() and torch.equal(a1, a2)        # a module call returns its own tensors
    before = be.call_detector(tiles[0])[1].copy()
    with torch.no_grad():
        dict(m.named_parameters())["detector.feature.top_conv.0.bias"].add_(0.5)
    after = be.call_detector(tiles[0])[1]
    assert np.abs(after - before - 0.5).max() < 1e-5                    # the feature head's output bias moved by exactly the edit
    assert np.array_equal(after, eager(tiles[0])[1], equal_nan=True)
    be.call_detector(tiles[1])
    hm, ft = be.call_detector(tiles[2])                                  # a replay of the re-captured graph
    hr, fr = eager(tiles[2])
    assert ents[0]["graph"] is not None and np.array_equal(hm, hr, equal_nan=True) and np.array_equal(ft, fr, equal_nan=True)
    # dropping the packed model drops the graphs with it
    eng.invalidate()
    assert all(e["graph"] is None for e in eng._graphs.values())
    assert np.array_equal(be.call_detector(tiles[2])[0], hr, equal_nan=True)
