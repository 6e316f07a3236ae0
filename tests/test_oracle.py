"""CPU: the oracle against the golden vectors produced by the reference's own code
(tests/golden/gen_golden.py).  Tolerances are stated per test."""
import gzip
import json
import os

import numpy as np
import pytest
import torch

import synth
from findtextcenternet_amd import schema
from findtextcenternet_amd.weights import deterministic_state_dict
from oracle import decode_oracle, detector_oracle

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def sd():
    return deterministic_state_dict(0, prefix_detector=False)


def test_state_dict_schema_matches_reference():
    with gzip.open(os.path.join(G, "state_dict_schema_xl.json.gz"), "rt") as f:
        ref = json.load(f)
    mine = schema.text_detector_schema("xl")
    assert ref["n_keys"] == 2444 == len(mine)
    assert [k for k, _, _ in ref["keys"]] == list(mine.keys())
    for k, shape, _ in ref["keys"]:
        assert tuple(shape) == tuple(mine[k][0]), k
    # published EfficientNetV2-XL size minus the classifier
    assert ref["backbone_params"] == 206_838_808
    n = sum(int(np.prod(s)) for k, (s, kind) in mine.items() if kind not in ("bn_mean", "bn_var", "bn_count"))
    assert n == ref["total_params"] == 262_350_422


def test_forward_128_matches_reference(sd):
    g = np.load(os.path.join(G, "g1_fwd128.npz"))
    x = np.concatenate([synth.noise_images(1234, 1, 128, 128), synth.page_images(77, 1, 128, 128)])
    hm, ft = detector_oracle.detector_forward(sd, torch.from_numpy(x).permute(0, 3, 1, 2))
    hm, ft = hm.numpy(), ft.numpy()
    fin = np.isfinite(g["heatmap"])
    assert np.array_equal(np.isfinite(hm), fin)                 # -inf (suppressed) positions exact
    assert np.abs(hm[fin] - g["heatmap"][fin]).max() < 2e-5     # fp32 summation-order noise only
    assert np.abs(ft - g["features"]).max() < 5e-5
    assert np.abs(g["heatmap"][fin]).max() > 3 and g["features"].std() > 1   # fixture is not degenerate


@pytest.mark.parametrize("name", ["test1", "page"])
def test_forward_768_matches_reference(sd, name):
    g = np.load(os.path.join(G, f"g2_fwd768_{name}.npz"))
    if name == "test1":
        from PIL import Image
        im = np.asarray(Image.open(os.path.join(G, "test1_padded.png")).convert("RGB")).astype(np.float32)
        x = torch.from_numpy(im[None] / 255.).permute(0, 3, 1, 2).float()
    else:
        x = torch.from_numpy(synth.page_images(4242, 1, 768, 768)).permute(0, 3, 1, 2)
    hm, ft = detector_oracle.detector_forward(sd, x)
    hm, ft = hm.numpy(), ft.numpy()
    fin = np.isfinite(g["heatmap"])
    assert np.array_equal(np.isfinite(hm), fin)
    assert np.abs(hm[fin] - g["heatmap"][fin]).max() < 1e-4
    assert np.abs(ft[0].reshape(100, -1)[:, g["feat_pos"]] - g["feat_at"]).max() < 2e-4


def test_nms_ties_match_reference():
    g = np.load(os.path.join(G, "g4_nms_ties.npz"))
    out = detector_oracle.nms_forward(torch.from_numpy(g["maps"])).numpy()
    assert np.array_equal(out, g["heatmap"])                   # comparisons only: bit exact
    k = g["heatmap"][0, 1]
    assert k[3, 3] == 5.0 and k[3, 4] == 5.0                   # ties are BOTH kept (`<`, detector.py:295)
    assert np.isfinite(k[8:10, 8:10]).all()


def test_sigmoid_known_answers():
    g = np.load(os.path.join(G, "g3_sigmoid.npz"))
    y = decode_oracle.sigmoid(g["x"])
    assert y.dtype == np.float32 and np.array_equal(y, g["y"])


class _Replay:
    def __init__(self, outs):
        self.outs, self.i = list(outs), 0

    def __call__(self, image_input):
        o = self.outs[self.i]
        self.i += 1
        return o


def test_decode_single_tile_matches_reference():
    g = np.load(os.path.join(G, "g3_decode_single.npz"))
    img = synth.page_uint8(31, 768, 768).astype(np.float32)
    ds = [{"input": img[None], "offsetx": 0, "offsety": 0}]
    loc, gf, lines, seps, raw = decode_oracle.run_detector(ds, img, _Replay([synth.detector_maps(101)]))
    assert loc.shape == g["locations"].shape and loc.shape[0] > 50
    assert np.array_equal(loc, g["locations"]) and np.array_equal(gf, g["glyphfeatures"])
    assert np.array_equal(lines, g["lines"]) and np.array_equal(seps, g["seps"])


def test_decode_2x2_tiles_matches_reference():
    g = np.load(os.path.join(G, "g3_decode_2x2.npz"))
    step = int(768 * 0.6)
    img = synth.page_uint8(32, 768 + step, 768 + step).astype(np.float32)
    ds, outs = [], []
    for n, (y, x) in enumerate([(0, 0), (0, step), (step, 0), (step, step)]):
        ds.append({"input": img[None, y:y + 768, x:x + 768], "offsetx": x, "offsety": y})
        outs.append(synth.detector_maps(200 + n))
    loc, gf, lines, seps, raw = decode_oracle.run_detector(ds, img, _Replay(outs))
    assert np.array_equal(loc, g["locations"]) and np.array_equal(gf, g["glyphfeatures"])
    assert np.array_equal(lines, g["lines"]) and np.array_equal(seps, g["seps"])


def test_decode_sparse_equals_per_tile_decode():
    """Fixture where the reference's page-level suppression removes nothing: its output is exactly
    the per-tile decode, which pins decode_tile (incl. cut-off boundary and the w/h skips)."""
    g = np.load(os.path.join(G, "g3_decode_sparse.npz"))
    hm = g["heatmap"]
    rng = np.random.Generator(np.random.PCG64(303))
    feat = rng.standard_normal((1, 100, 192, 192)).astype(np.float32)
    rect = decode_oracle.tile_keep_rect(0, 0, 768, 768, 0.6)
    loc, gf, idx = decode_oracle.decode_tile(hm, feat, 0, 0, 768, 768, 0.4, rect)
    ref = g["locations"]
    assert loc.shape[0] == ref.shape[0] > 100
    # page_merge only raises the code columns (3x3 max, process_ocr_base.py:628-648): compare p, ix, iy, w, h
    assert np.array_equal(loc[:, :5].astype(np.float32), ref[:, :5])
    assert np.array_equal(gf, g["glyphfeatures"])
    ys, xs = idx // 192, idx % 192
    assert (6, 6) in set(zip(ys.tolist(), xs.tolist())) and (6, 18) not in set(zip(ys.tolist(), xs.tolist()))
    assert (30, 30) not in set(zip(ys.tolist(), xs.tolist())) and (42, 42) not in set(zip(ys.tolist(), xs.tolist()))


def test_validation_step_oracle_matches_reference():
    """oracle/loss_oracle.py against what the reference's own models/detector.py + loss_func.py produced on CPU (g7)."""
    from oracle import loss_oracle
    g = np.load(os.path.join(G, "g7_validation_step.npz"))
    B, H, W = 2, 256, 256
    label, idmap = synth.train_labels(616, B, H // 4, W // 4)
    lab_t, id_t = torch.from_numpy(label), torch.from_numpy(idmap).to(torch.long)
    fmask = loss_oracle.get_fmask(lab_t)
    assert int(fmask.sum()) == int(g["n_mask"]) == 1024 * B
    assert np.array_equal(np.packbits(fmask.numpy()), g["fmask"])
    # forward: detector oracle (9 reference channels) + decoder on the gathered rows
    sd_full = deterministic_state_dict(0)
    sd_det = {k[len("detector."):]: v for k, v in sd_full.items() if k.startswith("detector.")}
    x = torch.from_numpy(synth.page_images(515, B, H, W)).permute(0, 3, 1, 2)
    hm10, ft = detector_oracle.detector_forward(sd_det, x)
    hm9 = hm10[:, [0, 2, 3, 4, 5, 6, 7, 8, 9]]
    assert float((hm9 - torch.from_numpy(g["heatmap"])).abs().max()) < 1e-4
    rows = ft.permute(0, 2, 3, 1).flatten(0, -2)[fmask]
    dec = loss_oracle.simple_decoder(sd_full, rows)
    for j in range(3):
        assert float((dec[j][torch.from_numpy(g["dec_rows"])] - torch.from_numpy(g[f"dec{j}_at"])).abs().max()) < 2e-3
        assert float((torch.logsumexp(dec[j], 1) - torch.from_numpy(g[f"dec{j}_lse"])).abs().max()) < 2e-3
    # losses on the network's outputs and on the synthetic case
    tgt = id_t[:, 0].flatten()[fmask].numpy()
    hm2, dec2 = synth.loss_case(717, B, H // 4, W // 4, tgt)
    for tag, hm_in, dec_in, tol in (("loss_", torch.from_numpy(g["heatmap"]), dec, 2e-4), ("loss2_", torch.from_numpy(hm2), [torch.from_numpy(d) for d in dec2], 2e-6)):
        out = loss_oracle.loss_function(fmask, lab_t, id_t, hm_in, dec_in)
        for k, v in out.items():
            want = float(g[tag + k])
            assert abs(float(v) - want) <= tol * max(1.0, abs(want)), (tag, k, float(v), want)
    assert float(g["loss2_correct"]) > 0 and float(g["loss2_total"]) > float(g["loss2_correct"])
    keys, seq = synth.cov_loss_sequence(818)
    cov = loss_oracle.CoVWeighting(len(keys))
    for step, vals in enumerate(seq):
        got = cov(vals)
        assert abs(got - float(g["cov_loss"][step])) < 2e-6 * max(1.0, abs(got)), (step, got, g["cov_loss"][step])
        assert np.abs(cov.alphas - g["cov_alphas"][step]).max() < 2e-6


@pytest.mark.parametrize("size", ["s", "m", "l"])
def test_small_model_sizes_match_reference(size):
    """model_size 's' / 'm' / 'l' (models/detector.py:131-136, 149-158): the schema loads strictly into the reference's own modules (the
    fixture was generated that way), and the oracle reproduces the reference's outputs (tests/golden/g8_fwd128_<size>.npz)."""
    g = np.load(os.path.join(G, f"g8_fwd128_{size}.npz"))
    assert int(g["n_keys"]) == len(schema.text_detector_schema(size))
    sd_ = deterministic_state_dict(0, model_size=size, prefix_detector=False, with_decoder=False)
    n_backbone = sum(int(np.prod(v.shape)) for k, v in sd_.items() if k.startswith("backbone.") and "running" not in k and "num_batches" not in k)
    assert n_backbone == int(g["backbone_params"])
    x = torch.from_numpy(synth.page_images(int(g["seed"]), 1, 128, 128)).permute(0, 3, 1, 2)
    hm, ft = detector_oracle.detector_forward(sd_, x)
    fin = np.isfinite(g["heatmap"])
    assert np.array_equal(np.isfinite(hm.numpy()), fin)
    assert np.abs(hm.numpy()[fin] - g["heatmap"][fin]).max() < 1e-4 and np.abs(ft.numpy() - g["features"]).max() < 1e-4


def test_train_mode_oracle_matches_reference():
    """oracle/detector_oracle.py's training-mode forward (batch-statistics BatchNorm, StochasticDepth with the saved draw) and
    SimpleDecoder in train() against the reference's own modules run in train() under no_grad (g9) -- the BN-refresh pass."""
    from oracle import loss_oracle
    g = np.load(os.path.join(G, "g9_train_forward.npz"))
    B, H, W = 3, 128, 128
    sd_full = deterministic_state_dict(0)
    x = torch.from_numpy(synth.page_images(929, B, H, W)).permute(0, 3, 1, 2)
    label, _ = synth.train_labels(930, B, H // 4, W // 4)
    fmask = loss_oracle.get_fmask(torch.from_numpy(label))
    assert np.array_equal(np.packbits(fmask.numpy()), g["fmask"])
    keep = {str(n): torch.from_numpy(k) for n, k in zip(g["keep_names"], g["keep"])}
    # every block the reference gave a StochasticDepth module that multiplies (p > 0, residual) is one of ours
    sd_det = {k[len("detector."):]: v for k, v in sd_full.items() if k.startswith("detector.")}
    res_blocks = set("detector." + p for p in detector_oracle.residual_blocks(sd_det))
    assert res_blocks <= set(keep)
    probs = detector_oracle.stochastic_depth_probs(sd_det)
    for n, k in keep.items():
        p = probs[n[len("detector."):]]
        assert all(abs(float(v)) < 1e-9 or abs(float(v) - 1.0 / (1.0 - p)) < 1e-5 for v in k)
    maps, feat, new = detector_oracle.detection_forward_train(sd_full, x, keep)
    assert float((maps - torch.from_numpy(g["maps"])).abs().max()) < 2e-4
    rows = feat.permute(0, 2, 3, 1).flatten(0, -2)[fmask]
    dec, new_dec = detector_oracle.decoder_forward_train(sd_full, rows)
    new = {"detector." + k: v for k, v in new.items()}
    new.update(new_dec)
    assert int(g["n_changed"]) == len(new)
    for i, k in enumerate(g["stat_names"]):
        want = torch.from_numpy(g[f"stat{i}"])
        assert float((new[str(k)] - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), k
    for j in range(3):
        assert float((dec[j][torch.from_numpy(g["dec_rows"])] - torch.from_numpy(g[f"dec{j}_at"])).abs().max()) < 5e-3
        assert float((torch.logsumexp(dec[j], 1) - torch.from_numpy(g[f"dec{j}_lse"])).abs().max()) < 5e-3


def test_train_step_oracle_matches_reference_backward():
    """oracle/train_oracle.py vs g10 (the reference's own train step + loss.backward() in fp32): loss, maps, every gradient norm, and
    the stored gradients entry by entry.  Gradients the reference holds only as rounding noise (|g| < 1e-7: a BatchNorm bias in front of
    another batch-statistics BatchNorm) are compared absolutely."""
    from oracle import train_oracle
    g = np.load(os.path.join(G, "g10_train_step.npz"))
    sd = deterministic_state_dict(0)
    B, H, W = 2, 256, 256
    x = torch.from_numpy(synth.page_images(1029, B, H, W)).permute(0, 3, 1, 2)
    label, idmap = synth.train_labels(1030, B, H // 4, W // 4)
    keep = {str(n): torch.from_numpy(k) for n, k in zip(g["keep_names"], g["keep"])}
    loss, raw, grads, maps = train_oracle.train_step(sd, x, torch.from_numpy(label), torch.from_numpy(idmap).long(), keep)
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    assert np.allclose(g["alphas"], 1.0 / 9)
    assert np.abs(maps.numpy() - g["heatmap"]).max() < 1e-3
    for n, nr in zip(g["grad_names"], g["grad_norms"]):
        gr = grads[str(n)]
        assert abs(float(gr.double().norm()) - nr) <= 1e-3 * nr + 1e-7 * np.sqrt(gr.numel()), n
    for i, n in enumerate(g["pick_names"]):
        st, ref = int(g[f"pick{i}_stride"]), g[f"pick{i}"]
        mine = grads[str(n)].numpy().reshape(-1)[::st]
        assert np.abs(mine - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-7, n
    assert sum(1 for a in g["grad_absmax"] if a == 0.0) >= 13      # a block dropped for every image: exact zeros


def test_oracle_reproduces_the_first_iterations_of_the_reference_training_loop():
    """g12 (four iterations of the reference's loop, train1.py:165-179): iterations 0 and 1 run on the initial weights -- their losses are
    what oracle/train_oracle.py computes with the CoV weights the fixture recorded (the oracle takes the alphas as an input)."""
    from oracle import train_oracle
    g = np.load(os.path.join(G, "g12_train_trajectory.npz"))
    sd_full = deterministic_state_dict(0)
    names = [str(n) for n in g["keep_names"]]
    for it in range(int(g["iters_to_accumulate"])):
        x = torch.from_numpy(synth.page_images(2000 + it, 2, 128, 128)).permute(0, 3, 1, 2)
        label, idmap = synth.train_labels(2100 + it, 2, 32, 32)
        keep = {n: torch.from_numpy(k) for n, k in zip(names, g["keep"][it])}
        loss, raw, _, _ = train_oracle.train_step(sd_full, x, torch.from_numpy(label), torch.from_numpy(idmap).long(), keep, g["alphas"][it].tolist(), 1.0)
        assert abs(float(loss) - float(g["loss"][it])) < 1e-5 * float(g["loss"][it]), it
        for j, k in enumerate(train_oracle.COV_KEYS):
            assert abs(float(raw[k]) - float(g["raw"][it][j])) < 2e-5 * max(1e-3, abs(float(g["raw"][it][j]))), (it, k)


class _ReplayMaps:
    def __init__(self, heat, feat):
        self.h, self.f, self.k = heat, feat, 0

    def __call__(self, image_input):
        r = (self.h[self.k:self.k + 1], self.f[self.k:self.k + 1])
        self.k += 1
        return r


def test_demo_script_eval_matches_reference():
    """g13 (tests/golden/gen_golden.py::gen_demo_eval: the source range of ``eval`` of /root/reference/test_image1_torch.py exec'd with a replayed
    detector): the coarse pass, the page with and without the two-pass seed rows.  decode_oracle.eval_demo / page_merge(variant="demo")
    must return the reference's float64 rows exactly; and the demo variant must actually differ from the production selection here."""
    g = np.load(os.path.join(G, "g13_demo_eval.npz"))
    T = int(g["tile"][0])
    ph, pw = (int(v) for v in g["page"])
    img = np.full((ph, pw, 3), 255.0, np.float32)
    ds = [{"input": None, "offsetx": int(x), "offsety": int(y)} for y, x in g["offsets"]]
    l0, g0, *_ = decode_oracle.eval_demo([{"input": None, "offsetx": 0, "offsety": 0}], np.zeros((T, T, 3), np.float32),
                                         _ReplayMaps(g["coarse_heat"], g["coarse_feat"]), 0.4, tile=T)
    assert np.array_equal(l0, g["coarse_locations"]) and np.array_equal(g0, g["coarse_glyphfeatures"]) and l0.dtype == np.float64
    l0s = l0.copy()
    l0s[:, 1:] = l0s[:, 1:] * float(g["seed_scale"][0])
    loc, gf, *_ = decode_oracle.eval_demo(ds, img, _ReplayMaps(g["heat"], g["feat"]), 0.4, l0s, g0, tile=T)
    assert np.array_equal(loc, g["locations"]) and np.array_equal(gf, g["glyphfeatures"])
    loc2, gf2, *_ = decode_oracle.eval_demo(ds, img, _ReplayMaps(g["heat"], g["feat"]), 0.4, tile=T)
    assert np.array_equal(loc2, g["locations_noseed"]) and np.array_equal(gf2, g["glyphfeatures_noseed"])
    assert len(loc) != len(loc2) and len(loc) > 30
    # the same candidates through the production rules (white page: the contrast filter of a uniform crop is 0 < NaN-free threshold 0 -> keeps all):
    cand, cfe, canv = decode_oracle.eval_demo(ds, img, _ReplayMaps(g["heat"], g["feat"]), 0.4, tile=T, return_candidates=True)
    prod, _ = decode_oracle.page_merge(cand.copy(), cfe.copy(), img, canv[2], canv[3:], 0.4)
    demo, _ = decode_oracle.page_merge(cand.copy(), cfe.copy(), img, canv[2], canv[3:], 0.4, variant="demo")
    assert np.array_equal(demo, loc2) and prod.shape != demo.shape or not np.array_equal(prod, demo.astype(np.float32))
