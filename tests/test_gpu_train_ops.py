"""-m gpu: the validation-step adjuncts on MI355X -- get_fmask, TextDetectorModel.forward with the
boolean-mask gather and SimpleDecoder, loss_function / heatmap_loss, CoVWeightingLoss -- against the reference's own outputs
(tests/golden/g7_validation_step.npz, written by its models/detector.py + loss_func.py on CPU) and against the CPU oracle
(oracle/loss_oracle.py, pinned by the same fixture) on fresh seeded inputs.

Tolerances: integer results (masks, index lists, correct / total) bit-exact; fp32 losses within 2e-5 relative (summation order);
network outputs within the detector's 1e-3 (fp32 mode)."""
import os

import numpy as np
import pytest
import torch

import synth
from findtextcenternet_amd import deterministic_state_dict
from gpu_harness import fresh_model
from findtextcenternet_amd import loss_func as LF
from oracle import loss_oracle

pytestmark = pytest.mark.gpu
KEYS = ["keymap_loss", "size_loss", "textline_loss", "separator_loss", "id_loss", "code1_loss", "code2_loss", "code4_loss", "code8_loss"]


@pytest.fixture(scope="module")
def g7(golden_dir):
    return np.load(os.path.join(golden_dir, "g7_validation_step.npz"))


@pytest.fixture(scope="module")
def model_fp32():
    return fresh_model("fp32").to("cuda").eval()


def _labels(B=2, hw=64, seed=616):
    label, idmap = synth.train_labels(seed, B, hw, hw)
    return torch.from_numpy(label).cuda(), torch.from_numpy(idmap).to(torch.long).cuda()


def test_topk_mask_kernel_edge_cases(model_fp32):
    rng = np.random.Generator(np.random.PCG64(3))
    for n, k_per in ((4096, 1024), (36864 * 2, 1024), (5000, 1024)):
        v = rng.standard_normal(n).astype(np.float32)
        v[rng.integers(0, n, n // 3)] = 0.0                      # a large plateau of ties, possibly at the boundary
        v[rng.integers(0, n, n // 10)] = 1.0
        lab = torch.zeros((1, 5, 1, n), dtype=torch.float32)
        lab[0, 0, 0] = torch.from_numpy(v)
        for B in (1, 2, 4):
            if n % B:
                continue
            labB = lab.reshape(1, 5, 1, n).clone().reshape(5, n)
            labB = labB.reshape(5, B, 1, n // B).permute(1, 0, 2, 3).contiguous()
            want = loss_oracle.get_fmask(labB)
            got = model_fp32.get_fmask(labB.cuda(), None)
            assert got.dtype == torch.bool and int(got.sum()) == min(n, 1024 * B)
            assert torch.equal(got.cpu(), want), (n, B)
    # an existing mask tensor of the right shape is reused (reference: mask.fill_(0); mask[...] = True)
    lab, _ = _labels()
    m0 = torch.ones(2 * 64 * 64, dtype=torch.bool, device="cuda")
    m1 = model_fp32.get_fmask(lab, m0)
    assert m1.data_ptr() == m0.data_ptr() and int(m1.sum()) == 2048


def test_get_fmask_matches_reference(model_fp32, g7):
    lab, _ = _labels()
    fm = model_fp32.get_fmask(lab, None)
    assert np.array_equal(np.packbits(fm.cpu().numpy()), g7["fmask"])


def test_validation_forward_matches_reference(model_fp32, g7):
    """TextDetectorModel.forward(image, fmask) in eval mode (models/detector.py:262-268): 9-channel heat-map + three decoder
    logit arrays of the 1024*B masked pixels, in the reference's row order."""
    lab, ids = _labels()
    x = torch.from_numpy(synth.page_images(515, 2, 256, 256)).permute(0, 3, 1, 2).cuda()
    fm = model_fp32.get_fmask(lab, None)
    with torch.no_grad():
        heatmap, dec = model_fp32(x, fm)
    assert heatmap.shape == (2, 9, 64, 64) and [tuple(d.shape) for d in dec] == [(2048, 1091), (2048, 1093), (2048, 1097)]
    assert float((heatmap.cpu() - torch.from_numpy(g7["heatmap"])).abs().max()) < 1e-3
    rows = torch.from_numpy(g7["dec_rows"])
    for j in range(3):
        d = dec[j].cpu()
        e = float((d[rows] - torch.from_numpy(g7[f"dec{j}_at"])).abs().max())
        assert e < 5e-3, (j, e)                                  # logits of O(10) through two 2048-wide fp32 GEMMs
        assert float((torch.logsumexp(d, 1) - torch.from_numpy(g7[f"dec{j}_lse"])).abs().max()) < 5e-3
        am, ref_am = d.argmax(1).numpy(), g7[f"dec{j}_argmax"]
        top2 = torch.topk(d, 2, dim=1).values
        clear = ((top2[:, 0] - top2[:, 1]) > 1e-2).numpy()       # arg-max may differ only where the two best logits are within noise
        assert np.array_equal(am[clear], ref_am[clear]) and clear.mean() > 0.95
    # loss_function on the GPU outputs == the reference's loss_function on its own outputs
    out = LF.loss_function(fm, lab, ids, heatmap, dec)
    for k in ["loss"] + KEYS:
        want = float(g7["loss_" + k])
        assert abs(float(out[k]) - want) <= 2e-3 * max(1.0, abs(want)), (k, float(out[k]), want)
    assert float(out["correct"]) == float(g7["loss_correct"]) and float(out["total"]) == float(g7["loss_total"])
    # SimpleDecoder is callable on its own rows too (CodeDecoder-style use, models/detector.py:298-305)
    feats = torch.randn(7, 100, device="cuda")
    d7 = model_fp32.decoder(feats)
    want7 = loss_oracle.simple_decoder(deterministic_state_dict(0), feats.cpu())
    for a, b in zip(d7, want7):
        assert float((a.cpu() - b).abs().max()) < 5e-3


def test_loss_function_synthetic_matches_reference(g7):
    """The loss kernels alone on seeded maps / logits: against the reference's loss_function (g7 loss2_*)."""
    lab, ids = _labels()
    fm = torch.from_numpy(np.unpackbits(g7["fmask"])[:2 * 64 * 64].astype(bool)).cuda()
    tgt = ids[:, 0].flatten()[fm].cpu().numpy()
    hm2, dec2 = synth.loss_case(717, 2, 64, 64, tgt)
    out = LF.loss_function(fm, lab, ids, torch.from_numpy(hm2).cuda(), [torch.from_numpy(d).cuda() for d in dec2])
    for k in ["loss"] + KEYS:
        want = float(g7["loss2_" + k])
        assert abs(float(out[k]) - want) <= 2e-5 * max(1.0, abs(want)), (k, float(out[k]), want)
    assert float(out["correct"]) == float(g7["loss2_correct"]) and float(out["total"]) == float(g7["loss2_total"])
    # NHWC memory behind the NCHW view (what the detector returns) gives the same numbers as a contiguous NCHW tensor
    hm_cl = torch.from_numpy(hm2).cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    out2 = LF.loss_function(fm, lab, ids, hm_cl, [torch.from_numpy(d).cuda() for d in dec2])
    assert all(float(out2[k]) == float(out[k]) for k in ["loss"] + KEYS)
    hl = LF.heatmap_loss(lab[:, 0], torch.from_numpy(hm2).cuda()[:, 0])
    assert abs(float(hl) * 10 - float(g7["loss2_keymap_loss"])) < 2e-5 * float(g7["loss2_keymap_loss"])


@pytest.mark.parametrize("seed", [1, 2])
def test_loss_function_fresh_inputs_vs_oracle(seed):
    B, hw = 3, 48
    label, idmap = synth.train_labels(900 + seed, B, hw, hw)
    lab_c, ids_c = torch.from_numpy(label), torch.from_numpy(idmap).to(torch.long)
    fm_c = loss_oracle.get_fmask(lab_c)
    tgt = ids_c[:, 0].flatten()[fm_c].numpy()
    hm, dec = synth.loss_case(40 + seed, B, hw, hw, tgt)
    want = loss_oracle.loss_function(fm_c, lab_c, ids_c, torch.from_numpy(hm), [torch.from_numpy(d) for d in dec])
    got = LF.loss_function(fm_c.cuda(), lab_c.cuda(), ids_c.cuda(), torch.from_numpy(hm).cuda(), [torch.from_numpy(d).cuda() for d in dec])
    for k in ["loss"] + KEYS:
        assert abs(float(got[k]) - float(want[k])) <= 2e-5 * max(1.0, abs(float(want[k]))), (k, float(got[k]), float(want[k]))
    assert float(got["correct"]) == float(want["correct"]) and float(got["total"]) == float(want["total"])


def test_cov_weighting_matches_reference(g7):
    keys, seq = synth.cov_loss_sequence(818)
    cov = LF.CoVWeightingLoss(device="cuda", losses=keys)
    for step, vals in enumerate(seq):
        loss = cov({k: torch.tensor(float(v), dtype=torch.float32, device="cuda") for k, v in zip(keys, vals)})
        assert abs(float(loss) - float(g7["cov_loss"][step])) < 2e-6 * max(1.0, abs(float(loss))), (step, float(loss))
        assert np.abs(cov.alphas.cpu().numpy() - g7["cov_alphas"][step]).max() < 2e-6


def test_validation_step_bf16_mode_tracks_fp32(model_fp32):
    """The speed mode runs the decoder GEMMs in bf16: same decisions on clear rows, losses within bf16 noise."""
    m = fresh_model("bf16").to("cuda").eval()
    lab, ids = _labels()
    x = torch.from_numpy(synth.page_images(515, 2, 256, 256)).permute(0, 3, 1, 2).cuda()
    fm = m.get_fmask(lab, None)
    with torch.no_grad():
        h16, d16 = m(x, fm)
        h32, d32 = model_fp32(x, fm)
    assert float((h16 - h32).abs().max()) < 0.3
    for a, b in zip(d16, d32):
        assert float((a - b).abs().max()) < 0.05 * float(b.abs().max())
    l16, l32 = LF.loss_function(fm, lab, ids, h16, d16), LF.loss_function(fm, lab, ids, h32, d32)
    assert abs(float(l16["loss"]) - float(l32["loss"])) < 0.05 * float(l32["loss"])


# measured on MI355X (maps range 39): fp32 5.2e-4 / 1.9e-5, fp16 0.36 / 1.5e-2, bf16 2.1 / 9.3e-2 -- batch statistics over as few as 48 samples (4x4 maps x
# batch 3 in the last stages) amplify the operand rounding of the 16-bit modes; the fp16 : bf16 ratio is the 8x of their mantissas
@pytest.mark.parametrize("precision,tol_maps,tol_stat", [("fp32", 2e-3, 2e-4), ("fp16", 1.0, 5e-2), ("bf16", 5.0, 0.25)])
def test_train_mode_forward_matches_oracle_and_reference(precision, tol_maps, tol_stat, golden_dir):
    """model.train() + forward under no_grad (the reference's BN-refresh pass, train1.py:203-211): batch-statistics BatchNorm,
    StochasticDepth with the draw saved in golden g9, running statistics moved -- against the reference's own train()-mode outputs
    (g9) and, for every BatchNorm layer, against the CPU oracle's new running statistics."""
    from oracle import detector_oracle
    g = np.load(os.path.join(golden_dir, "g9_train_forward.npz"))
    B, H, W = 3, 128, 128
    sd = deterministic_state_dict(0)
    m = fresh_model(precision)
    m = m.to("cuda")
    x = torch.from_numpy(synth.page_images(929, B, H, W)).cuda().permute(0, 3, 1, 2)
    label, _ = synth.train_labels(930, B, H // 4, W // 4)
    keep = {str(n): torch.from_numpy(k) for n, k in zip(g["keep_names"], g["keep"])}
    m.train()
    with pytest.raises(ValueError):
        m(x, None)                                           # gradients enabled = the train step's forward (round 5): it needs the pixel mask
    m.stochastic_depth_keep = keep
    with torch.no_grad():
        fmask = m.get_fmask(torch.from_numpy(label).cuda(), None)
        assert np.array_equal(np.packbits(fmask.cpu().numpy()), g["fmask"])
        maps, dec = m(x, fmask)
    err = float((maps.cpu() - torch.from_numpy(g["maps"])).abs().max())
    print(f"train-mode forward {precision}: maps L-inf vs reference {err:.3e} (range {float(g['maps'].max() - g['maps'].min()):.1f})")
    assert err < tol_maps
    # running statistics: every BatchNorm of the model against the oracle, the saved subset against the reference
    x_cpu = torch.from_numpy(synth.page_images(929, B, H, W)).permute(0, 3, 1, 2)
    o_maps, o_feat, new = detector_oracle.detection_forward_train(sd, x_cpu, keep)
    new = {"detector." + k: v for k, v in new.items()}
    rows = o_feat.permute(0, 2, 3, 1).flatten(0, -2)[fmask.cpu()]
    o_dec, new_dec = detector_oracle.decoder_forward_train(sd, rows)
    new.update(new_dec)
    got = {k: v.cpu() for k, v in m.state_dict().items()}
    worst = 0.0
    for k, v in new.items():
        d = float((got[k] - v).abs().max()) / max(1.0, float(v.abs().max()))
        worst = max(worst, d)
        assert d < tol_stat, (k, d)
    assert len(new) == int(g["n_changed"])
    for i, k in enumerate(g["stat_names"]):
        want = torch.from_numpy(g[f"stat{i}"])
        assert float((got[str(k)] - want).abs().max()) <= tol_stat * max(1.0, float(want.abs().max())), k
    assert all(int(v) == int(sd[k]) + 1 for k, v in got.items() if k.endswith("num_batches_tracked"))
    dtol = 2e-2 if precision == "fp32" else 3.0
    for j in range(3):
        assert float((dec[j].cpu()[torch.from_numpy(g["dec_rows"])] - torch.from_numpy(g[f"dec{j}_at"])).abs().max()) < dtol
    print(f"train-mode forward {precision}: worst running-stat deviation {worst:.3e}")
    # back to eval(): the inference engine refolds the refreshed statistics
    m.eval()
    with torch.no_grad():
        hm, _ = m(x, fmask)
    assert torch.isfinite(hm).all()


@pytest.mark.parametrize("size", ["s", "m"])
def test_train_mode_forward_small_models_vs_oracle(size):
    """The training-mode op list is built from the state_dict alone: EfficientNetV2-S / -M backbones (other stage tables and tap widths,
    models/detector.py:131-136) against the CPU oracle, fp32, random StochasticDepth draw supplied to both."""
    from oracle import detector_oracle
    B, H, W = 2, 128, 128
    sd = deterministic_state_dict(0, model_size=size)
    m = fresh_model("fp32", model_size=size)
    m = m.to("cuda").train()
    x_cpu = torch.from_numpy(synth.page_images(31, B, H, W)).permute(0, 3, 1, 2)
    label, _ = synth.train_labels(32, B, H // 4, W // 4)
    sd_det = {k[len("detector."):]: v for k, v in sd.items() if k.startswith("detector.")}
    probs = detector_oracle.stochastic_depth_probs(sd_det)
    g = torch.Generator().manual_seed(33)
    keep = {p: (torch.rand(B, generator=g) < 1.0 - probs[p]).float() / (1.0 - probs[p]) for p in detector_oracle.residual_blocks(sd_det)}
    m.stochastic_depth_keep = keep
    with torch.no_grad():
        fmask = m.get_fmask(torch.from_numpy(label).cuda(), None)
        maps, dec = m(x_cpu.cuda(), fmask)
    o_maps, o_feat, new = detector_oracle.detection_forward_train(sd, x_cpu, {"detector." + k: v for k, v in keep.items()})
    scale = float(o_maps.abs().max())
    assert float((maps.cpu() - o_maps).abs().max()) < 1e-4 * max(1.0, scale)
    got = {k: v.cpu() for k, v in m.state_dict().items()}
    for k, v in new.items():
        assert float((got["detector." + k] - v).abs().max()) <= 2e-4 * max(1.0, float(v.abs().max())), k
    rows = o_feat.permute(0, 2, 3, 1).flatten(0, -2)[fmask.cpu()]
    o_dec, _ = detector_oracle.decoder_forward_train(sd, rows)
    for j in range(3):
        assert float((dec[j].cpu() - o_dec[j]).abs().max()) < 2e-2 * max(1.0, float(o_dec[j].abs().max()))
