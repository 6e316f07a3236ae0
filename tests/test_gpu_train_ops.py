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
from findtextcenternet_amd import TextDetectorModel, deterministic_state_dict
from findtextcenternet_amd import loss_func as LF
from oracle import loss_oracle

pytestmark = pytest.mark.gpu
KEYS = ["keymap_loss", "size_loss", "textline_loss", "separator_loss", "id_loss", "code1_loss", "code2_loss", "code4_loss", "code8_loss"]


@pytest.fixture(scope="module")
def g7(golden_dir):
    return np.load(os.path.join(golden_dir, "g7_validation_step.npz"))


@pytest.fixture(scope="module")
def model_fp32():
    m = TextDetectorModel(pre_weights=False, precision="fp32")
    m.load_state_dict(deterministic_state_dict(0))
    return m.to("cuda").eval()


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
    m = TextDetectorModel(pre_weights=False, precision="bf16")
    m.load_state_dict(deterministic_state_dict(0))
    m.to("cuda").eval()
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
