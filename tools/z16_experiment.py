"""What-if for the next round: the reference trains under bf16 autocast and STORES its conv outputs in bf16; this repo keeps them fp32 (its
gradient gate is the fp32 reference).  FTC_TRAIN_EMULATE_Z16=1 rounds every conv output to the compute type and back (numerics of 16-bit
storage, same bytes): how far do the gradients move against the reference's fp32 gradients (golden g10)?
    python tools/z16_experiment.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from findtextcenternet_amd import TextDetectorModel, TrainStep, synth  # noqa: E402

g10 = np.load(os.path.join(ROOT, "tests", "golden", "g10_train_step.npz"), allow_pickle=True)
sd = {str(k): torch.from_numpy(v) for k, v in zip(g10["sd_names"], [g10[f"sd{i}"] for i in range(len(g10["sd_names"]))])} if "sd_names" in g10 else None


def model(precision):
    from findtextcenternet_amd import deterministic_state_dict
    m = TextDetectorModel(pre_weights=False, precision=precision)
    m.load_state_dict(deterministic_state_dict(0))
    return m.to("cuda").train()


B, H, W = 2, 256, 256
x = torch.from_numpy(synth.page_images(1029, B, H, W)).permute(0, 3, 1, 2).cuda()
label, idmap = synth.train_labels(1030, B, H // 4, W // 4)
keep = {str(n): torch.from_numpy(k) for n, k in zip(g10["keep_names"], g10["keep"])}
amax = dict(zip([str(n) for n in g10["grad_names"]], g10["grad_absmax"]))
for precision in ("bf16", "fp16"):
    for emu in ("0", "1"):
        os.environ["FTC_TRAIN_EMULATE_Z16"] = emu
        ts = TrainStep(model(precision))
        ts.zero_grad()
        loss, _ = ts.forward_backward(x, torch.from_numpy(label).cuda(), torch.from_numpy(idmap).cuda(), keep=keep)
        cos = []
        for i, n in enumerate(g10["pick_names"]):
            n = str(n)
            sib = n[:-4] + "weight" if n.endswith(".bias") else n
            if amax[n] < 1e-5 * amax.get(sib, amax[n]) or amax[n] == 0.0:
                continue
            ref, st = g10[f"pick{i}"], int(g10[f"pick{i}_stride"])
            mine = dict(ts.params)[n].grad.detach().float().cpu().numpy().reshape(-1)[::st]
            cos.append((float((mine * ref).sum() / (np.linalg.norm(mine) * np.linalg.norm(ref) + 1e-30)), n))
        cos.sort()
        top = [c for c, n in cos if ".upsamplers.3." in n or ".top_conv." in n]
        print(f"{precision} z16-emulation={emu}: loss {float(loss):.5f} (reference {float(g10['loss']):.5f}); cosine to the fp32 reference gradients: "
              f"min {cos[0][0]:.3f}  p10 {cos[len(cos) // 10][0]:.3f}  median {cos[len(cos) // 2][0]:.3f}  last-level/top min {min(top):.3f}  ({len(cos)} tensors)", flush=True)
        del ts
        torch.cuda.empty_cache()
