"""Diagnostic: the fp32 train step on the GPU vs tests/golden/g10_train_step.npz, every parameter listed (gradient-norm ratio and the
entry-wise error of the stored gradients).  `python tools/train_step_check.py [precision]`"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from findtextcenternet_amd import TextDetectorModel, deterministic_state_dict  # noqa: E402
from findtextcenternet_amd.train_step import TrainStep  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
g = np.load(os.path.join(ROOT, "tests", "golden", "g10_train_step.npz"))
B, H, W = 2, 256, 256
m = TextDetectorModel(pre_weights=False, precision=prec)
m.load_state_dict(deterministic_state_dict(0))
m = m.to("cuda").train()
ts = TrainStep(m)
x = torch.from_numpy(synth.page_images(1029, B, H, W)).permute(0, 3, 1, 2).cuda()
label, idmap = synth.train_labels(1030, B, H // 4, W // 4)
keep = {str(n): torch.from_numpy(k) for n, k in zip(g["keep_names"], g["keep"])}
ts.zero_grad()
loss, raw = ts.forward_backward(x, torch.from_numpy(label).cuda(), torch.from_numpy(idmap).cuda(), keep=keep)
torch.cuda.synchronize()
print("loss", float(loss), "ref", float(g["loss"]))
for k, v in raw.items():
    print("  ", k, float(v), float(g["raw_" + k]) if "raw_" + k in g else "")
maps = ts.maps(B, H, W).cpu().numpy()
print("maps err", np.abs(maps - g["heatmap"]).max())
names = [str(n) for n in g["grad_names"]]
norms = dict(zip(names, g["grad_norms"]))
pick = {str(n): i for i, n in enumerate(g["pick_names"])}
nbad = 0
for n, p in reversed(ts.params):
    mine = float(p.grad.double().norm())
    ref = norms[n]
    line = f"{n:70s} norm {mine:.4e} ref {ref:.4e} ratio {mine / max(ref, 1e-30):.4f}"
    ok = abs(mine - ref) <= 1e-3 * ref + 1e-7 * np.sqrt(p.numel())
    if n in pick:
        i = pick[n]
        st, r = int(g[f"pick{i}_stride"]), g[f"pick{i}"]
        mv = p.grad.detach().float().cpu().numpy().reshape(-1)[::st]
        e = np.abs(mv - r).max()
        cos = float((mv * r).sum() / (np.linalg.norm(mv) * np.linalg.norm(r) + 1e-30))
        line += f" | entry err {e:.2e} of {np.abs(r).max():.2e} cos {cos:.4f}"
        ok = ok and e <= 1e-3 * np.abs(r).max() + 1e-7
    if not ok:
        nbad += 1
    if not ok or n in pick:
        print(("BAD " if not ok else "ok  ") + line)
print("bad:", nbad, "of", len(ts.params))
