#!/usr/bin/env python3
"""Experiment (GPU): where does the bf16 speed mode lose peak agreement with the fp32 reference golden?
Runs the 768x768 page fixture with the backbone / the heads switched between bf16 and fp32 (FTC_EXP_* switches of plan.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import synth
from findtextcenternet_amd import CenterNetDetector, TextDetectorModel, deterministic_state_dict, tile_keep_rect
from oracle import decode_oracle


def run(tag, env):
    for k in ("FTC_EXP_BACKBONE", "FTC_EXP_HEADS", "FTC_EXP_KEYHEAD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    sd = deterministic_state_dict(0)
    m = TextDetectorModel(pre_weights=False, precision="bf16")
    m.load_state_dict(sd)
    d = CenterNetDetector(m.detector).to("cuda").eval()
    out = []
    for name, seed in (("page", 4242),):
        g = np.load(os.path.join(ROOT, "tests", "golden", f"g2_fwd768_{name}.npz"))
        x = torch.from_numpy(synth.page_images(seed, 1, 768, 768)).permute(0, 3, 1, 2).to("cuda")
        with torch.no_grad():
            hm, ft = d(x)
        hm = hm.cpu().numpy()
        gh = g["heatmap"]
        both = np.isfinite(hm) & np.isfinite(gh)
        e = float(np.abs(hm[both] - gh[both]).max())
        ek = float(np.abs(hm[:, 0] - gh[:, 0]).max())
        ekm = float(np.abs(hm[:, 0] - gh[:, 0]).mean())
        rect = tile_keep_rect(0, 0, 768, 768, 0.6)
        z = np.zeros((1, 100, 192, 192), np.float32)
        _, _, ir = decode_oracle.decode_tile(gh, z, 0, 0, 768, 768, 0.4, rect)
        _, _, ib = decode_oracle.decode_tile(hm, z, 0, 0, 768, 768, 0.4, rect)
        inter = len(set(ir) & set(ib))
        jac = inter / max(1, len(set(ir) | set(ib)))
        out.append(f"{name}: Linf {e:.3e} key Linf {ek:.3e} key mean|e| {ekm:.3e} peaks ref {len(ir)} got {len(ib)} common {inter} jaccard {jac:.3f}")
    print(f"[{tag}] " + " | ".join(out), flush=True)


if __name__ == "__main__":
    run("all bf16", {})
    run("backbone fp32, heads bf16", {"FTC_EXP_BACKBONE": "fp32"})
    run("backbone bf16, heads fp32", {"FTC_EXP_HEADS": "fp32"})
    run("all fp32", {"FTC_EXP_BACKBONE": "fp32", "FTC_EXP_HEADS": "fp32"})
