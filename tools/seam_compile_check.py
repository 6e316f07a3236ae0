#!/usr/bin/env python3
"""Does the reference's `@torch.compile def train_step(...)` (train1.py:124-131) run on top of findtextcenternet_amd's modules?  The HIP entry points are
torch.compiler.disable'd, so Dynamo should break the graph around them and run the rest eagerly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import synth
from findtextcenternet_amd import AdamWScheduleFree, TextDetectorModel, deterministic_state_dict
from findtextcenternet_amd.loss_func import CoVWeightingLoss, loss_function
from findtextcenternet_amd.train_step import COV_KEYS

model = TextDetectorModel(pre_weights=False, precision="bf16")
model.load_state_dict(deterministic_state_dict(0))
model = model.to("cuda").train()
opt = AdamWScheduleFree(list(model.parameters()), lr=1e-4)
CoWloss = CoVWeightingLoss(device="cuda", losses=COV_KEYS)


@torch.compile
def train_step(image, map, idmap, fmask):
    with torch.autocast(device_type='cuda', dtype=torch.bfloat16):
        heatmap, decoder_outputs = model(image, fmask)
        rawloss = loss_function(fmask, map, idmap, heatmap, decoder_outputs)
        loss = CoWloss(rawloss)
    return loss, rawloss


B, H, W = 2, 128, 128
x = torch.from_numpy(synth.page_images(5, B, H, W)).permute(0, 3, 1, 2).cuda()
lab, idm = synth.train_labels(6, B, H // 4, W // 4)
lab, idm = torch.from_numpy(lab).cuda(), torch.from_numpy(idm).cuda().long()
opt.train(); CoWloss.train(); opt.zero_grad()
fmask = None
for i in range(3):
    t0 = time.time()
    fmask = model.get_fmask(lab, fmask)
    loss, rawloss = train_step(x, lab, idm, fmask)
    (loss / 1).backward()
    opt.step(); opt.zero_grad()
    torch.cuda.synchronize()
    print(f"iter {i}: loss {float(loss):.5f} raw {float(rawloss['loss']):.5f}  {time.time() - t0:.2f} s", flush=True)
print("TORCH_COMPILE_OK")
