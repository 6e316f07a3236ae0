#!/usr/bin/env python3
"""Host-side cost of ftc_plan_run (no synchronisation inside the timed region)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from findtextcenternet_amd import CenterNetDetector, TextDetectorModel, deterministic_state_dict  # noqa: E402
from findtextcenternet_amd import _lib as L  # noqa: E402

model = TextDetectorModel(pre_weights=False, precision="bf16")
model.load_state_dict(deterministic_state_dict(0))
det = CenterNetDetector(model.detector).to("cuda").eval()
B = 8
x = torch.rand((B, 768, 768, 3), device="cuda").permute(0, 3, 1, 2)
with torch.no_grad():
    det.forward_nhwc(x)
torch.cuda.synchronize()
lib = L.load()
eng = model.detector._engine
pl = eng.plan(B, 768, 768, False)
heat = torch.empty((B, pl.h, pl.w, 10), dtype=torch.float32, device="cuda")
feat = torch.empty((B, pl.h, pl.w, 100), dtype=torch.float32, device="cuda")
bases = (C.c_void_p * L.NUM_BASES)(None, eng.workspace.data_ptr(), eng.wdev.data_ptr(), x.data_ptr(), heat.data_ptr(), feat.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lib.ftc_plan_run(pl.handle, bases, st, 0, -1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"ftc_plan_run host {1e3 * (t1 - t0):.3f} ms for {len(pl.ops)} ops ({1e6 * (t1 - t0) / len(pl.ops):.1f} us/op), until idle {1e3 * (t2 - t0):.3f} ms")
# only the short-kernel stretch (stage 6)
names = [m.name for m in pl.meta]
a = next(i for i, n in enumerate(names) if n.startswith("backbone.features.6.1."))
b = next(i for i, n in enumerate(names) if n.startswith("backbone.features.7.0."))
torch.cuda.synchronize()
t0 = time.perf_counter()
lib.ftc_plan_run(pl.handle, bases, st, a, b - 1)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"stage 6 blocks 1..: host {1e3 * (t1 - t0):.3f} ms for {b - a} ops, until idle {1e3 * (t2 - t0):.3f} ms")
