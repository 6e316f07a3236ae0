"""What-if: the batch-1 forward (368 launches) captured in a HIP graph -- is the reference's one-tile-at-a-time calling convention launch-bound?
    python tools/graph_experiment.py [precision] [batch]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from findtextcenternet_amd import CenterNetDetector, TextDetectorModel, deterministic_state_dict  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = TextDetectorModel(pre_weights=False, precision=prec)
m.load_state_dict(deterministic_state_dict(0))
det = CenterNetDetector(m.detector).to("cuda").eval()
x = torch.rand((B, 768, 768, 3), device="cuda").permute(0, 3, 1, 2)
heat = torch.empty((B, 192, 192, 10), device="cuda")
feat = torch.empty((B, 192, 192, 100), device="cuda")
with torch.no_grad():
    for _ in range(3):
        det.forward_nhwc(x, out=(heat, feat))
torch.cuda.synchronize()
ref_h, ref_f = heat.clone(), feat.clone()


def bench(fn, n=50):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()                                  # latency: one call at a time
    return 1e3 * (time.perf_counter() - t0) / n


def eager():
    with torch.no_grad():
        det.forward_nhwc(x, out=(heat, feat))


print(f"eager        {bench(eager):7.3f} ms per forward (batch {B}, {prec})", flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eager()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    eager()
heat.zero_()
feat.zero_()
g.replay()
torch.cuda.synchronize()
print("graph replay identical to eager:", bool(torch.equal(heat, ref_h) and torch.equal(feat, ref_f)))
print(f"graph replay {bench(g.replay):7.3f} ms per forward", flush=True)
