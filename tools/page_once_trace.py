import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from findtextcenternet_amd import CenterNetDetector, TextDetectorModel, deterministic_state_dict, synth, PageDetector
m = TextDetectorModel(pre_weights=False, precision="bf16"); m.load_state_dict(deterministic_state_dict(0))
det = CenterNetDetector(m.detector).to("cuda").eval()
pd = PageDetector(det, batch=8, lanes=2)
img = synth.page_uint8(31, 3508, 2480)
for _ in range(3): pd.detect_page(img)
torch.cuda.synchronize()
# marker kernel: a distinctive fill
mark = torch.zeros(12345, device="cuda", dtype=torch.float64)
mark.fill_(1.0); torch.cuda.synchronize()
t0 = time.perf_counter(); out = pd.detect_page(img); torch.cuda.synchronize(); print("wall ms", 1e3 * (time.perf_counter() - t0))
mark.fill_(2.0); torch.cuda.synchronize()
