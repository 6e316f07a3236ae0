import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from findtextcenternet_amd import CenterNetDetector, TextDetectorModel, deterministic_state_dict, synth, PageDetector
import findtextcenternet_amd.page as page
m = TextDetectorModel(pre_weights=False, precision="bf16"); m.load_state_dict(deterministic_state_dict(0))
det = CenterNetDetector(m.detector).to("cuda").eval()
pd = PageDetector(det, batch=8, lanes=2)
img = synth.page_uint8(31, 3508, 2480)
pd.detect_page(img)
orig = page.page_merge_gpu
acc = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize(); acc[name] = acc.get(name, 0) + time.perf_counter() - t0; return r
    return w
page.page_merge_gpu = timed("page_merge_gpu", orig)
for _ in range(3):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pd.detect_page(img)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print("total %.1f ms" % (1e3 * tot), {k: round(1e3 * v, 1) for k, v in acc.items()}, "boxes", len(out[0]))
# pieces
t0 = time.perf_counter(); org = np.full((3840, 2688, 3), 255, np.uint8); org[:3508, :2480] = img; f = org.astype(np.float32); print("host pad+astype %.1f ms" % (1e3 * (time.perf_counter() - t0)))
t0 = time.perf_counter(); d = torch.from_numpy(f).cuda(); torch.cuda.synchronize(); print("upload float page %.1f ms" % (1e3 * (time.perf_counter() - t0)))
