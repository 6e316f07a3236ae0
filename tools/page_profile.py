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

# ---- phase timers (synchronising; development aid): where the non-merge time of a page goes
import ctypes as C
from findtextcenternet_amd import _lib as L
from findtextcenternet_amd.decode import decode_peaks, TileGeom, tile_keep_rect, tiles_to_device
lib = L.load()
dev = torch.device("cuda")
def T(name, fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); print(f"{name:40s} {1e3 * (time.perf_counter() - t0) / reps:7.2f} ms"); return r
ph, pw = page.padded_page_size(3508, 2480, pd.stepx, pd.stepy)
origins = page.tile_origins(ph, pw, pd.stepx, pd.stepy)
page_dev = T("upload uint8 page", lambda: torch.from_numpy(np.ascontiguousarray(img)).to(dev))
def gather(lo, hi):
    o = torch.tensor(origins[lo:hi], dtype=torch.int32, device=dev)
    out = torch.empty((hi - lo, 768, 768, 3), dtype=torch.float32, device=dev)
    L.check(lib.ftc_tile_gather(page_dev.data_ptr(), 3508, 2480, o.data_ptr(), hi - lo, 768, 768, out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "g")
    return out
x8 = T("tile gather (8 tiles)", lambda: gather(0, 8))
hf = T("forward_nhwc (8 tiles, alloc outputs)", lambda: det.forward_nhwc(x8.permute(0, 3, 1, 2)))
geoms = [TileGeom(ox, oy, pw, ph, tile_keep_rect(ox, oy, pw, ph, 0.6)) for (oy, ox) in origins[:8]]
tl = T("TileGeom + tiles_to_device", lambda: tiles_to_device([TileGeom(ox, oy, pw, ph, tile_keep_rect(ox, oy, pw, ph, 0.6)) for (oy, ox) in origins[:8]], dev, 192, 192))
T("decode_peaks (fresh workspace)", lambda: decode_peaks(hf[0], hf[1], tl, cut_off=0.4, max_boxes=4096))
padded = np.full((ph, pw, 3), 255, np.uint8); padded[:3508, :2480] = img
T("upload padded uint8 + float()", lambda: torch.from_numpy(padded).to(dev).float())
T("host pad (np.full + copy)", lambda: np.full((ph, pw, 3), 255, np.uint8).__setitem__((slice(0, 3508), slice(0, 2480)), img))
