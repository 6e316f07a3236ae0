#!/usr/bin/env python3
"""ftc_page_order + ftc_page_merge on the dense test page, the scratch header (n_keep, ticket, use_seq, lock, edges, first stalled wait)
printed: python tools/page_merge_debug.py [n_boxes]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_gpu_page as T  # noqa: E402
from findtextcenternet_amd import _lib as L  # noqa: E402
from oracle import decode_oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dense = len(sys.argv) > 2 and sys.argv[2] == "dense"               # the dense page of the tests: large boxes, long dependency chains
loc32, feats, img, seps, code_all = T._merge_case(11, n, 1228, 1228, 420.0, 60.0) if dense else T._merge_case(1, n, 900, 1100, 90.0, 14.0)
lib = L.load()
dev = torch.device("cuda")
N = loc32.shape[0]
mh, mw = seps.shape
boxes = torch.from_numpy(loc32).to(dev)
page_d = torch.from_numpy(img).to(dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
hist = torch.empty((2, N), dtype=torch.float64, device=dev)
L.check(lib.ftc_box_hists(boxes.data_ptr(), N, page_d.data_ptr(), img.shape[0], img.shape[1], C.c_float(0.4), hist.data_ptr(), st), "hists")
order = torch.empty((N,), dtype=torch.int32, device=dev)
th = torch.empty((1,), dtype=torch.float64, device=dev)
ob = int(lib.ftc_page_order_scratch_bytes(N))
osc = torch.empty(ob, dtype=torch.uint8, device=dev)
L.check(lib.ftc_page_order(boxes.data_ptr(), N, hist[0].data_ptr(), C.c_float(0.4), order.data_ptr(), th.data_ptr(), osc.data_ptr(), ob, st), "order")
torch.cuda.synchronize()
M = int((loc32[:, 0] >= np.float32(0.4)).sum())
print("order ok:", np.array_equal(order.cpu().numpy()[:M], np.argsort(-loc32[:, 0].astype(np.float64), kind="stable").astype(np.int32)[:M]), "th", float(th.item()), flush=True)
nbytes = int(lib.ftc_page_merge_scratch_bytes(N, img.shape[0], img.shape[1]))
scratch = torch.full((nbytes,), 0xCD, dtype=torch.uint8, device=dev)           # garbage: nothing may rely on a zeroed block
out_loc = torch.empty((N, 9), dtype=torch.float32, device=dev)
out_idx = torch.empty((N,), dtype=torch.int32, device=dev)
out_n = torch.zeros((1,), dtype=torch.int32, device=dev)
canv = torch.zeros((7, mh, mw), dtype=torch.float32, device=dev)
canv[2] = torch.from_numpy(seps).to(dev)
for k in range(4):
    canv[3 + k] = torch.from_numpy(code_all[k]).to(dev)
codes = canv[3:7].contiguous()
for rep in range(3):
    t0 = time.perf_counter()
    L.check(lib.ftc_page_merge(boxes.data_ptr(), order.data_ptr(), N, hist[1].data_ptr(), th.data_ptr(), C.c_float(0.4), canv[2].data_ptr(),
                               codes.data_ptr(), mh, mw, 4, img.shape[0], img.shape[1], out_loc.data_ptr(), out_idx.data_ptr(), out_n.data_ptr(), scratch.data_ptr(),
                               nbytes, st), "merge")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hdr = scratch[:32].view(torch.int32).cpu().numpy()
    print(f"rep {rep}: {dt * 1e3:.2f} ms  n_keep {hdr[0]} ticket {hdr[1]} use_seq {hdr[2]} lock {hdr[3]} edges {hdr[4]} stall r={hdr[5]} j={hdr[6]} n={hdr[7]}  out_n {int(out_n.item())}", flush=True)
ref_loc, ref_gf = decode_oracle.page_merge(loc32.astype(np.float64), feats.copy(), img, seps, code_all, 0.4)
k = int(out_n.item())
print("oracle kept", len(ref_loc), "identical:", k == len(ref_loc) and np.array_equal(out_loc[:k].cpu().numpy(), ref_loc.astype(np.float32)))
