#!/usr/bin/env python3
"""Single-op timing + phase timeline (s_memtime of wave 0 of every workgroup) of FTC_OP_MBHEAD (csrc/mbconv_slice.hip) on the stage-6/7
shapes, and of the FTC_OP_SE that consumes its fc1 partial products.
    python tools/mbslice_bench.py [B ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from findtextcenternet_amd import _lib as L  # noqa: E402

lib = L.load()
batches = [int(a) for a in sys.argv[1:]] or [8]
XFLAGS = int(os.environ.get("MBS_FLAGS", "0"), 0)          # 0x100: the general (band) kernel on the 24x24 shapes
for B in batches:
    for (H, W, Cin, Cx, S, N, R) in [(24, 24, 512, 3072, 128, 512, 0), (24, 24, 640, 3840, 160, 640, 0), (48, 48, 256, 1536, 64, 256, 10), (48, 48, 192, 768, 48, 192, 10)]:
        nb = -(-H // R) if R else 1
        ns = nb * (Cx // L.MBHEAD_SLICE)
        nwg = B * ns
        sizes = dict(x=B * H * W * Cin * 2, we=Cx * Cin * 2, be=Cx * 4, wd=9 * Cx * 4, bd=Cx * 4, out=B * H * W * Cx * 2, sums=B * nb * Cx * 4, w1=S * Cx * 4,
                     hp=B * ns * S * 4, tl=nwg * 256 + 256, b1=S * 4, w2t=S * Cx * 4, b2=Cx * 4, sc=B * Cx * 4, hid=B * S * 4, wp=N * Cx * 2, wb=B * N * Cx * 2)
        off, cur = {}, 0
        for k, n in sizes.items():
            off[k] = cur
            cur = (cur + n + 255) // 256 * 256
        ws = torch.empty(cur + 256, dtype=torch.uint8, device="cuda")
        ws.view(torch.float32)[: (cur + 256) // 4].normal_(0, 0.3)
        ws[off["x"]:off["x"] + sizes["x"]].view(torch.bfloat16).normal_(0, 0.5)
        ws[off["we"]:off["we"] + sizes["we"]].view(torch.bfloat16).normal_(0, 0.05)
        ws[off["wp"]:off["wp"] + sizes["wp"]].view(torch.bfloat16).normal_(0, 0.05)

        def ref(o, fld, key):
            r = getattr(o, fld)
            r.base, r.offset = L.BASE_WORKSPACE, off[key]

        op = (L.Op * 2)()
        o = op[0]
        o.kind, o.flags, o.act = L.OP_MBHEAD, 0x1000 | XFLAGS, L.ACT_SILU
        o.in_dtype = o.out_dtype = o.w_dtype = L.BF16
        o.B, o.H, o.W, o.Ho, o.Wo = B, H, W, H, W
        o.Cin, o.Cout, o.ksize, o.stride, o.aux0, o.aux1 = Cin, Cx, 3, 1, S, R
        for fld, key in (("in_", "x"), ("w2", "we"), ("bias2", "be"), ("w", "wd"), ("bias", "bd"), ("out", "out"), ("aux", "sums"), ("in2", "tl"),
                         ("scale", "w1"), ("out2", "hp")):
            ref(o, fld, key)
        o = op[1]
        o.kind, o.flags, o.w_dtype = L.OP_SE, L.FLAG_SE_HPART | L.FLAG_SE_FOLD, L.BF16
        o.B, o.H, o.W, o.Cin, o.Cout, o.Cout_total, o.aux0, o.aux1 = B, H, W, Cx, Cx, N, S, ns
        for fld, key in (("aux", "hp"), ("out", "sc"), ("in2", "hid"), ("w2", "w2t"), ("bias", "b1"), ("bias2", "b2"), ("in_", "wp"), ("out2", "wb")):
            ref(o, fld, key)
        h = C.c_void_p()
        L.check(lib.ftc_plan_create(op, 2, cur + 256, 0, C.byref(h)), "create")
        bases = (C.c_void_p * L.NUM_BASES)(None, ws.data_ptr(), None, None, None, None)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ms = (C.c_float * 2)()
        ts, tse = [], []
        for _ in range(15):
            L.check(lib.ftc_plan_profile(h, bases, st, ms), "profile")
            ts.append(ms[0])
            tse.append(ms[1])
        torch.cuda.synchronize()
        t0 = off["tl"]
        tl = ws[t0:t0 + nwg * 256].view(torch.int64).reshape(nwg, 32).cpu().numpy()
        d = np.diff(tl[:, :5], axis=1)
        span = (tl[:, 4].max() - tl[:, 0].min())
        fl = 2.0 * B * H * W * Cx * (Cin + 9)
        print(f"B{B} {H}x{W} {Cin}->{Cx} R{R}: mbhead {np.median(ts) * 1e3:7.1f} us  {fl / np.median(ts) / 1e9:6.1f} TF  {nwg} workgroups;  cycles (median over workgroups): "
              f"K loop {np.median(d[:, 0]):.0f}  expand epilogue {np.median(d[:, 1]):.0f}  depthwise {np.median(d[:, 2]):.0f}  sums+fc1 {np.median(d[:, 3]):.0f}  "
              f"total {np.median(tl[:, 4] - tl[:, 0]):.0f}  K-loop waits (vmcnt + barrier) {np.median(tl[:, 5]):.0f} of which the first {np.median(tl[:, 6]):.0f};   se(hpart+fold) {np.median(tse) * 1e3:6.1f} us", flush=True)
        nkk = min(24, Cin // 32)
        step = np.median(np.diff(np.concatenate([tl[:, 0:1], tl[:, 8:8 + nkk], tl[:, 1:2]], axis=1), axis=1), axis=0)
        print("      wave 0: start -> top of step 0, then step durations (cycles):", " ".join(f"{v:.0f}" for v in step), flush=True)
        lib.ftc_plan_destroy(h)
