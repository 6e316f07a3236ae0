#!/usr/bin/env python3
"""Single-op timing + phase timeline (s_memtime of wave 0) of the fused MBConv head kernel (csrc/mbconv_fused.hip) on the stage-6/7 shapes.
    python tools/mbfused_bench.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from findtextcenternet_amd import _lib as L  # noqa: E402

lib = L.load()
for (B, H, W, Cin, Cx) in [(8, 24, 24, 512, 3072), (8, 24, 24, 640, 3840), (8, 48, 48, 256, 1536)]:
    tymax = 352 // W - 2
    P = -(-H // tymax)
    nwg = B * P * (Cx // 128)
    sizes = dict(x=B * H * W * Cin * 2, we=Cx * Cin * 2, be=Cx * 4, wd=9 * Cx * 4, bd=Cx * 4, out=B * H * W * Cx * 2, part=B * P * Cx * 4, tl=nwg * 64)
    off, cur = {}, 0
    for k, n in sizes.items():
        off[k] = cur
        cur = (cur + n + 255) // 256 * 256
    ws = torch.empty(cur + 256, dtype=torch.uint8, device="cuda")
    ws.view(torch.float32)[: (cur + 256) // 4].normal_(0, 0.3)
    ws[off["x"]:off["x"] + sizes["x"]].view(torch.bfloat16).normal_(0, 0.5)
    ws[off["we"]:off["we"] + sizes["we"]].view(torch.bfloat16).normal_(0, 0.05)
    op = (L.Op * 1)()
    o = op[0]
    o.kind, o.flags, o.act = L.OP_DWCONV, L.FLAG_EXPAND_IN | 0x1000, L.ACT_SILU
    o.in_dtype = o.out_dtype = o.w_dtype = L.BF16
    o.B, o.H, o.W, o.Ho, o.Wo = B, H, W, H, W
    o.Cin = o.Cout = Cx
    o.Cin_total = Cin
    o.ksize, o.stride, o.aux0 = 3, 1, P
    for fld, key in (("in_", "x"), ("w2", "we"), ("bias2", "be"), ("w", "wd"), ("bias", "bd"), ("out", "out"), ("aux", "part"), ("in2", "tl")):
        r = getattr(o, fld)
        r.base, r.offset = L.BASE_WORKSPACE, off[key]
    h = C.c_void_p()
    L.check(lib.ftc_plan_create(op, 1, cur + 256, 0, C.byref(h)), "create")
    bases = (C.c_void_p * L.NUM_BASES)(None, ws.data_ptr(), None, None, None, None)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = (C.c_float * 1)()
    ts = []
    for _ in range(12):
        L.check(lib.ftc_plan_profile(h, bases, st, ms), "profile")
        ts.append(ms[0])
    torch.cuda.synchronize()
    tl = ws[off["tl"]:off["tl"] + nwg * 64].view(torch.int64).reshape(nwg, 8).cpu().numpy()
    d = np.diff(tl[:, :5], axis=1)
    fl = 2.0 * B * H * W * Cx * (Cin + 9)
    print(f"B{B} {H}x{W} {Cin}->{Cx}: {np.median(ts) * 1e3:7.1f} us  {fl / np.median(ts) / 1e9:6.1f} TF  {nwg} workgroups;  cycles (median over workgroups): "
          f"K loop {np.median(d[:, 0]):.0f}  expand epilogue {np.median(d[:, 1]):.0f}  depthwise {np.median(d[:, 2]):.0f}  sums {np.median(d[:, 3]):.0f}  total {np.median(tl[:, 4] - tl[:, 0]):.0f}")
    lib.ftc_plan_destroy(h)
