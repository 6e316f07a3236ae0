#!/usr/bin/env python3
"""Micro-benchmark of the depthwise 3x3 kernels on the detector's MBConv shapes (GPU box).

    python tools/dw_bench.py [--batch 8] [--reps 20]

Each shape runs as a single-op plan through the C ABI (HIP-event time from ftc_plan_profile); `old` = the
8-channel-per-lane kernel (op flag 0x100), `new` = the default selection."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from findtextcenternet_amd import _lib as L  # noqa: E402

SHAPES = [("stage4 96x96x768", 96, 96, 768), ("stage5 48x48x1536", 48, 48, 1536), ("stage6 24x24x3072", 24, 24, 3072),
          ("stage7 24x24x3840", 24, 24, 3840)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda")
    B = a.batch
    for name, H, W, Cc in SHAPES:
        P = ((H + 7) // 8) * ((W + 7) // 8)
        n = B * H * W * Cc
        x = (torch.randn(n, device=dev) * 0.5).to(torch.bfloat16)
        sizes = [n * 2, 9 * Cc * 4, Cc * 4, n * 2, B * P * Cc * 4]
        off, cur = [], 0
        for s in sizes:
            off.append(cur)
            cur = (cur + s + 255) // 256 * 256
        ws = torch.empty(cur, dtype=torch.uint8, device=dev)
        ws[:n * 2].view(torch.bfloat16).copy_(x)
        ws[off[1]:off[1] + 9 * Cc * 4].view(torch.float32).normal_(0, 0.3)
        ws[off[2]:off[2] + Cc * 4].view(torch.float32).normal_(0, 0.1)
        row = f"{name:22s} {2 * n * 2 / 1e6:7.1f} MB "
        outs = []
        for label, flags in (("old", 0x100), ("new", 0)):
            op = (L.Op * 1)()
            o = op[0]
            o.kind, o.flags, o.act = L.OP_DWCONV, flags, L.ACT_SILU
            o.in_dtype = o.out_dtype = L.BF16
            o.B, o.H, o.W, o.Ho, o.Wo = B, H, W, H, W
            o.Cin = o.Cin_total = o.Cout = o.Cout_total = Cc
            o.ksize, o.stride, o.aux0 = 3, 1, P
            for fld, k in (("in_", 0), ("w", 1), ("bias", 2), ("out", 3), ("aux", 4)):
                r = getattr(o, fld); r.base, r.offset = L.BASE_WORKSPACE, off[k]
            h = C.c_void_p()
            L.check(lib.ftc_plan_create(op, 1, cur, 0, C.byref(h)), "create")
            bases = (C.c_void_p * L.NUM_BASES)(None, ws.data_ptr(), None, None, None, None)
            ms = (C.c_float * 1)()
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                lib.ftc_plan_run(h, bases, st, 0, -1)
            ts = []
            for _ in range(a.reps):
                L.check(lib.ftc_plan_profile(h, bases, st, ms), "profile")
                ts.append(ms[0])
            t = float(np.median(ts))
            torch.cuda.synchronize()
            outs.append((ws[off[3]:off[3] + n * 2].view(torch.bfloat16).float().clone(), ws[off[4]:off[4] + B * P * Cc * 4].view(torch.float32).clone()))
            row += f" {label} {t * 1e3:7.1f} us {2 * n * 2 / (t * 1e-3) / 1e9:7.0f} GB/s "
            lib.ftc_plan_destroy(h)
        d = float((outs[0][0] - outs[1][0]).abs().max())
        dp = float((outs[0][1] - outs[1][1]).abs().max() / outs[0][1].abs().max())
        print(row + f" | max|old-new| {d:.2e}  partial rel {dp:.1e}", flush=True)


if __name__ == "__main__":
    main()
