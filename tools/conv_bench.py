#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv kernel on representative detector layers (GPU box).

    python tools/conv_bench.py [--mode bf16|fp32] [--batch 8] [--reps 20] [--only NAME_SUBSTR]

Each shape is run as a single-op plan through the C ABI; time = HIP events (ftc_plan_profile)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from findtextcenternet_amd import _lib as L  # noqa: E402

# name, H, W, Cin, Cout, k, stride, act, residual, se, in_trunk(f32 in bf16 mode), out_trunk
SHAPES = [
    ("stage1 3x3 32->32", 384, 384, 32, 32, 3, 1, 1, True, False, True, True),
    ("stage2 3x3 64->256", 192, 192, 64, 256, 3, 1, 1, False, False, True, False),
    ("stage2 1x1 256->64", 192, 192, 256, 64, 1, 1, 0, True, False, False, True),
    ("stage3 3x3 96->384", 96, 96, 96, 384, 3, 1, 1, False, False, True, False),
    ("stage3 1x1 384->96", 96, 96, 384, 96, 1, 1, 0, True, False, False, True),
    ("stage4 exp 192->768", 48, 48, 192, 768, 1, 1, 1, False, False, True, False),
    ("stage4 proj 768->192", 48, 48, 768, 192, 1, 1, 0, True, True, False, True),
    ("stage5 exp 256->1536", 48, 48, 256, 1536, 1, 1, 1, False, False, True, False),
    ("stage5 proj 1536->256", 48, 48, 1536, 256, 1, 1, 0, True, True, False, True),
    ("stage6 exp 512->3072", 24, 24, 512, 3072, 1, 1, 1, False, False, True, False),
    ("stage6 proj 3072->512", 24, 24, 3072, 512, 1, 1, 0, True, True, False, True),
    ("stage7 proj 3840->640", 24, 24, 3840, 640, 1, 1, 0, True, True, False, True),
    ("fpn L0 1280->1728", 24, 24, 1280, 1728, 3, 1, 2, False, False, False, False),
    ("fpn L1 448->192", 48, 48, 448, 192, 3, 1, 2, False, False, False, False),
    ("fpn L2 288->192", 96, 96, 288, 192, 3, 1, 2, False, False, False, False),
    ("fpn L3 256->192", 192, 192, 256, 192, 3, 1, 2, False, False, False, False),
    ("fpn L3 noact 256->192", 192, 192, 256, 192, 3, 1, 0, False, False, False, False),
    ("fpn L3 silu 256->192", 192, 192, 256, 192, 3, 1, 1, False, False, False, False),
    ("top 192->100", 192, 192, 192, 100, 3, 1, 0, False, False, False, True),
    ("top 192->1", 192, 192, 192, 1, 3, 1, 0, False, False, False, True),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--timeline", action="store_true", help="halo kernel: dump the s_memtime timeline of wave 0 of the first workgroups")
    ap.add_argument("--ablate", type=int, default=0, help="halo kernel ablation: 1 = no DMA in the K loop, 2 = no MFMA/LDS reads (wrong results; timing only)")
    ap.add_argument("--nbuf", type=int, default=0, help="tuning hints (ftc_op.aux0): 4 = no direct-to-LDS, 8 = 3-deep DMA ring, 16 = force direct-to-LDS")
    ap.add_argument("--no-se", action="store_true", help="drop the SE-scale flag (what-if: scale folded into per-image weights)")
    ap.add_argument("--wl1", action="store_true", help="3x3 192-channel layers: the weights-through-L1 kernel (FTC_FLAG_W_FRAG; weight values are random anyway)")
    ap.add_argument("--split16", action="store_true", help="fp32 mode: FTC_FLAG_SPLIT16 (fp16x3 arithmetic)")
    ap.add_argument("--sweep", action="store_true", help="try every tuner candidate for the layer and print the five fastest")
    a = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda")
    cdt = L.F32 if a.mode == "fp32" else L.BF16
    peak = 157.3 if a.mode == "fp32" else 2500.0
    print(f"{'layer':26s} {'M':>8s} {'N':>5s} {'K':>6s} {'us':>9s} {'TFLOP/s':>8s} {'%peak':>6s}  kernel")
    for (name, H, W, Cin, Cout, k, stride, act, res, se, in_tr, out_tr) in SHAPES:
        if a.only and a.only not in name:
            continue
        B = a.batch
        se = se and not a.no_se
        idt = L.F32 if a.mode == "fp32" else L.BF16      # bf16 mode: GEMMs read the bf16 trunk copy
        odt = L.F32 if (a.mode == "fp32" or out_tr) else L.BF16
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        es = lambda d: 4 if d == L.F32 else 2  # noqa: E731
        sizes = {"in": B * H * W * Cin * es(idt), "w": Cout * k * k * Cin * es(cdt), "bias": Cout * 4,
                 "res": B * Ho * Wo * Cout * 4 if (res or a.timeline) else 0, "se": B * Cin * 4 if se else 0, "out": B * Ho * Wo * Cout * es(odt)}
        off, cur = {}, 0
        for key, n in sizes.items():
            off[key] = cur
            cur = (cur + n + 255) // 256 * 256
        ws = torch.empty(cur + 256, dtype=torch.uint8, device=dev)
        # random contents (bench on random data, not zeros: DVFS)
        nfl = (cur + 256) // 4
        ws.view(torch.float32)[:nfl].normal_(0, 0.5)
        if idt == L.BF16:
            ws[off["in"]:off["in"] + sizes["in"]].view(torch.bfloat16).normal_(0, 0.5)
        if cdt == L.BF16:
            ws[off["w"]:off["w"] + sizes["w"]].view(torch.bfloat16).normal_(0, 0.05)
        op = (L.Op * 1)()
        o = op[0]
        o.kind, o.flags, o.act = L.OP_CONV, (L.FLAG_RESIDUAL if res else 0) | (L.FLAG_SE_SCALE if se else 0) | (a.ablate << 8) | (0x1000 if a.timeline else 0) | (L.FLAG_SPLIT16 if a.split16 else 0), act
        o.in_dtype, o.out_dtype, o.w_dtype, o.res_dtype = idt, odt, cdt, L.F32
        o.B, o.H, o.W, o.Ho, o.Wo = B, H, W, Ho, Wo
        o.Cin = o.Cin_total = Cin
        o.Cout = o.Cout_total = Cout
        o.ksize, o.stride = k, stride
        o.aux0 = a.nbuf
        if a.wl1:
            o.flags |= L.FLAG_W_FRAG
            o.aux0 = 193
        for fld, key in (("in_", "in"), ("w", "w"), ("bias", "bias"), ("out", "out")):
            r = getattr(o, fld); r.base, r.offset = L.BASE_WORKSPACE, off[key]
        if res or a.timeline:
            o.in2.base, o.in2.offset = L.BASE_WORKSPACE, off["res"]
        if se:
            o.scale.base, o.scale.offset = L.BASE_WORKSPACE, off["se"]
        bases = (C.c_void_p * L.NUM_BASES)(None, ws.data_ptr(), None, None, None, None)
        ms = (C.c_float * 1)()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        fl = 2.0 * B * Ho * Wo * Cout * Cin * k * k
        if a.sweep:
            from findtextcenternet_amd import tuning as T
            res_ = []
            for aux in T.candidates(o):
                o.aux0 = aux
                h = C.c_void_p()
                if lib.ftc_plan_create(op, 1, cur + 256, 0, C.byref(h)) != 0:
                    continue
                if lib.ftc_plan_run(h, bases, st, 0, -1) == 0:
                    ts = []
                    for _ in range(7):
                        if lib.ftc_plan_profile(h, bases, st, ms) == 0:
                            ts.append(ms[0])
                    if ts:
                        res_.append((float(np.median(ts)), aux))
                lib.ftc_plan_destroy(h)
            res_.sort()
            print(f"{name:26s} M={B * Ho * Wo} N={Cout} K={Cin * k * k} se={se}")
            for t, aux in res_[:5]:
                print(f"    {t * 1e3:8.1f} us {fl / (t * 1e-3) / 1e12:7.1f} TF  {T.describe(aux)}")
            del ws
            continue
        h = C.c_void_p()
        L.check(lib.ftc_plan_create(op, 1, cur + 256, 0, C.byref(h)), "create")
        for _ in range(3):
            lib.ftc_plan_run(h, bases, st, 0, -1)
        ts = []
        for _ in range(a.reps):
            L.check(lib.ftc_plan_profile(h, bases, st, ms), "profile")
            ts.append(ms[0])
        t = float(np.median(ts))
        buf = C.create_string_buffer(128)
        lib.ftc_op_kernel_label(C.byref(o), buf, 128)
        tf = fl / (t * 1e-3) / 1e12
        print(f"{name:26s} {B * Ho * Wo:8d} {Cout:5d} {Cin * k * k:6d} {t * 1e3:9.1f} {tf:8.1f} {100 * tf / peak:6.1f}  {buf.value.decode()}")
        if a.timeline:
            torch.cuda.synchronize()
            tl = ws[off["res"]:off["res"] + 512 * 64 * 8].view(torch.int64).reshape(512, 64).cpu().numpy()
            import numpy as _np
            for blk in (0, 1, 100, 255, 256, 300, 511):
                r = tl[blk]
                steps = _np.diff(r[2:42])
                print(f"  wg {blk:3d}: setup+prologue issue {r[1]-r[0]:6d}  first-step wait {r[3]-r[2]:6d}  steps 1..35 mean {steps[1:35].mean():7.0f} (min {steps[1:35].min()}, max {steps[1:35].max()})  "
                      f"loop {r[42]-r[2]:7d}  epilogue {r[43]-r[42]:6d}  total {r[43]-r[0]:7d}  start_rel {r[0]-tl[0][0]:8d}")
        lib.ftc_plan_destroy(h)
        del ws


if __name__ == "__main__":
    main()
