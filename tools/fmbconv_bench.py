#!/usr/bin/env python3
"""Times FTC_OP_FMBCONV (csrc/fused_mbconv.hip) alone on the two shapes of the batch-8 plan: stage 2 (192x192, 64 -> 256 -> 64) and stage 3
(96x96, 96 -> 384 -> 96).  (Round 6 compared four kernel forms through FTC_FMB_WM / FTC_FMB_NBUF switches -- profiles/r06_fmbconv_forms.txt; the library now
instantiates the adopted one only, the switches are gone: git show b32c3b2:findtextcenternet_amd/csrc/fused_mbconv.hip has them.)
    python tools/fmbconv_bench.py [bf16|f16]
Reference (profiles/r05e_bf16_b8_ops.json, the two-launch form inside the plan): stage 2 126 + 65 us, stage 3 75 + 28.5 us."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from findtextcenternet_amd import _lib as L          # noqa: E402
from gpu_harness import Arena, to_dev_bytes          # noqa: E402


def bench(B, H, W, Cin, E, Cout, dt, reps=30):
    lib = L.load()
    g = torch.Generator().manual_seed(1)
    ar = Arena()
    x = torch.randn(B, H, W, Cin, generator=g)
    o_x = ar.put(to_dev_bytes(x, dt))
    o_w1 = ar.put(to_dev_bytes(torch.randn(E, 9, Cin, generator=g) / (9 * Cin) ** 0.5, dt))
    o_b1 = ar.put(torch.randn(E, generator=g) * 0.1)
    o_w2 = ar.put(to_dev_bytes(torch.randn(Cout, E, generator=g) / E ** 0.5, dt))
    o_b2 = ar.put(torch.randn(Cout, generator=g) * 0.1)
    o_res = ar.put(torch.randn(B, H, W, Cout, generator=g))
    o_out, o_out2 = ar.reserve(B * H * W * Cout * 4), ar.reserve(B * H * W * Cout * 2)
    ar.materialize()
    op = (L.Op * 1)()
    f = dict(kind=L.OP_FMBCONV, flags=L.FLAG_RESIDUAL, act=L.ACT_SILU, in_dtype=dt, out_dtype=L.F32, w_dtype=dt, res_dtype=L.F32, B=B, H=H, W=W, Ho=H, Wo=W,
             Cin=Cin, Cin_total=Cin, Cout=Cout, Cout_total=Cout, ksize=3, stride=1, aux1=E)
    for k, v in f.items():
        setattr(op[0], k, int(v))
    for k, v in dict(in_=o_x, in2=o_res, w2=o_w1, bias2=o_b1, w=o_w2, bias=o_b2, out=o_out, out2=o_out2).items():
        r = getattr(op[0], k)
        r.base, r.offset = L.BASE_WORKSPACE, int(v)
    h = C.c_void_p()
    L.check(lib.ftc_plan_create(op, 1, ar.size + 256, 0, C.byref(h)), "ftc_plan_create")
    bases = (C.c_void_p * L.NUM_BASES)(None, ar.buf.data_ptr(), None, None, None, None)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        L.check(lib.ftc_plan_run(h, bases, C.c_void_p(st), 0, -1), "run")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        L.check(lib.ftc_plan_run(h, bases, C.c_void_p(st), 0, -1), "run")
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000)
    lib.ftc_plan_destroy(h)
    ts.sort()
    fl = 2.0 * B * H * W * E * (9 * Cin + Cout)
    med = ts[len(ts) // 2]
    return med, fl / med / 1e6


if __name__ == "__main__":
    dt = L.F16 if len(sys.argv) > 1 and sys.argv[1] == "f16" else L.BF16
    form = "fmbconv_fused (8 waves, one operand buffer)"
    for name, shp in (("stage 2", (8, 192, 192, 64, 256, 64)), ("stage 3", (8, 96, 96, 96, 384, 96))):
        us, tf = bench(*shp, dt)
        print(f"{form}  {name} {shp}: {us:7.1f} us  {tf:6.1f} TFLOP/s", flush=True)
