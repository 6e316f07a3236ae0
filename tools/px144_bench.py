"""A/B of the project-convolution tile configs on the shapes of the bf16 batch-8 plan (one op per plan, HIP events over 200 launches).
usage (GPU box): python tools/px144_bench.py"""
import ctypes as C
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from findtextcenternet_amd import _lib as L
from gpu_harness import Arena, to_dev_bytes, presplit_f16x3

X3 = "--x3" in sys.argv            # the fp16x3 form: fp32 tensors, both operands pre-split

SHAPES = [("stage4", 8, 48, 48, 768, 192), ("stage5", 8, 48, 48, 1536, 256), ("stage6", 8, 24, 24, 3072, 512), ("stage7", 8, 24, 24, 3840, 640), ("stage6_b32", 32, 24, 24, 3072, 512)]
CFGS = [("64x64_dma3", 7 + 48 + 512), ("128x64_dma3", 5 + 48 + 512), ("96x128_dma2", 3 + 32 + 512), ("64x144", 8), ("80x144", 9), ("128x144", 10), ("96x144", 11)]
if X3:
    CFGS = [("64x64_dma2", 7 + 32 + 256), ("128x64_dma2", 5 + 32 + 256), ("64x144", 8), ("80x144", 9), ("128x144", 10), ("96x144", 11)]


def main():
    lib = L.load()
    for name, B, H, W, Cin, Cout in SHAPES:
        g = torch.Generator().manual_seed(1)
        ar = Arena()
        conv = presplit_f16x3 if X3 else (lambda t: to_dev_bytes(t, L.BF16))
        o_in = ar.put(conv(torch.randn(B * H * W, Cin, generator=g)))
        o_w = ar.put(conv(torch.randn(B, Cout, Cin, generator=g) / Cin ** 0.5))
        o_b, o_res = ar.put(torch.randn(Cout, generator=g)), ar.put(torch.randn(B * H * W, Cout, generator=g))
        o_out, o_out2 = ar.reserve(B * H * W * Cout * 4), ar.reserve(B * H * W * Cout * 4)
        o_tl = ar.reserve(2048 * 64)
        ar.materialize()
        line = [f"{name:11s}"]
        for cname, aux0 in CFGS:
            op = (L.Op * 1)()
            dt = L.F32 if X3 else L.BF16
            f = dict(kind=L.OP_CONV, flags=L.FLAG_RESIDUAL | L.FLAG_W_PER_IMAGE | ((L.FLAG_SPLIT16 | L.FLAG_PRESPLIT) if X3 else L.FLAG_KBLOCK32), act=L.ACT_NONE, in_dtype=dt, out_dtype=L.F32, w_dtype=dt,
                     B=B, H=H, W=W, Ho=H, Wo=W, Cin=Cin, Cin_total=Cin, Cout=Cout, Cout_total=Cout, ksize=1, stride=1, res_dtype=L.F32, aux0=aux0)
            for k, v in f.items():
                setattr(op[0], k, int(v))
            for k, v in dict(in_=o_in, in2=o_res, out=o_out, out2=o_out2, w=o_w, bias=o_b).items():
                r = getattr(op[0], k)
                r.base, r.offset = L.BASE_WORKSPACE, int(v)
            h = C.c_void_p()
            rc = lib.ftc_plan_create(op, 1, ar.size + 256, 0, C.byref(h))
            if rc != 0:
                line.append(f"{cname}: illegal")
                continue
            bases = (C.c_void_p * L.NUM_BASES)(None, ar.buf.data_ptr(), None, None, None, None)
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(20):
                lib.ftc_plan_run(h, bases, C.c_void_p(st), 0, -1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                lib.ftc_plan_run(h, bases, C.c_void_p(st), 0, -1)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 200
            tf = (3 if X3 else 1) * 2.0 * B * H * W * Cin * Cout / us * 1e-6
            line.append(f"{cname}: {us:6.1f} us {tf:5.0f} TF")
            lib.ftc_plan_destroy(h)
            if aux0 in (8, 9, 10, 11) and "--timeline" in sys.argv:
                nblk = (Cout // {8: 64, 9: 80, 10: 128, 11: 96}[aux0]) * (B * H * W // 144)
                op[0].flags |= 0x1000
                r = op[0].w2
                r.base, r.offset = L.BASE_WORKSPACE, int(o_tl)
                h = C.c_void_p()
                assert lib.ftc_plan_create(op, 1, ar.size + 256, 0, C.byref(h)) == 0
                for _ in range(3):
                    lib.ftc_plan_run(h, bases, C.c_void_p(st), 0, -1)
                torch.cuda.synchronize()
                tlv = ar.buf[o_tl:o_tl + nblk * 64].view(torch.int64).reshape(nblk, 8).cpu().double()
                ph = [(tlv[:, i + 1] - tlv[:, i]).median().item() for i in range(4)]
                span = (tlv[:, 4].max() - tlv[:, 0].min()).item()
                print(f"    timeline {cname}: first stage {ph[0]:.0f}  K loop {ph[1]:.0f}  exchange {ph[2]:.0f}  finish {ph[3]:.0f}  barrier waits {tlv[:, 5].median().item():.0f}  "
                      f"(cycles)", flush=True)
                lib.ftc_plan_destroy(h)
        print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
