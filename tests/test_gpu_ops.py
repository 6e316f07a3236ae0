"""-m gpu: every HIP kernel, called through the C ABI (single-op plans), against the same op
computed by PyTorch on the CPU in fp32 -- the floating-point reference for one kernel; the
end-to-end oracle/golden comparison is in test_gpu_detector.py.

Tolerances: fp32 kernels 2e-4 * max|ref| (summation order only: exact-f32 MFMA vs MKL-DNN);
bf16 kernels are compared with the CPU result on bf16-ROUNDED operands (so only accumulation
order / output rounding differ): 1.5e-2 * max|ref|.
"""
import ctypes as C
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from findtextcenternet_amd import _lib as L
from gpu_harness import round16, Arena, bf16_round, presplit_f16x3, run_op, tdtype, to_dev_bytes

pytestmark = pytest.mark.gpu

ACT = {L.ACT_NONE: lambda v: v, L.ACT_SILU: F.silu, L.ACT_GELU: F.gelu}


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _log(msg):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/test_ops.log", "a") as f:
        f.write(msg + "\n")


# (name, B,H,W, Cin,CinT,cin_off, Cout,CoutT,cout_off, k, stride, act, residual, se)
CONV_CASES = [
    ("pw_128tile", 2, 24, 24, 64, 64, 0, 256, 256, 0, 1, 1, L.ACT_SILU, False, False),
    ("pw_project_se_res", 2, 12, 12, 384, 384, 0, 64, 64, 0, 1, 1, L.ACT_NONE, True, True),
    ("pw_bigtile", 4, 48, 48, 96, 96, 0, 384, 384, 0, 1, 1, L.ACT_SILU, False, False),
    ("pw_n640", 1, 24, 24, 128, 128, 0, 640, 640, 0, 1, 1, L.ACT_NONE, True, False),
    ("c3_s1_32", 1, 40, 36, 32, 32, 0, 32, 32, 0, 3, 1, L.ACT_SILU, True, False),
    ("c3_s2_32_128", 2, 32, 32, 32, 32, 0, 128, 128, 0, 3, 2, L.ACT_SILU, False, False),
    ("c3_s1_64_256", 1, 24, 28, 64, 64, 0, 256, 256, 0, 3, 1, L.ACT_SILU, False, False),
    ("c3_96", 1, 20, 20, 96, 96, 0, 96, 96, 0, 3, 1, L.ACT_NONE, True, False),
    ("fpn_192_gelu", 1, 24, 24, 288, 288, 0, 192, 192, 0, 3, 1, L.ACT_GELU, False, False),
    ("fpn_slice_in", 1, 16, 16, 64, 256, 192, 192, 192, 0, 3, 1, L.ACT_GELU, False, False),
    ("top_1ch_into10", 2, 16, 24, 192, 192, 0, 1, 10, 4, 3, 1, L.ACT_NONE, False, False),
    ("top_2ch_into10", 1, 16, 16, 192, 192, 0, 2, 10, 2, 3, 1, L.ACT_NONE, False, False),
    ("top_100ch", 1, 24, 24, 192, 192, 0, 100, 100, 0, 3, 1, L.ACT_NONE, False, False),
    ("odd_hw_s2", 1, 15, 17, 64, 64, 0, 64, 64, 0, 3, 2, L.ACT_SILU, False, False),
    ("cin_not_mult32", 1, 12, 12, 24, 24, 0, 48, 48, 0, 3, 1, L.ACT_SILU, False, False),
    ("k_deep", 1, 8, 8, 1280, 1280, 0, 192, 192, 0, 3, 1, L.ACT_GELU, False, False),
]
# (mode name, w_dtype, in_dtype, out_dtype)
CONV_MODES = [("f32", L.F32, L.F32, L.F32), ("bf16", L.BF16, L.BF16, L.BF16), ("bf16_f32in", L.BF16, L.F32, L.BF16),
              ("bf16_f32out", L.BF16, L.BF16, L.F32), ("bf16_f32io", L.BF16, L.F32, L.F32),
              ("f16", L.F16, L.F16, L.F16), ("f16_f32in", L.F16, L.F32, L.F16), ("f16_f32out", L.F16, L.F16, L.F32), ("f16_f32io", L.F16, L.F32, L.F32),
              # fp32 tensors and weights, products as three fp16 MFMAs of hi / lo split operands (FTC_FLAG_SPLIT16): held to the fp32 tolerance
              ("f32x3", L.F32, L.F32, L.F32)]
TOL16 = {L.BF16: 1.5e-2, L.F16: 2.5e-3}          # relative error of a 16-bit-operand conv against the fp32 reference on the same rounded operands


HALO_CASES = [c for c in CONV_CASES if c[10] == 3 and c[11] == 1] + [
    ("halo_2tiles_edge", 2, 40, 24, 64, 64, 0, 192, 192, 0, 3, 1, L.ACT_GELU, False, False),
    ("halo_288", 1, 32, 32, 288, 288, 0, 192, 192, 0, 3, 1, L.ACT_GELU, False, False),
]


@pytest.mark.parametrize("tile", [65, 66, 68], ids=["halo192", "halo128", "halo64"])
@pytest.mark.parametrize("mode", [CONV_MODES[0], CONV_MODES[1], CONV_MODES[3], CONV_MODES[5], CONV_MODES[7], CONV_MODES[9]],
                         ids=["f32", "bf16", "bf16_f32out", "f16", "f16_f32out", "f32x3"])
@pytest.mark.parametrize("case", HALO_CASES, ids=[c[0] for c in HALO_CASES])
def test_conv_halo_kernel(case, mode, tile):
    """The LDS-halo 3x3 kernel (ftc_op.aux0 bit 6) on every stride-1 3x3 case, all three channel tiles."""
    if case[4] % (8 if mode[1] == L.F32 else 32):
        pytest.skip("halo kernel needs whole 32-element channel blocks")
    if mode[1] == L.F32 and case[4] % 32:
        pytest.skip("fp32 halo kernel needs Cin % 32 == 0")
    _run_conv_case(case, mode, aux0=tile)


SPLITK_CASES = [
    ("sk_proj_se_res", 2, 24, 24, 1024, 1024, 0, 128, 128, 0, 1, 1, L.ACT_NONE, True, True),
    ("sk_pw_silu", 1, 24, 24, 512, 512, 0, 192, 192, 0, 1, 1, L.ACT_SILU, False, False),
    ("sk_3x3", 1, 12, 12, 128, 128, 0, 64, 64, 0, 3, 1, L.ACT_GELU, False, False),
    ("sk_odd_n", 1, 16, 16, 768, 768, 0, 100, 100, 0, 1, 1, L.ACT_NONE, False, False),
]


@pytest.mark.parametrize("aux0", [1559, 2583, 1557, 1556 + 1024, 7 + 16 + 768 + 1024], ids=["64x64_sk2", "64x64_sk4", "128x64_sk2", "64x128_sk4", "64x64_bk128_sk2"])
@pytest.mark.parametrize("mode", [CONV_MODES[1], CONV_MODES[3]], ids=["bf16", "bf16_f32out"])
@pytest.mark.parametrize("case", SPLITK_CASES, ids=[c[0] for c in SPLITK_CASES])
def test_conv_intra_workgroup_split_k(case, mode, aux0):
    cin_k = case[4] * case[10] * case[10]
    bk = 128 if (aux0 >> 8) & 3 == 3 else 64
    kg = 2 if (aux0 >> 10) & 3 == 1 else 4
    if case[4] % bk or (cin_k // bk) % kg or (cin_k // bk) // kg < 2:
        pytest.skip("K loop does not split evenly")
    _run_conv_case(case, mode, aux0=aux0)


@pytest.mark.parametrize("mode", CONV_MODES, ids=[m[0] for m in CONV_MODES])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv(case, mode):
    _run_conv_case(case, mode, aux0=0)


def _run_conv_case(case, mode, aux0):
    name, B, H, W, Cin, CinT, cin_off, Cout, CoutT, cout_off, k, stride, act, residual, se = case
    mname, wdt, idt, odt = mode
    if wdt != L.F32 and Cin % 8:
        pytest.skip("16-bit operands need Cin % 8 == 0")
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    x_full = torch.randn(B, H, W, CinT, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.3
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Wo, Cout, generator=g) if residual else None
    sc = torch.rand(B, Cin, generator=g) + 0.25 if se else None
    x_full = round16(x_full, idt)
    wq = round16(w, wdt)
    x = x_full[..., cin_off:cin_off + Cin]
    xin = x * sc[:, None, None, :] if se else x
    if (se and wdt != L.F32) or (wdt != L.F32 and idt == L.F32):
        xin = round16(xin, wdt)                    # the kernel narrows the (scaled) activation to the compute type
    ref = F.conv2d(xin.permute(0, 3, 1, 2), wq, None, stride, pad).permute(0, 2, 3, 1) + bias
    ref = ACT[act](ref)
    if residual:
        ref = ref + res
    ar = Arena()
    o_in = ar.put(to_dev_bytes(x_full, idt))
    wk = w.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin)
    o_w = ar.put(presplit_f16x3(wk) if mname == "f32x3" else to_dev_bytes(wk, wdt))      # fp16x3: the weights come pre-split
    o_b = ar.put(bias)
    o_res = ar.put(res) if residual else None
    o_sc = ar.put(sc) if se else None
    esz = 4 if odt == L.F32 else 2
    o_out = ar.reserve(B * Ho * Wo * CoutT * esz)
    ar.materialize()
    run_op(dict(kind=L.OP_CONV, flags=(L.FLAG_RESIDUAL if residual else 0) | (L.FLAG_SE_SCALE if se else 0) | (L.FLAG_SPLIT16 if mname == "f32x3" else 0),
                act=act, in_dtype=idt, out_dtype=odt, w_dtype=wdt, B=B, H=H, W=W, Ho=Ho, Wo=Wo, Cin=Cin, Cin_total=CinT, cin_off=cin_off,
                Cout=Cout, Cout_total=CoutT, cout_off=cout_off, ksize=k, stride=stride, res_dtype=L.F32, aux0=aux0,
                in_=o_in, in2=o_res, out=o_out, w=o_w, bias=o_b, scale=o_sc), ar)
    full = ar.read(o_out, (B, Ho, Wo, CoutT), tdtype(odt))
    out = full[..., cout_off:cout_off + Cout].float()
    err = _rel(out, ref)
    _log(f"conv {name:18s} {mname:12s} aux0={aux0} rel_err {err:.3e}")
    tol = 2e-4 if wdt == L.F32 else TOL16[wdt]
    if mname == "f32x3":
        tol = 2e-5                                  # 22-bit operands: measured 1e-6 .. 4e-6 (the exact-fp32 kernel: 1e-7 .. 1e-6)
    assert err < tol, (name, mname, err)
    if CoutT != Cout:      # untouched channels keep the 0xCD fill: the kernel wrote only its slice
        raw = ar.buf[o_out:o_out + B * Ho * Wo * CoutT * esz].cpu().view(B * Ho * Wo, CoutT * esz)
        keep = torch.ones(CoutT * esz, dtype=torch.bool)
        keep[cout_off * esz:(cout_off + Cout) * esz] = False
        assert (raw[:, keep] == 0xCD).all()


@pytest.mark.parametrize("aux0", [0, 5 + 32 + 512, 7 + 48 + 512, 7 + 16 + 512, 65], ids=["default", "128x64_dma2", "64x64_dma3", "64x64_reg", "halo"])
@pytest.mark.parametrize("odt,dt", [(L.F32, L.BF16), (L.BF16, L.BF16), (L.F32, L.F16), (L.F16, L.F16)], ids=["bf16_f32out", "bf16", "f16_f32out", "f16"])
def test_se_fold_then_per_image_weight_conv(odt, dt, aux0):
    """bf16 mode of an MBConv tail: the SE op writes W_b = bf16(W * scale[b]) (FTC_FLAG_SE_FOLD) and the project
    convolution runs with one weight set per image (FTC_FLAG_W_PER_IMAGE) -- against project(x * scale) in fp32."""
    g = torch.Generator().manual_seed(23)
    r16 = lambda t: round16(t, dt)
    B, H, W, Cc, N, S, P = 3, 8, 16, 256, 192, 16, 2
    k = 3 if aux0 == 65 else 1
    x = r16(torch.randn(B, H, W, Cc, generator=g))
    part = torch.randn(B, P, Cc, generator=g) * 30
    w1 = torch.randn(S, Cc, generator=g) / Cc ** 0.5
    b1 = torch.randn(S, generator=g) * 0.3
    w2 = torch.randn(Cc, S, generator=g) / S ** 0.5
    b2 = torch.randn(Cc, generator=g) * 0.3
    wp = r16(torch.randn(N, Cc, k, k, generator=g) / (Cc * k * k) ** 0.5)
    bias = torch.randn(N, generator=g) * 0.2
    res = torch.randn(B, H, W, N, generator=g)
    mean = part.sum(1) / (H * W)
    sc = torch.sigmoid(F.silu(mean @ w1.t() + b1) @ w2.t() + b2)                          # [B, C]
    ref = torch.stack([F.conv2d((x[b] * sc[b]).permute(2, 0, 1)[None], wp, bias, 1, (k - 1) // 2)[0].permute(1, 2, 0) for b in range(B)]) + res
    ar = Arena()
    o_x = ar.put(to_dev_bytes(x, dt))
    o_part = ar.put(part)
    o_w1, o_b1, o_w2t, o_b2 = ar.put(w1), ar.put(b1), ar.put(w2.t().contiguous()), ar.put(b2)
    o_wp = ar.put(to_dev_bytes(wp.permute(0, 2, 3, 1).reshape(N, k * k * Cc), dt))
    o_bias, o_res = ar.put(bias), ar.put(res)
    o_scale, o_hid = ar.reserve(B * Cc * 4), ar.reserve(B * S * 4)
    o_wb = ar.reserve(B * N * k * k * Cc * 2)
    esz = 4 if odt == L.F32 else 2
    o_out = ar.reserve(B * H * W * N * esz)
    ar.materialize()
    # the fold treats the weight matrix as [rows][C]: for the 3x3 variant of this test rows = N*9 (K-major layout)
    run_op(dict(kind=L.OP_SE, flags=L.FLAG_SE_FOLD, w_dtype=dt, B=B, H=H, W=W, Cin=Cc, Cout=Cc, Cout_total=N * k * k, aux0=S, aux1=P,
                aux=o_part, out=o_scale, in2=o_hid, w=o_w1, w2=o_w2t, bias=o_b1, bias2=o_b2, in_=o_wp, out2=o_wb), ar)
    got_sc = ar.read(o_scale, (B, Cc), torch.float32)
    assert float((got_sc - sc).abs().max()) < 2e-6
    wb = ar.read(o_wb, (B, N, k * k, Cc), tdtype(dt)).float()
    want = r16(wp.permute(0, 2, 3, 1).reshape(1, N, k * k, Cc) * got_sc[:, None, None, :])
    assert float((wb - want).abs().max()) <= float(want.abs().max()) * (2 ** -8 if dt == L.BF16 else 2 ** -11)      # same product, at most one ulp of the 16-bit type apart
    run_op(dict(kind=L.OP_CONV, flags=L.FLAG_RESIDUAL | L.FLAG_W_PER_IMAGE, act=L.ACT_NONE, in_dtype=dt, out_dtype=odt, w_dtype=dt,
                B=B, H=H, W=W, Ho=H, Wo=W, Cin=Cc, Cin_total=Cc, Cout=N, Cout_total=N, ksize=k, stride=1, res_dtype=L.F32, aux0=aux0,
                in_=o_x, in2=o_res, out=o_out, w=o_wb, bias=o_bias), ar)
    out = ar.read(o_out, (B, H, W, N), tdtype(odt)).float()
    err = _rel(out, ref)
    _log(f"se_fold+per-image conv odt={odt} aux0={aux0} rel_err {err:.3e}")
    assert err < TOL16[dt]


@pytest.mark.parametrize("variant", ["plain", "res_copy", "res_kblock", "per_image", "slices"])
@pytest.mark.parametrize("dt", [L.BF16, L.F16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(3, 12, 12, 256, 192, 8), (8, 24, 24, 1536, 256, 8), (2, 24, 24, 3072, 512, 8), (1, 12, 24, 64, 64, 8), (5, 24, 24, 320, 640, 8),
                                   (5, 24, 24, 320, 640, 9), (3, 12, 12, 192, 80, 9), (2, 48, 48, 1536, 256, 10), (3, 12, 12, 64, 384, 10),
                                   (2, 48, 48, 768, 192, 11), (3, 12, 12, 128, 96, 11)],
                         ids=lambda s: "x".join(map(str, s[:5])) + "-" + {8: "64x144", 9: "80x144", 10: "128x144", 11: "96x144"}[s[5]])
def test_conv1x1_px144_tile(shape, dt, variant):
    """The (64 | 80 | 128)-channel x 144-pixel 1x1 kernel (aux0 low nibble 8 | 9 | 10: csrc/conv1x1_px144.hip -- the MBConv project convolutions) against the fp32
    convolution of the rounded operands, and against the 64x64 tile config within fp32 summation-order noise: residual, the 16-bit trunk copy
    (NHWC / 32-channel planes), per-image weight sets, channel slices of wider tensors."""
    B, H, W, Cin, Cout, px_aux0 = shape
    if variant == "res_kblock" and Cout % 32:
        pytest.skip("32-channel planes need Cout % 32 == 0")
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout)
    r16 = lambda t: round16(t, dt)
    sl = variant == "slices"
    CinT, cin_off = (Cin + 64, 32) if sl else (Cin, 0)
    CoutT, cout_off = (Cout + 24, 16) if sl else (Cout, 0)
    per = variant == "per_image"
    res_on = variant in ("res_copy", "res_kblock", "per_image")
    copy = variant in ("res_copy", "res_kblock")
    xw = r16(torch.randn(B, H, W, CinT, generator=g))
    x = xw[..., cin_off:cin_off + Cin]
    w = r16(torch.randn(B if per else 1, Cout, Cin, generator=g) / Cin ** 0.5)
    bias = torch.randn(Cout, generator=g) * 0.3
    res = torch.randn(B, H, W, Cout, generator=g)
    ref = torch.einsum("bhwk,bnk->bhwn", x.double(), w.expand(B, -1, -1).double()).float() + bias
    if res_on:
        ref = ref + res
    outs = []
    for aux0 in (px_aux0, 7 + 48 + 512) if not (per and H * W % 64) else (px_aux0, px_aux0):      # (per-image sets: the 64-pixel tiles must divide the image)
        ar = Arena()
        o_in, o_w, o_b, o_res = ar.put(to_dev_bytes(xw, dt)), ar.put(to_dev_bytes(w, dt)), ar.put(bias), ar.put(res)
        o_out, o_out2 = ar.reserve(B * H * W * CoutT * 4), ar.reserve(B * H * W * Cout * 2)
        ar.materialize()
        ar.buf[o_out:o_out + B * H * W * CoutT * 4] = 0xCD
        flags = (L.FLAG_RESIDUAL if res_on else 0) | (L.FLAG_W_PER_IMAGE if per else 0) | (L.FLAG_KBLOCK32 if variant == "res_kblock" else 0)
        run_op(dict(kind=L.OP_CONV, flags=flags, act=L.ACT_NONE, in_dtype=dt, out_dtype=L.F32, w_dtype=dt, B=B, H=H, W=W, Ho=H, Wo=W, Cin=Cin, Cin_total=CinT,
                    cin_off=cin_off, Cout=Cout, Cout_total=CoutT, cout_off=cout_off, ksize=1, stride=1, res_dtype=L.F32, aux0=aux0,
                    in_=o_in, in2=o_res if res_on else None, out=o_out, out2=o_out2 if copy else None, w=o_w, bias=o_b), ar)
        full = ar.read(o_out, (B, H, W, CoutT), torch.float32)
        out = full[..., cout_off:cout_off + Cout]
        if sl:
            raw = ar.buf[o_out:o_out + B * H * W * CoutT * 4].cpu().view(B * H * W, CoutT * 4)
            keep = torch.ones(CoutT * 4, dtype=torch.bool)
            keep[cout_off * 4:(cout_off + Cout) * 4] = False
            assert (raw[:, keep] == 0xCD).all()
        if copy:
            if variant == "res_kblock":
                out2 = ar.read(o_out2, (B, Cout // 32, H * W, 32), tdtype(dt)).permute(0, 2, 1, 3).reshape(B, H, W, Cout)
            else:
                out2 = ar.read(o_out2, (B, H, W, Cout), tdtype(dt))
            assert torch.equal(out2, out.to(tdtype(dt)))
        outs.append(out)
    e0, e1, d = _rel(outs[0], ref), _rel(outs[1], ref), _rel(outs[0], outs[1])
    _log(f"conv1x1_px144 {shape} dt={dt} {variant}: rel_err {e0:.2e} (64x64: {e1:.2e}), between the two {d:.2e}")
    assert e0 < 1e-5 and d < 1e-5          # operands are exactly representable: only the fp32 summation order differs


@pytest.mark.parametrize("variant", ["plain", "res_copy", "per_image"])
@pytest.mark.parametrize("shape", [(3, 12, 12, 256, 192, 8), (2, 24, 24, 3072, 512, 8), (1, 12, 24, 64, 64, 8), (5, 24, 24, 192, 640, 9), (2, 48, 48, 1536, 256, 10), (3, 12, 12, 128, 384, 11)],
                         ids=lambda s: "x".join(map(str, s[:5])) + "-" + {8: "64x144", 9: "80x144", 10: "128x144", 11: "96x144"}[s[5]])
def test_conv1x1_px144_tile_fp16x3(shape, variant):
    """The 144-pixel 1x1 kernel in the fp16x3 plan's form: fp32 tensors, BOTH operands stored pre-split (FTC_FLAG_SPLIT16 | FTC_FLAG_PRESPLIT), 256-byte operand rows,
    three fp16 MFMAs per product; the trunk copy (out2) is the pre-split form of the fp32 output.  Held to the fp32 tolerance of the other fp16x3 kernels
    and compared with the 64x64 tile config of the generic kernel."""
    B, H, W, Cin, Cout, px_aux0 = shape
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + 1)
    per = variant == "per_image"
    res_on = variant in ("res_copy", "per_image")
    copy = variant == "res_copy"
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(B if per else 1, Cout, Cin, generator=g) / Cin ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.3
    res = torch.randn(B, H, W, Cout, generator=g)
    ref = torch.einsum("bhwk,bnk->bhwn", x.double(), w.expand(B, -1, -1).double()).float() + bias
    if res_on:
        ref = ref + res
    outs = []
    for aux0 in (px_aux0, 7 + 48 + 256) if not (per and H * W % 64) else (px_aux0, px_aux0):
        ar = Arena()
        o_in, o_w, o_b, o_res = ar.put(presplit_f16x3(x)), ar.put(presplit_f16x3(w)), ar.put(bias), ar.put(res)
        o_out, o_out2 = ar.reserve(B * H * W * Cout * 4), ar.reserve(B * H * W * Cout * 4)
        ar.materialize()
        flags = L.FLAG_SPLIT16 | L.FLAG_PRESPLIT | (L.FLAG_RESIDUAL if res_on else 0) | (L.FLAG_W_PER_IMAGE if per else 0)
        run_op(dict(kind=L.OP_CONV, flags=flags, act=L.ACT_NONE, in_dtype=L.F32, out_dtype=L.F32, w_dtype=L.F32, B=B, H=H, W=W, Ho=H, Wo=W, Cin=Cin, Cin_total=Cin,
                    Cout=Cout, Cout_total=Cout, ksize=1, stride=1, res_dtype=L.F32, aux0=aux0,
                    in_=o_in, in2=o_res if res_on else None, out=o_out, out2=o_out2 if copy else None, w=o_w, bias=o_b), ar)
        out = ar.read(o_out, (B, H, W, Cout), torch.float32)
        if copy:
            raw = ar.buf[o_out2:o_out2 + B * H * W * Cout * 4].cpu()
            assert torch.equal(raw, presplit_f16x3(out).cpu())               # the copy is exactly the pre-split form of the fp32 value
        outs.append(out)
    e0, e1, d = _rel(outs[0], ref), _rel(outs[1], ref), _rel(outs[0], outs[1])
    _log(f"conv1x1_px144 fp16x3 {shape} {variant}: rel_err {e0:.2e} (64x64: {e1:.2e}), between the two {d:.2e}")
    assert e0 < 1e-5 and d < 1e-5


@pytest.mark.parametrize("aux0", [0, 4 + 32 + 512, 2 + 16 + 512, 65, 68], ids=["default", "64x128_dma2", "128x128_reg", "halo192", "halo64"])
@pytest.mark.parametrize("mode", [CONV_MODES[0], CONV_MODES[1]], ids=["f32", "bf16"])
@pytest.mark.parametrize("out_slice", [False, True], ids=["stacked", "out_slice"])
def test_grouped_conv_runs_independent_instances(out_slice, mode, aux0):
    """ftc_op.groups: G convolutions with their own inputs / weights / biases in one launch (the nine FPN heads)."""
    _, wdt, idt, odt = mode
    if out_slice:
        odt = L.F32
    g = torch.Generator().manual_seed(41)
    G, B, H, W, Cin = 3, 2, 20, 12, 64
    Cout = 2 if out_slice else 192
    CoutT, coff = (10, 3) if out_slice else (Cout, 0)
    x = torch.randn(G, B, H, W, Cin, generator=g)
    w = torch.randn(G, Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(G, Cout, generator=g) * 0.3
    if idt == L.BF16:
        x = bf16_round(x)
    wq = bf16_round(w) if wdt == L.BF16 else w
    ref = torch.stack([F.gelu(F.conv2d(x[i].permute(0, 3, 1, 2), wq[i], bias[i], 1, 1)).permute(0, 2, 3, 1) for i in range(G)])
    ar = Arena()
    o_in = ar.put(to_dev_bytes(x, idt))
    o_w = ar.put(to_dev_bytes(w.permute(0, 1, 3, 4, 2).reshape(G, Cout, 9, Cin), wdt))
    o_b = ar.put(bias)
    esz = 4 if odt == L.F32 else 2
    o_out = ar.reserve((1 if out_slice else G) * B * H * W * CoutT * esz)
    ar.materialize()
    run_op(dict(kind=L.OP_CONV, flags=L.FLAG_GROUP_OUT_SLICE if out_slice else 0, act=L.ACT_GELU, in_dtype=idt, out_dtype=odt, w_dtype=wdt,
                B=B, H=H, W=W, Ho=H, Wo=W, Cin=Cin, Cin_total=Cin, Cout=Cout, Cout_total=CoutT, cout_off=coff, ksize=3, stride=1,
                aux0=aux0, groups=G, in_=o_in, out=o_out, w=o_w, bias=o_b), ar)
    if out_slice:
        full = ar.read(o_out, (B, H, W, CoutT), tdtype(odt)).float()
        out = torch.stack([full[..., coff + i * Cout:coff + (i + 1) * Cout] for i in range(G)])
    else:
        out = ar.read(o_out, (G, B, H, W, Cout), tdtype(odt)).float()
    err = _rel(out, ref)
    _log(f"grouped conv out_slice={out_slice} {mode[0]} aux0={aux0} rel_err {err:.3e}")
    assert err < (2e-4 if wdt == L.F32 else 1.5e-2)


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
@pytest.mark.parametrize("in_slice", [True, False], ids=["in_slice", "stacked"])
def test_grouped_upcat(in_slice, dt):
    g = torch.Generator().manual_seed(43)
    G, B, Hi, Wi, Cy, Ct = 3, 2, 6, 5, 64, 32
    y = torch.randn(G, B, Hi, Wi, Cy, generator=g)
    tap = torch.randn(B, 2 * Hi, 2 * Wi, Ct, generator=g)
    if dt == L.BF16:
        y = bf16_round(y)
    sc, sh = torch.rand(G, Ct, generator=g) + 0.5, torch.randn(G, Ct, generator=g)
    ref = torch.stack([torch.cat([F.interpolate(y[i].permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1),
                                  tap * sc[i] + sh[i]], -1) for i in range(G)])
    ar = Arena()
    y_dev = y.permute(1, 2, 3, 0, 4).reshape(B, Hi, Wi, G * Cy) if in_slice else y      # one tensor with G channel slices | stacked
    o_y = ar.put(to_dev_bytes(y_dev.contiguous(), dt))
    o_tap, o_sc, o_sh = ar.put(tap), ar.put(sc), ar.put(sh)
    esz = 4 if dt == L.F32 else 2
    o_out = ar.reserve(G * B * 4 * Hi * Wi * (Cy + Ct) * esz)
    ar.materialize()
    run_op(dict(kind=L.OP_UPCAT, flags=L.FLAG_GROUP_IN_SLICE if in_slice else 0, in_dtype=dt, out_dtype=dt, res_dtype=L.F32, B=B, H=Hi, W=Wi,
                Ho=2 * Hi, Wo=2 * Wi, Cin=Cy + Ct, Cin_total=G * Cy if in_slice else Cy, Cout=Cy + Ct, aux0=Cy, aux1=Ct, groups=G,
                in_=o_y, in2=o_tap, out=o_out, scale=o_sc, shift=o_sh), ar)
    out = ar.read(o_out, (G, B, 2 * Hi, 2 * Wi, Cy + Ct), tdtype(dt)).float()
    err = _rel(out, ref)
    _log(f"grouped upcat in_slice={in_slice} dt={dt} rel_err {err:.3e}")
    assert err < (1e-5 if dt == L.F32 else 6e-3)


@pytest.mark.parametrize("mode", ["bf16", "f32", "f32x3"])
@pytest.mark.parametrize("shape", [(2, 32, 48), (1, 21, 19)], ids=["32x48", "21x19_ragged"])
def test_top_fuse_epilogue_plus_tapsum_equals_two_convolutions(shape, mode):
    """FTC_FLAG_TOP_FUSE + FTC_OP_TAPSUM: conv3x3+GELU (192 ch, never stored) followed by a 3x3 top convolution with bias,
    for G heads with 1 / 2 / 1 output channels, against the two convolutions in fp32 (bf16: intermediate rounded to bf16 as the
    kernel's LDS image is; fp32 / fp16x3 plans, round 5: the tap matrix is applied in fp32 FMA straight from the accumulators)."""
    B, H, W = shape
    G, Cin, Cm, TW = 3, 64, 192, 20
    cos = [1, 2, 1]
    chs = [[0], [2, 3], [5]]
    g = torch.Generator().manual_seed(47)
    bf = mode == "bf16"
    rnd = bf16_round if bf else (lambda t: t)
    dt = L.BF16 if bf else L.F32
    x = rnd(torch.randn(G, B, H, W, Cin, generator=g))
    w = rnd(torch.randn(G, Cm, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    bias = torch.randn(G, Cm, generator=g) * 0.3
    wt = [rnd(torch.randn(co, Cm, 3, 3, generator=g) / (Cm * 9) ** 0.5) for co in cos]
    bt = [torch.randn(co, generator=g) * 0.2 for co in cos]
    ref = torch.full((B, H, W, 10), float("nan"))
    for i in range(G):
        y = rnd(F.gelu(F.conv2d(x[i].permute(0, 3, 1, 2).double(), w[i].double(), bias[i].double(), 1, 1)).float())
        o = F.conv2d(y.double(), wt[i].double(), bt[i].double(), 1, 1).float().permute(0, 2, 3, 1)
        for k, ch in enumerate(chs[i]):
            ref[..., ch] = o[..., k]
    wt_mat = torch.zeros(G, 32, Cm)
    omap, ob = [], []
    for i in range(G):
        for o in range(cos[i]):
            for tap in range(9):
                wt_mat[i, tap * cos[i] + o] = wt[i][o, :, tap // 3, tap % 3]
            omap.append((i, o, cos[i], chs[i][o]))
            ob.append(float(bt[i][o]))
    ar = Arena()
    wk = w.permute(0, 1, 3, 4, 2).reshape(G, Cm, 9, Cin)
    o_in = ar.put(to_dev_bytes(x, dt))
    o_w = ar.put(presplit_f16x3(wk) if mode == "f32x3" else to_dev_bytes(wk, dt))
    o_b = ar.put(bias)
    o_wt = ar.put(to_dev_bytes(wt_mat, dt))                 # (fp32 plans: a plain fp32 tap matrix)
    o_map = ar.put(torch.tensor(omap, dtype=torch.int32))
    o_ob = ar.put(torch.tensor(ob))
    o_T = ar.reserve(G * B * H * W * TW * 4)
    o_out = ar.reserve(B * H * W * 10 * 4)
    ar.materialize()
    run_op(dict(kind=L.OP_CONV, flags=L.FLAG_TOP_FUSE | (L.FLAG_SPLIT16 if mode == "f32x3" else 0), act=L.ACT_GELU, in_dtype=dt, out_dtype=dt, w_dtype=dt, B=B, H=H, W=W, Ho=H,
                Wo=W, Cin=Cin, Cin_total=Cin, Cout=Cm, Cout_total=Cm, ksize=3, stride=1, aux0=65, aux1=TW, groups=G, in_=o_in, out=o_T, w=o_w,
                bias=o_b, w2=o_wt), ar)
    run_op(dict(kind=L.OP_TAPSUM, B=B, H=H, W=W, Ho=H, Wo=W, Cout_total=10, aux0=TW, aux1=len(omap), groups=G, in_=o_T, out=o_out, w=o_map,
                bias=o_ob), ar)
    out = ar.read(o_out, (B, H, W, 10), torch.float32)
    used = [c for cc in chs for c in cc]
    err = _rel(out[..., used], ref[..., used])
    _log(f"top_fuse+tapsum {shape} {mode} rel_err {err:.3e}")
    assert err < (1.5e-2 if bf else 2e-5 if mode == "f32x3" else 1e-5)
    raw = ar.buf[o_out:o_out + B * H * W * 40].cpu().view(B * H * W, 40)
    untouched = [c for c in range(10) if c not in used]
    for c in untouched:                                  # channels not listed in the map keep the 0xCD fill
        assert (raw[:, 4 * c:4 * c + 4] == 0xCD).all()


def _upcat_in_fuzz_shapes(n):
    out = []
    for seed in range(n):
        rng = np.random.Generator(np.random.PCG64(700 + seed))
        if rng.integers(0, 2):
            cy, ct = int(rng.choice([64, 128, 192])), int(rng.choice([64, 128]))          # 64-channel K blocks
        else:
            cy, ct = int(rng.choice([64, 192])), int(rng.choice([32, 96]))                 # 32-channel K blocks
        out.append((int(rng.integers(1, 4)), int(rng.integers(1, 3)), 2 * int(rng.integers(3, 26)), 2 * int(rng.integers(3, 26)), cy, ct))
    return out


@pytest.mark.parametrize("shape", [(2, 2, 16, 24, 192, 64), (3, 1, 22, 10, 64, 32), (1, 2, 40, 36, 192, 96), (2, 1, 8, 8, 128, 256)] + _upcat_in_fuzz_shapes(12),
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("dt", [L.BF16, L.F16], ids=["bf16", "f16"])
def test_upcat_in_conv_equals_upsample_concat_conv(shape, dt):
    """FTC_FLAG_UPCAT_IN: conv3x3(cat[bilinear_x2(prev), tapbn]) with the concatenation formed in the halo loader, against
    F.interpolate(align_corners=True) + cat + conv2d in fp32 (upsampled values rounded to bf16 as the kernel's LDS image is)."""
    G, B, H, W, Cy, Ct = shape
    r16 = lambda t: round16(t, dt)
    g = torch.Generator().manual_seed(53)
    prev = r16(torch.randn(G, B, H // 2, W // 2, Cy, generator=g))
    tap = r16(torch.randn(G, B, H, W, Ct, generator=g))
    Cin, Cout = Cy + Ct, 192
    w = r16(torch.randn(G, Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    bias = torch.randn(G, Cout, generator=g) * 0.3
    ref = []
    for i in range(G):
        up = r16(F.interpolate(prev[i].permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True))
        xin = torch.cat([up, tap[i].permute(0, 3, 1, 2)], 1)
        ref.append(F.gelu(F.conv2d(xin, w[i], bias[i], 1, 1)).permute(0, 2, 3, 1))
    ref = torch.stack(ref)
    ar = Arena()
    o_prev = ar.put(to_dev_bytes(prev, dt))
    o_tap = ar.put(to_dev_bytes(tap, dt))
    o_w = ar.put(to_dev_bytes(w.permute(0, 1, 3, 4, 2).reshape(G, Cout, 9, Cin), dt))
    o_b = ar.put(bias)
    o_out = ar.reserve(G * B * H * W * Cout * 2)
    ar.materialize()
    run_op(dict(kind=L.OP_CONV, flags=L.FLAG_UPCAT_IN, act=L.ACT_GELU, in_dtype=dt, out_dtype=dt, w_dtype=dt, B=B, H=H, W=W, Ho=H,
                Wo=W, Cin=Cin, Cin_total=Cy, Cout=Cout, Cout_total=Cout, ksize=3, stride=1, aux0=65, groups=G if G > 1 else 0,
                in_=o_prev, in2=o_tap, out=o_out, w=o_w, bias=o_b), ar)
    out = ar.read(o_out, (G, B, H, W, Cout), tdtype(dt)).float()
    err = _rel(out, ref)
    _log(f"upcat_in conv {shape} rel_err {err:.3e}")
    assert err < TOL16[dt]


def _frag_major(wk):
    """[G, Cout=192, 9, Cin] K-major weights -> FTC_FLAG_W_FRAG layout [G][6 row blocks][9][Cin/64][4][64 lanes][8]."""
    G, Co, T, Ci = wk.shape
    assert Co == 192 and T == 9 and Ci % 64 == 0
    w = wk.reshape(G, 6, 32, 9, Ci // 64, 4, 2, 8)              # g, rb, l31, tap, cb, kg, half, e
    return w.permute(0, 1, 3, 4, 5, 6, 2, 7).contiguous().reshape(G, -1)   # g, rb, tap, cb, kg, (half, l31), e


@pytest.mark.parametrize("kern,Cy,TW", [("halo", 192, 12), ("wl1", 192, 12), ("wl1_walk", 192, 12), ("wl1_walk", 128, 32)],
                         ids=["halo", "wl1", "wl1_walk", "wl1_walk_odd_blocks_wide_T"])
@pytest.mark.parametrize("dt", [L.BF16, L.F16], ids=["bf16", "f16"])
@pytest.mark.parametrize("top", [False, True], ids=["plain", "top_fuse"])
def test_upcat_in_with_tap_batchnorm_folded_and_shared_tap(top, dt, kern, Cy, TW, monkeypatch):
    """Last FPN level as the bf16 plan runs it: the nine heads read ONE backbone tap (FTC_FLAG_GROUP_IN2_SHARED); each head's input
    BatchNorm of the tap is folded into its weights and a 16-case border bias table (FTC_FLAG_BORDER_BIAS) -- against
    conv3x3(cat[upsample(prev_g), BN_g(tap)]) + GELU (then the 3x3 top convolution for the TOP_FUSE variant) in fp32."""
    G, B, H, W, Ct, Cm = 3, 2, 22, 36, 64, 192
    if kern == "wl1_walk":
        # the persistent kernel with 8 workgroups for its 36 tiles: every workgroup walks 4-5 tiles, prefetches the next tile's halo
        # (even number of channel blocks, narrow T) or not (3 blocks / 32-float T rows: no room behind halo buffer 0), and crosses
        # from one head's operands to the next
        monkeypatch.setenv("FTC_WL1_GRID_CAP", "8")
        kern = "wl1"
    r16 = lambda t: round16(t, dt)
    g = torch.Generator().manual_seed(67)
    prev = r16(torch.randn(G, B, H // 2, W // 2, Cy, generator=g))
    tap = r16(torch.randn(B, H, W, Ct, generator=g))
    si, ti = torch.rand(G, Ct, generator=g) + 0.5, torch.randn(G, Ct, generator=g) * 0.5
    Cin = Cy + Ct
    w = torch.randn(G, Cm, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bo = torch.randn(G, Cm, generator=g) * 0.3
    wt = r16(torch.randn(G, 1, Cm, 3, 3, generator=g) / (Cm * 9) ** 0.5)
    bt = torch.randn(G, 1, generator=g) * 0.2
    wm = w.clone()
    wm[:, :, Cy:] *= si[:, None, :, None, None]
    wm = r16(wm)
    b16 = torch.zeros(G, 16, Cm, dtype=torch.float64)
    for i in range(G):
        tmap = torch.einsum("ncrs,c->nrs", w[i, :, Cy:].double(), ti[i].double())
        for idx in range(16):
            rows = [r for r in range(3) if not (r == 0 and idx & 1) and not (r == 2 and idx & 2)]
            cols = [c for c in range(3) if not (c == 0 and idx & 4) and not (c == 2 and idx & 8)]
            b16[i, idx] = bo[i].double() + tmap[:, rows][:, :, cols].sum((1, 2))
    # reference: the folded bf16 weights applied to (upsample, tap) plus what the shift contributes through the in-image taps
    ys = []
    for i in range(G):
        up = r16(F.interpolate(prev[i].permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True))
        xin = torch.cat([up, tap.permute(0, 3, 1, 2)], 1)
        shift = F.conv2d(torch.ones(B, Ct, H, W) * ti[i][None, :, None, None], w[i, :, Cy:], None, 1, 1)      # zero padded: border aware
        ys.append(F.gelu(F.conv2d(xin, wm[i], None, 1, 1) + shift + bo[i][None, :, None, None]))
    ar = Arena()
    o_prev = ar.put(to_dev_bytes(prev, dt))
    o_tap = ar.put(to_dev_bytes(tap, dt))
    wk = wm.permute(0, 1, 3, 4, 2).reshape(G, Cm, 9, Cin)
    # "wl1": the weights-through-L1 kernel (aux0 bits 6+7) on fragment-major weights; "halo": the LDS-ring kernel on K-major weights
    o_w = ar.put(to_dev_bytes(_frag_major(wk) if kern == "wl1" else wk, dt))
    aux0 = 193 if kern == "wl1" else 65
    o_b = ar.put(b16.float())
    flags = L.FLAG_UPCAT_IN | L.FLAG_BORDER_BIAS | L.FLAG_GROUP_IN2_SHARED | (L.FLAG_W_FRAG if kern == "wl1" else 0)
    if not top:
        o_out = ar.reserve(G * B * H * W * Cm * 2)
        ar.materialize()
        run_op(dict(kind=L.OP_CONV, flags=flags, act=L.ACT_GELU, in_dtype=dt, out_dtype=dt, w_dtype=dt, B=B, H=H, W=W, Ho=H, Wo=W,
                    Cin=Cin, Cin_total=Cy, Cout=Cm, Cout_total=Cm, ksize=3, stride=1, aux0=aux0, groups=G, in_=o_prev, in2=o_tap, out=o_out,
                    w=o_w, bias=o_b), ar)
        out = ar.read(o_out, (G, B, H, W, Cm), tdtype(dt)).float()
        ref = torch.stack([y.permute(0, 2, 3, 1) for y in ys])
    else:
        wt_mat = torch.zeros(G, 32, Cm)
        for i in range(G):
            for tp in range(9):
                wt_mat[i, tp] = wt[i, 0, :, tp // 3, tp % 3]
        o_wt = ar.put(to_dev_bytes(wt_mat, dt))
        o_map = ar.put(torch.tensor([(i, 0, 1, i) for i in range(G)], dtype=torch.int32))
        o_ob = ar.put(bt[:, 0].contiguous())
        o_T = ar.reserve(G * B * H * W * TW * 4)
        o_out = ar.reserve(B * H * W * G * 4)
        ar.materialize()
        run_op(dict(kind=L.OP_CONV, flags=flags | L.FLAG_TOP_FUSE, act=L.ACT_GELU, in_dtype=dt, out_dtype=dt, w_dtype=dt, B=B, H=H,
                    W=W, Ho=H, Wo=W, Cin=Cin, Cin_total=Cy, Cout=Cm, Cout_total=Cm, ksize=3, stride=1, aux0=aux0, aux1=TW, groups=G,
                    in_=o_prev, in2=o_tap, out=o_T, w=o_w, bias=o_b, w2=o_wt), ar)
        run_op(dict(kind=L.OP_TAPSUM, B=B, H=H, W=W, Ho=H, Wo=W, Cout_total=G, aux0=TW, aux1=G, groups=G, in_=o_T, out=o_out, w=o_map,
                    bias=o_ob), ar)
        out = ar.read(o_out, (B, H, W, G), torch.float32)
        ref = torch.stack([F.conv2d(r16(ys[i]), wt[i], bt[i], 1, 1)[:, 0] for i in range(G)], -1)
    err = _rel(out, ref)
    _log(f"upcat_in + folded tap BN (top={top}) rel_err {err:.3e}")
    assert err < TOL16[dt]
    # the image border is where a wrong border-bias case would show: same bound on the outermost ring alone
    ring = torch.ones(H, W, dtype=torch.bool)
    ring[1:-1, 1:-1] = False
    sel = (lambda a: a[:, :, ring]) if not top else (lambda a: a[:, ring])
    assert float((sel(out) - sel(ref)).abs().max()) < TOL16[dt] * float(ref.abs().max())


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
def test_conv_border_bias_folds_preceding_batchnorm(dt):
    """conv3x3(zero_pad(x*s + t)) == conv3x3_{W*s}(zero_pad(x)) + bias_table[border case]: how the
    per-head input BatchNorm of FPN level 0 is folded (Leafmap.forward, detector.py:194-197)."""
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 2, 6, 7, 64, 192
    x = torch.randn(B, H, W, Cin, generator=g)
    if dt == L.BF16:
        x = bf16_round(x)
    si, ti = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bo = torch.randn(Cout, generator=g) * 0.2
    ref = F.gelu(F.conv2d((x * si + ti).permute(0, 3, 1, 2), w, bo, 1, 1)).permute(0, 2, 3, 1)
    wm = w * si[None, :, None, None]
    tmap = torch.einsum("ncrs,c->nrs", w.double(), ti.double())
    b16 = torch.zeros(16, Cout, dtype=torch.float64)
    for idx in range(16):
        rows = [r for r in range(3) if not (r == 0 and idx & 1) and not (r == 2 and idx & 2)]
        cols = [c for c in range(3) if not (c == 0 and idx & 4) and not (c == 2 and idx & 8)]
        b16[idx] = bo.double() + tmap[:, rows][:, :, cols].sum((1, 2))
    ar = Arena()
    o_in = ar.put(to_dev_bytes(x, dt))
    o_w = ar.put(to_dev_bytes(wm.permute(0, 2, 3, 1).reshape(Cout, 9, Cin), dt))
    o_b = ar.put(b16.float())
    o_out = ar.reserve(B * H * W * Cout * (4 if dt == L.F32 else 2))
    ar.materialize()
    run_op(dict(kind=L.OP_CONV, flags=L.FLAG_BORDER_BIAS, act=L.ACT_GELU, in_dtype=dt, out_dtype=dt, w_dtype=dt, B=B, H=H, W=W, Ho=H, Wo=W,
                Cin=Cin, Cin_total=Cin, Cout=Cout, Cout_total=Cout, ksize=3, stride=1, in_=o_in, out=o_out, w=o_w, bias=o_b), ar)
    out = ar.read(o_out, (B, H, W, Cout), tdtype(dt)).float()
    err = _rel(out, ref)
    _log(f"conv border-bias dt={dt} rel_err {err:.3e}")
    assert err < (1e-5 if dt == L.F32 else 2e-2)


@pytest.mark.parametrize("aux0", [0, 4 + 32 + 512, 2 + 16 + 512, 7 + 16 + 512 + 1024], ids=["default", "64x128_dma2", "128x128_reg", "64x64_splitk2"])
@pytest.mark.parametrize("kblock", [False, True], ids=["nhwc", "kblock32"])
def test_conv_dual_output_bf16_copy(kblock, aux0):
    """fp32 trunk output + bf16 copy (ftc_op.out2) written by the same epilogue; with FTC_FLAG_KBLOCK32 the copy is stored in 32-channel
    planes [B][Cout/32][H*W][32] (what FTC_OP_MBHEAD streams) -- every epilogue form (direct, LDS-staged, split-K)."""
    g = torch.Generator().manual_seed(77)
    B, H, W, Cin, Cout = 2, 16, 16, 384, 64
    x = bf16_round(torch.randn(B, H, W, Cin, generator=g))
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.3
    res = torch.randn(B, H, W, Cout, generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2), bf16_round(w), None).permute(0, 2, 3, 1) + bias + res
    ar = Arena()
    o_in, o_w, o_b, o_res = ar.put(to_dev_bytes(x, L.BF16)), ar.put(to_dev_bytes(w.reshape(Cout, 1, Cin), L.BF16)), ar.put(bias), ar.put(res)
    o_out, o_out2 = ar.reserve(B * H * W * Cout * 4), ar.reserve(B * H * W * Cout * 2)
    ar.materialize()
    run_op(dict(kind=L.OP_CONV, flags=L.FLAG_RESIDUAL | (L.FLAG_KBLOCK32 if kblock else 0), act=L.ACT_NONE, in_dtype=L.BF16, out_dtype=L.F32, w_dtype=L.BF16,
                B=B, H=H, W=W, Ho=H, Wo=W, Cin=Cin, Cin_total=Cin, Cout=Cout, Cout_total=Cout, ksize=1, stride=1, res_dtype=L.F32, aux0=aux0,
                in_=o_in, in2=o_res, out=o_out, out2=o_out2, w=o_w, bias=o_b), ar)
    out = ar.read(o_out, (B, H, W, Cout), torch.float32)
    if kblock:
        out2 = ar.read(o_out2, (B, Cout // 32, H * W, 32), torch.bfloat16).permute(0, 2, 1, 3).reshape(B, H, W, Cout)
    else:
        out2 = ar.read(o_out2, (B, H, W, Cout), torch.bfloat16)
    assert _rel(out, ref) < 1.5e-2
    assert torch.equal(out2, out.to(torch.bfloat16))            # the copy is the RNE rounding of the fp32 value


@pytest.mark.parametrize("C0", [32, 24, 12])
@pytest.mark.parametrize("odt", [L.F32, L.BF16, L.F16])
@pytest.mark.parametrize("nchw", [False, True])
def test_stem(nchw, odt, C0):
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 64, 94
    x = torch.rand(B, H, W, 3, generator=g)
    w = torch.randn(C0, 3, 3, 3, generator=g) * 0.3
    bias = torch.randn(C0, generator=g) * 0.2
    ref = F.silu(F.conv2d((x * 2 - 1).permute(0, 3, 1, 2), w, bias, 2, 1)).permute(0, 2, 3, 1)
    ar = Arena()
    o_in = ar.put(x.permute(0, 3, 1, 2).contiguous() if nchw else x)
    o_w = ar.put(w.permute(2, 3, 1, 0).reshape(27, C0).contiguous())
    o_b = ar.put(bias)
    o_out = ar.reserve(B * (H // 2) * (W // 2) * C0 * 4)
    ar.materialize()
    run_op(dict(kind=L.OP_STEM, flags=L.FLAG_IN_NCHW if nchw else 0, act=L.ACT_SILU, in_dtype=L.F32, out_dtype=odt, B=B, H=H, W=W,
                Ho=H // 2, Wo=W // 2, Cin=3, Cout=C0, ksize=3, stride=2, in_=o_in, out=o_out, w=o_w, bias=o_b), ar)
    out = ar.read(o_out, (B, H // 2, W // 2, C0), tdtype(odt)).float()
    err = _rel(out, ref)
    _log(f"stem nchw={nchw} odt={odt} rel_err {err:.3e}")
    assert err < (1e-5 if odt == L.F32 else 6e-3)


@pytest.mark.parametrize("dt", [L.F32, L.BF16, L.F16])
@pytest.mark.parametrize("shape", [(2, 48, 48, 192, 1), (2, 48, 48, 128, 2), (1, 21, 13, 72, 1), (1, 21, 13, 72, 2), (3, 24, 24, 3840, 1)])
def test_dwconv_and_se(shape, dt):
    B, H, W, Cc, stride = shape
    g = torch.Generator().manual_seed(11)
    x = round16(torch.randn(B, H, W, Cc, generator=g), dt)
    w = torch.randn(Cc, 1, 3, 3, generator=g) * 0.4
    bias = torch.randn(Cc, generator=g) * 0.2
    ref = F.silu(F.conv2d(x.permute(0, 3, 1, 2), w, bias, stride, 1, 1, Cc)).permute(0, 2, 3, 1)
    Ho, Wo = ref.shape[1:3]
    th = 8 if stride == 1 else 4
    P = ((Ho + th - 1) // th) * ((Wo + 7) // 8)
    S = max(1, Cc // 24)
    w1 = torch.randn(S, Cc, generator=g) / Cc ** 0.5
    b1 = torch.randn(S, generator=g) * 0.3
    w2 = torch.randn(Cc, S, generator=g) / S ** 0.5
    b2 = torch.randn(Cc, generator=g) * 0.3
    ar = Arena()
    o_in = ar.put(to_dev_bytes(x, dt))
    o_w = ar.put(w.reshape(Cc, 9).t().contiguous())
    o_b = ar.put(bias)
    esz = 4 if dt == L.F32 else 2
    o_out = ar.reserve(B * Ho * Wo * Cc * esz)
    o_part = ar.reserve(B * P * Cc * 4)
    o_w1, o_b1, o_w2t, o_b2 = ar.put(w1), ar.put(b1), ar.put(w2.t().contiguous()), ar.put(b2)
    o_scale = ar.reserve(B * Cc * 4)
    o_hid = ar.reserve(B * S * 4)
    ar.materialize()
    run_op(dict(kind=L.OP_DWCONV, act=L.ACT_SILU, in_dtype=dt, out_dtype=dt, B=B, H=H, W=W, Ho=Ho, Wo=Wo, Cin=Cc, Cout=Cc, ksize=3,
                stride=stride, aux0=P, in_=o_in, out=o_out, w=o_w, bias=o_b, aux=o_part), ar)
    out = ar.read(o_out, (B, Ho, Wo, Cc), tdtype(dt)).float()
    err = _rel(out, ref)
    part = ar.read(o_part, (B, P, Cc), torch.float32)
    mean = part.sum(1) / (Ho * Wo)
    # the kernel sums its fp32 results BEFORE narrowing to the storage dtype
    err_mean = float((mean - ref.mean((1, 2))).abs().max())
    _log(f"dwconv {shape} dt={dt} rel_err {err:.3e} mean_err {err_mean:.3e}")
    assert err < (1e-5 if dt == L.F32 else 6e-3)
    assert err_mean < (1e-5 if dt == L.F32 else 2e-3)
    run_op(dict(kind=L.OP_SE, B=B, H=Ho, W=Wo, Cin=Cc, Cout=Cc, aux0=S, aux1=P, aux=o_part, out=o_scale, in2=o_hid, w=o_w1, w2=o_w2t,
                bias=o_b1, bias2=o_b2), ar)
    sc = ar.read(o_scale, (B, Cc), torch.float32)
    ref_sc = torch.sigmoid(F.silu(mean @ w1.t() + b1) @ w2.t() + b2)
    err_se = float((sc - ref_sc).abs().max())
    _log(f"se {shape} abs_err {err_se:.3e}")
    assert err_se < 2e-6


@pytest.mark.parametrize("kblock", [False, True], ids=["nhwc", "kblock32"])
@pytest.mark.parametrize("dt", [L.BF16, L.F16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(8, 24, 24, 512, 384, 24, 0), (3, 24, 24, 640, 128, 160, 0), (2, 8, 8, 64, 128, 7, 0), (5, 4, 4, 96, 256, 16, 0),
                                   (1, 16, 20, 32, 128, 3, 0), (2, 24, 23, 64, 128, 5, 0),
                                   # band mode (aux1 = output rows per band, + one halo row above and below)
                                   (2, 48, 48, 256, 256, 64, 10), (3, 48, 48, 192, 128, 48, 10), (2, 24, 24, 64, 128, 9, 7), (1, 40, 30, 32, 128, 4, 17),
                                   (2, 12, 12, 32, 128, 4, 1)],
                         ids=["24x24_512_384", "24x24_640_128_s160", "8x8", "4x4", "16x20", "24x23", "48x48_band10", "48x48_192_band10", "24x24_band7",
                              "40x30_band17", "12x12_band1"])
def test_mbconv_slice_head_and_se_from_partial_products(shape, dt, kblock):
    """FTC_OP_MBHEAD (csrc/mbconv_slice.hip): expand 1x1 + BN + SiLU -> depthwise 3x3 + BN + SiLU -> channel sums + per-slice fc1 partial
    products in one launch, against the same chain in fp32 on the CPU (expanded tensor rounded to the 16-bit type, as the three-kernel
    path stores it); then FTC_OP_SE with FTC_FLAG_SE_HPART (with and without the per-image weight fold) against the SE MLP on those
    means.  torchvision MBConv block[0..2] as instantiated by /root/reference/models/detector.py:17-20."""
    _mbhead_case(shape, dt, kblock, L.MBHEAD_SLICE)


@pytest.mark.parametrize("dt", [L.BF16, L.F16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(8, 24, 24, 512, 384, 24, 0), (3, 24, 24, 640, 192, 160, 0), (2, 8, 8, 64, 96, 7, 0), (1, 16, 20, 32, 288, 3, 0),
                                   (2, 48, 48, 256, 288, 64, 10), (2, 24, 24, 64, 96, 9, 7)],
                         ids=["24x24_512_384", "24x24_640_192_s160", "8x8", "16x20", "48x48_band10", "24x24_band7"])
def test_mbconv_slice_head_96_channel_slices(shape, dt):
    """FTC_OP_MBHEAD with 96-channel slices (ftc_op.Cout_total = 96; round 5: stage 6 at batch 8 then launches 256 workgroups instead of 192 on the
    256 CUs): same checks as the 128-channel form, NHWC and 32-channel-plane input."""
    _mbhead_case(shape, dt, False, 96)
    _mbhead_case(shape, dt, True, 96)


def _mbhead_case(shape, dt, kblock, slice_w):
    B, H, W, K, Cc, S, R = shape
    NB = -(-H // R) if R else 1
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cc)
    r16 = lambda t: round16(t, dt)
    x = r16(torch.randn(B, H, W, K, generator=g))
    we = r16(torch.randn(Cc, K, generator=g) / K ** 0.5 * 1.5)
    be = torch.randn(Cc, generator=g) * 0.3
    wd = torch.randn(Cc, 1, 3, 3, generator=g) * 0.4
    bd = torch.randn(Cc, generator=g) * 0.2
    w1 = torch.randn(S, Cc, generator=g) / Cc ** 0.5
    b1 = torch.randn(S, generator=g) * 0.3
    w2 = torch.randn(Cc, S, generator=g) / S ** 0.5
    b2 = torch.randn(Cc, generator=g) * 0.3
    e = r16(F.silu(x.reshape(-1, K) @ we.t() + be)).reshape(B, H, W, Cc)
    ref = F.silu(F.conv2d(e.permute(0, 3, 1, 2), wd, bd, 1, 1, 1, Cc)).permute(0, 2, 3, 1)
    NS = Cc // slice_w
    ar = Arena()
    xdev = x.reshape(B, H * W, K // 32, 32).permute(0, 2, 1, 3) if kblock else x      # FTC_FLAG_KBLOCK32: [B][K/32][H*W][32]
    o_x, o_we, o_be = ar.put(to_dev_bytes(xdev, dt)), ar.put(to_dev_bytes(we, dt)), ar.put(be)
    o_wd, o_bd = ar.put(wd.reshape(Cc, 9).t().contiguous()), ar.put(bd)
    o_w1, o_b1, o_w2t, o_b2 = ar.put(w1), ar.put(b1), ar.put(w2.t().contiguous()), ar.put(b2)
    o_out = ar.reserve(B * H * W * Cc * 2)
    o_sums, o_hp = ar.reserve(B * NB * Cc * 4), ar.reserve(B * NB * NS * S * 4)
    o_scale, o_hid = ar.reserve(B * Cc * 4), ar.reserve(B * S * 4)
    N = 96
    wp = r16(torch.randn(N, Cc, generator=g) / Cc ** 0.5)
    o_wp, o_wb = ar.put(to_dev_bytes(wp, dt)), ar.reserve(B * N * Cc * 2)
    ar.materialize()
    run_op(dict(kind=L.OP_MBHEAD, flags=L.FLAG_KBLOCK32 if kblock else 0, act=L.ACT_SILU, in_dtype=dt, out_dtype=dt, w_dtype=dt, B=B, H=H, W=W, Ho=H, Wo=W, Cin=K, Cout=Cc, Cout_total=0 if slice_w == L.MBHEAD_SLICE else slice_w, ksize=3, stride=1,
                aux0=S, aux1=R, in_=o_x, w2=o_we, bias2=o_be, w=o_wd, bias=o_bd, out=o_out, aux=o_sums, scale=o_w1, out2=o_hp), ar)
    out = ar.read(o_out, (B, H, W, Cc), tdtype(dt)).float()
    err = _rel(out, ref)
    bsums = ar.read(o_sums, (B, NB, Cc), torch.float32)                      # per band: the channel sums of its output rows
    mean = bsums.sum(1) / (H * W)
    err_mean = float((mean - ref.mean((1, 2))).abs().max())
    hp = ar.read(o_hp, (B, NB, NS, S), torch.float32)
    want_hp = torch.einsum("bnjc,sjc->bnjs", (bsums / (H * W)).reshape(B, NB, NS, slice_w), w1.reshape(S, NS, slice_w))
    err_hp = float((hp - want_hp).abs().max())
    if R:
        want_b = torch.stack([ref[:, j * R:(j + 1) * R].sum((1, 2)) for j in range(NB)], 1)
        assert float((bsums - want_b).abs().max()) < (3e-3 if dt == L.BF16 else 5e-4) * H * W
    _log(f"mbhead {shape} dt={dt} rel_err {err:.3e} mean_err {err_mean:.3e} hpart_err {err_hp:.3e}")
    # one more rounding of the expanded tensor than the plain depthwise test: an element of e one ulp off moves 9 outputs
    assert err < (1.2e-2 if dt == L.BF16 else 2e-3)
    assert err_mean < (3e-3 if dt == L.BF16 else 5e-4)
    assert err_hp < 2e-6 * max(1.0, float(want_hp.abs().max()))
    ref_sc = torch.sigmoid(F.silu(mean @ w1.t() + b1) @ w2.t() + b2)
    run_op(dict(kind=L.OP_SE, flags=L.FLAG_SE_HPART, B=B, H=H, W=W, Cin=Cc, Cout=Cc, aux0=S, aux1=NB * NS, aux=o_hp, out=o_scale, in2=o_hid, w2=o_w2t,
                bias=o_b1, bias2=o_b2), ar)
    sc = ar.read(o_scale, (B, Cc), torch.float32)
    assert float((sc - ref_sc).abs().max()) < 3e-6
    ar.buf[o_scale:o_scale + B * Cc * 4] = 0xCD
    run_op(dict(kind=L.OP_SE, flags=L.FLAG_SE_HPART | L.FLAG_SE_FOLD, w_dtype=dt, B=B, H=H, W=W, Cin=Cc, Cout=Cc, Cout_total=N, aux0=S, aux1=NB * NS, aux=o_hp,
                out=o_scale, in2=o_hid, w2=o_w2t, bias=o_b1, bias2=o_b2, in_=o_wp, out2=o_wb), ar)
    sc2 = ar.read(o_scale, (B, Cc), torch.float32)
    assert float((sc2 - ref_sc).abs().max()) < 3e-6
    wb = ar.read(o_wb, (B, N, Cc), tdtype(dt)).float()
    want = r16(wp[None] * sc2[:, None, :])
    assert float((wb - want).abs().max()) <= float(want.abs().max()) * (2 ** -8 if dt == L.BF16 else 2 ** -11)


def _unsplit_f16x3(raw: torch.Tensor, shape) -> torch.Tensor:
    """inverse of presplit_f16x3: uint8 bytes -> fp32 values hi + lo"""
    h = raw.view(torch.float16).reshape(-1, 8).float()
    return (h[:, :4] + h[:, 4:]).reshape(shape)


@pytest.mark.parametrize("shape", [(8, 24, 24, 512, 192, 24, 0), (3, 24, 24, 640, 128, 160, 0), (2, 8, 8, 64, 64, 7, 0), (1, 16, 20, 32, 128, 3, 0), (2, 24, 23, 96, 64, 5, 0),
                                   (2, 48, 48, 256, 128, 64, 10), (3, 48, 48, 192, 64, 48, 10), (2, 24, 24, 64, 128, 9, 7), (1, 40, 30, 32, 64, 4, 17)],
                         ids=["24x24_512_192", "24x24_640_128_s160", "8x8", "16x20", "24x23", "48x48_band10", "48x48_192_band10", "24x24_band7", "40x30_band17"])
def test_mbconv_slice_head_fp16x3(shape):
    """FTC_OP_MBHEAD in the fp32-tensor form (in_dtype = FTC_F32 + FTC_FLAG_SPLIT16, csrc/mbconv_slice_x3.hip; round 5): pre-split input and
    weights, 64-channel slices, fp32 expanded image in LDS, expf SiLU -- against the chain in float64 on the CPU (torchvision MBConv block[0..2]
    as instantiated by /root/reference/models/detector.py:17-20), held to the fp32 plans' tolerance; the pre-split copy is produced by an
    fp16x3 FTC_OP_CONV with out2 (the producer in the plan); then FTC_OP_SE with FTC_FLAG_SE_HPART | SE_FOLD | SPLIT16 on the partial products."""
    B, H, W, K, Cc, S, R = shape
    NB = -(-H // R) if R else 1
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cc)
    Kp = 64                                                           # the producer: a 1x1 fp16x3 convolution Kp -> K whose out2 is the head's input
    x0 = torch.randn(B, H, W, Kp, generator=g)
    wprod = torch.randn(K, Kp, generator=g) / Kp ** 0.5
    bprod = torch.randn(K, generator=g) * 0.1
    we = torch.randn(Cc, K, generator=g) / K ** 0.5 * 1.5
    be = torch.randn(Cc, generator=g) * 0.3
    wd = torch.randn(Cc, 1, 3, 3, generator=g) * 0.4
    bd = torch.randn(Cc, generator=g) * 0.2
    w1 = torch.randn(S, Cc, generator=g) / Cc ** 0.5
    b1 = torch.randn(S, generator=g) * 0.3
    w2 = torch.randn(Cc, S, generator=g) / S ** 0.5
    b2 = torch.randn(Cc, generator=g) * 0.3
    NS = Cc // 64
    N = 96
    wp = torch.randn(N, Cc, generator=g) / Cc ** 0.5
    ar = Arena()
    o_x0, o_wprod, o_bprod = ar.put(x0), ar.put(presplit_f16x3(wprod)), ar.put(bprod)
    o_x, o_xs = ar.reserve(B * H * W * K * 4), ar.reserve(B * H * W * K * 4)
    o_we, o_be = ar.put(presplit_f16x3(we)), ar.put(be)
    o_wd, o_bd = ar.put(wd.reshape(Cc, 9).t().contiguous()), ar.put(bd)
    o_w1, o_b1, o_w2t, o_b2 = ar.put(w1), ar.put(b1), ar.put(w2.t().contiguous()), ar.put(b2)
    o_out = ar.reserve(B * H * W * Cc * 4)
    o_sums, o_hp = ar.reserve(B * NB * Cc * 4), ar.reserve(B * NB * NS * S * 4)
    o_scale, o_hid = ar.reserve(B * Cc * 4), ar.reserve(B * S * 4)
    o_wp, o_wb = ar.put(presplit_f16x3(wp)), ar.reserve(B * N * Cc * 4)
    o_dps, o_y0, o_y1, o_bp = ar.reserve(B * H * W * Cc * 4), ar.reserve(B * H * W * N * 4), ar.reserve(B * H * W * N * 4), ar.put(torch.zeros(N))
    ar.materialize()
    run_op(dict(kind=L.OP_CONV, flags=L.FLAG_SPLIT16, act=L.ACT_NONE, in_dtype=L.F32, out_dtype=L.F32, w_dtype=L.F32, B=B, H=H, W=W, Ho=H, Wo=W, Cin=Kp, Cin_total=Kp,
                Cout=K, Cout_total=K, ksize=1, stride=1, in_=o_x0, w=o_wprod, bias=o_bprod, out=o_x, out2=o_xs), ar)
    x = ar.read(o_x, (B, H, W, K), torch.float32)
    assert _rel(x.reshape(-1, K), x0.reshape(-1, Kp) @ wprod.t() + bprod) < 2e-5
    xs = _unsplit_f16x3(ar.buf[o_xs:o_xs + B * H * W * K * 4].cpu(), (B, H, W, K))
    assert float((xs - x).abs().max()) <= float(x.abs().max()) * 2 ** -21           # hi + lo carries 22 significand bits
    assert torch.equal(ar.buf[o_xs:o_xs + B * H * W * K * 4].cpu(), presplit_f16x3(x))   # ... and is exactly the split of the stored fp32 value
    run_op(dict(kind=L.OP_MBHEAD, flags=L.FLAG_SPLIT16, act=L.ACT_SILU, in_dtype=L.F32, out_dtype=L.F32, w_dtype=L.F32, B=B, H=H, W=W, Ho=H, Wo=W, Cin=K, Cout=Cc, ksize=3,
                stride=1, aux0=S, aux1=R, in_=o_xs, w2=o_we, bias2=o_be, w=o_wd, bias=o_bd, out=o_out, aux=o_sums, scale=o_w1, out2=o_hp), ar)
    xd = x.double()
    e = F.silu(xd.reshape(-1, K) @ we.double().t() + be.double()).reshape(B, H, W, Cc)
    ref = F.silu(F.conv2d(e.permute(0, 3, 1, 2), wd.double(), bd.double(), 1, 1, 1, Cc)).permute(0, 2, 3, 1)
    out = ar.read(o_out, (B, H, W, Cc), torch.float32)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    bsums = ar.read(o_sums, (B, NB, Cc), torch.float32)
    mean = bsums.sum(1) / (H * W)
    err_mean = float((mean.double() - ref.mean((1, 2))).abs().max())
    hp = ar.read(o_hp, (B, NB, NS, S), torch.float32)
    want_hp = torch.einsum("bnjc,sjc->bnjs", (bsums / (H * W)).reshape(B, NB, NS, 64), w1.reshape(S, NS, 64))
    err_hp = float((hp - want_hp).abs().max())
    if R:
        want_b = torch.stack([ref[:, j * R:(j + 1) * R].sum((1, 2)) for j in range(NB)], 1)
        assert float((bsums.double() - want_b).abs().max()) < 2e-5 * H * W
    _log(f"mbhead fp16x3 {shape} rel_err {err:.3e} mean_err {err_mean:.3e} hpart_err {err_hp:.3e}")
    assert err < 2e-5 and err_mean < 2e-5
    assert err_hp < 2e-6 * max(1.0, float(want_hp.abs().max()))
    ref_sc = torch.sigmoid(F.silu(mean @ w1.t() + b1) @ w2.t() + b2)
    run_op(dict(kind=L.OP_SE, flags=L.FLAG_SE_HPART | L.FLAG_SE_FOLD | L.FLAG_SPLIT16, w_dtype=L.F32, B=B, H=H, W=W, Cin=Cc, Cout=Cc, Cout_total=N, aux0=S, aux1=NB * NS,
                aux=o_hp, out=o_scale, in2=o_hid, w2=o_w2t, bias=o_b1, bias2=o_b2, in_=o_wp, out2=o_wb), ar)
    sc = ar.read(o_scale, (B, Cc), torch.float32)
    assert float((sc - ref_sc).abs().max()) < 3e-6
    wb = _unsplit_f16x3(ar.buf[o_wb:o_wb + B * N * Cc * 4].cpu(), (B, N, Cc))
    want = _unsplit_f16x3(presplit_f16x3(wp), (N, Cc))[None] * sc[:, None, :]
    assert float((wb - want).abs().max()) <= float(want.abs().max()) * 2 ** -20
    # FTC_FLAG_PRESPLIT: the head writes its output pre-split, and the project convolution (per-image folded weights, FTC_FLAG_W_PER_IMAGE) consumes
    # it without splitting: the same bits as the fp32 output split on the way, so the two project results are IDENTICAL
    if (H * W) % 64 == 0:
        run_op(dict(kind=L.OP_MBHEAD, flags=L.FLAG_SPLIT16 | L.FLAG_PRESPLIT, act=L.ACT_SILU, in_dtype=L.F32, out_dtype=L.F32, w_dtype=L.F32, B=B, H=H, W=W, Ho=H, Wo=W, Cin=K,
                    Cout=Cc, ksize=3, stride=1, aux0=S, aux1=R, in_=o_xs, w2=o_we, bias2=o_be, w=o_wd, bias=o_bd, out=o_dps, aux=o_sums, scale=o_w1, out2=o_hp), ar)
        assert torch.equal(ar.buf[o_dps:o_dps + B * H * W * Cc * 4].cpu(), presplit_f16x3(out))
        for flags, src, dst in ((0, o_out, o_y0), (L.FLAG_PRESPLIT, o_dps, o_y1)):
            run_op(dict(kind=L.OP_CONV, flags=L.FLAG_SPLIT16 | L.FLAG_W_PER_IMAGE | flags, act=L.ACT_NONE, in_dtype=L.F32, out_dtype=L.F32, w_dtype=L.F32, B=B, H=H, W=W, Ho=H,
                        Wo=W, Cin=Cc, Cin_total=Cc, Cout=N, Cout_total=N, ksize=1, stride=1, in_=src, w=o_wb, bias=o_bp, out=dst), ar)
        y0, y1 = ar.read(o_y0, (B, H * W, N), torch.float32), ar.read(o_y1, (B, H * W, N), torch.float32)
        ref_y = torch.einsum("bpc,bnc->bpn", out.reshape(B, H * W, Cc).double(), wb.double())
        assert float((y0.double() - ref_y).abs().max() / ref_y.abs().max()) < 2e-5
        assert torch.equal(y0, y1)


def _fbits(v):
    import struct
    return struct.unpack("<i", struct.pack("<f", v))[0]


@pytest.mark.parametrize("cfg", [("f32", L.F32, L.F32, L.BF16), ("f32_to_bf16", L.F32, L.BF16, L.BF16), ("bf16", L.BF16, L.BF16, L.BF16), ("f16_to_f32", L.F16, L.F32, L.F16)],
                         ids=lambda c: c[0])
@pytest.mark.parametrize("shape", [(2, 12, 10, 96), (3, 5, 7, 200), (8, 24, 24, 64)], ids=["12x10x96", "5x7x200", "24x24x64"])
def test_training_mode_batchnorm_ops(shape, cfg):
    """FTC_OP_BNSTAT + FTC_OP_BNACT (csrc/train_ops.hip) against torch.nn.functional.batch_norm(training=True): normalised output with
    activation, StochasticDepth keep-scale and residual, the SE squeeze sums, and the running statistics (momentum, unbiased variance)."""
    _, idt, odt, tc = cfg
    B, H, W, C = shape
    g = torch.Generator().manual_seed(B * 100 + C)
    x = round16(torch.randn(B, H, W, C, generator=g) * 1.7 + 0.4, idt)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    rm, rv = torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5
    res = torch.randn(B, H, W, C, generator=g)
    keep = torch.tensor([0.0 if b % 3 == 1 else 1.25 for b in range(B)])
    eps, mom = 1e-3, 0.1
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(x.permute(0, 3, 1, 2), rm_ref, rv_ref, gamma, beta, True, mom, eps)
    ref = F.silu(y).permute(0, 2, 3, 1) * keep[:, None, None, None] + res
    M = B * H * W
    nchunk = max(1, min(512, -(-M // 64)))
    P = 3
    ar = Arena()
    o_x, o_g, o_b, o_res, o_keep = ar.put(to_dev_bytes(x, idt)), ar.put(gamma), ar.put(beta), ar.put(res), ar.put(keep)
    o_run = ar.put(torch.stack([rm, rv]))
    o_ss, o_part = ar.reserve(4 * C * 4), ar.reserve(nchunk * 2 * C * 8)      # scale | shift | mean | 1/std
    o_out = ar.reserve(M * C * (4 if odt == L.F32 else 2))
    o_out2 = ar.reserve(M * C * 2) if odt == L.F32 else None
    o_sums = ar.reserve(B * P * C * 4)
    ar.materialize()
    run_op(dict(kind=L.OP_BNSTAT, in_dtype=idt, B=B, H=H, W=W, Cin=C, aux0=_fbits(eps), aux1=_fbits(mom), in_=o_x, w=o_g, bias=o_b, aux=o_run,
                out=o_ss, in2=o_part), ar)
    run_op(dict(kind=L.OP_BNACT, flags=L.FLAG_RESIDUAL, act=L.ACT_SILU, in_dtype=idt, out_dtype=odt, w_dtype=tc, res_dtype=L.F32, B=B, H=H, W=W, Cin=C,
                aux0=P, in_=o_x, scale=o_ss, shift=o_ss + C * 4, in2=o_res, w2=o_keep, out=o_out, out2=o_out2, aux=o_sums), ar)
    run = ar.read(o_run, (2, C), torch.float32)
    assert float((run[0] - rm_ref).abs().max()) < 2e-6 and float((run[1] - rv_ref).abs().max()) < 2e-5
    out = ar.read(o_out, (B, H, W, C), tdtype(odt)).float()
    err = _rel(out, ref)
    _log(f"bnstat+bnact {shape} {cfg[0]} rel_err {err:.3e}")
    assert err < (5e-6 if odt == L.F32 else TOL16[odt])
    if o_out2 is not None:
        c2 = ar.read(o_out2, (B, H, W, C), tdtype(tc)).float()
        assert torch.equal(c2, round16(out, tc))
    sums = ar.read(o_sums, (B, P, C), torch.float32).sum(1)
    assert float((sums - ref.sum((1, 2))).abs().max()) < 2e-3 * float(ref.sum((1, 2)).abs().max() + 1)


@pytest.mark.parametrize("cfg", [("f32", L.F32, L.F32), ("bf16_f32tap", L.BF16, L.F32), ("bf16", L.BF16, L.BF16), ("f16_f32tap", L.F16, L.F32),
                                 ("f16", L.F16, L.F16)], ids=lambda c: c[0])
@pytest.mark.parametrize("with_y", [True, False])
def test_upcat(with_y, cfg):
    _, dt, tdt = cfg
    g = torch.Generator().manual_seed(5)
    B, Hi, Wi, Cy, Ct = 2, 12, 10, 192, 96
    Ho, Wo = (2 * Hi, 2 * Wi) if with_y else (Hi, Wi)
    y = torch.randn(B, Hi, Wi, Cy, generator=g)
    tap = torch.randn(B, Ho, Wo, Ct, generator=g)
    y, tap = round16(y, dt), round16(tap, tdt)
    sc, sh = torch.rand(Ct, generator=g) + 0.5, torch.randn(Ct, generator=g)
    parts = []
    if with_y:
        parts.append(F.interpolate(y.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1))
    parts.append(tap * sc + sh)
    ref = torch.cat(parts, -1)
    ar = Arena()
    o_y = ar.put(to_dev_bytes(y, dt)) if with_y else None
    o_tap = ar.put(to_dev_bytes(tap, tdt))
    o_sc, o_sh = ar.put(sc), ar.put(sh)
    Ctot = (Cy if with_y else 0) + Ct
    o_out = ar.reserve(B * Ho * Wo * Ctot * (4 if dt == L.F32 else 2))
    ar.materialize()
    run_op(dict(kind=L.OP_UPCAT, in_dtype=dt, out_dtype=dt, res_dtype=tdt, B=B, H=Hi, W=Wi, Ho=Ho, Wo=Wo, Cin=Ctot, Cout=Ctot,
                aux0=Cy if with_y else 0, aux1=Ct, in_=o_y, in2=o_tap, out=o_out, scale=o_sc, shift=o_sh), ar)
    out = ar.read(o_out, (B, Ho, Wo, Ctot), tdtype(dt)).float()
    err = _rel(out, ref)
    _log(f"upcat with_y={with_y} {cfg[0]} rel_err {err:.3e}")
    assert err < (2e-6 if dt == L.F32 else 5e-3)


def test_upcat_reads_channel_slice_of_wider_tensor():
    g = torch.Generator().manual_seed(8)
    B, Hi, Wi, Cy, Ct, CyT, off = 1, 6, 5, 192, 64, 576, 192
    wide = torch.randn(B, Hi, Wi, CyT, generator=g)
    tap = torch.randn(B, 2 * Hi, 2 * Wi, Ct, generator=g)
    sc, sh = torch.rand(Ct, generator=g) + 0.5, torch.randn(Ct, generator=g)
    y = wide[..., off:off + Cy]
    ref = torch.cat([F.interpolate(y.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1),
                     tap * sc + sh], -1)
    ar = Arena()
    o_y, o_tap, o_sc, o_sh = ar.put(wide), ar.put(tap), ar.put(sc), ar.put(sh)
    o_out = ar.reserve(B * 4 * Hi * Wi * (Cy + Ct) * 4)
    ar.materialize()
    run_op(dict(kind=L.OP_UPCAT, in_dtype=L.F32, out_dtype=L.F32, res_dtype=L.F32, B=B, H=Hi, W=Wi, Ho=2 * Hi, Wo=2 * Wi, Cin=Cy + Ct,
                Cin_total=CyT, cin_off=off, Cout=Cy + Ct, aux0=Cy, aux1=Ct, in_=o_y, in2=o_tap, out=o_out, scale=o_sc, shift=o_sh), ar)
    out = ar.read(o_out, (B, 2 * Hi, 2 * Wi, Cy + Ct), torch.float32)
    assert _rel(out, ref) < 2e-6


def test_nms_ties_golden(golden_dir):
    """Bit-exact against the reference's own CenterNetDetector.forward on a map with ties,
    plateaus, -inf and border maxima (tests/golden/g4_nms_ties.npz)."""
    g = np.load(os.path.join(golden_dir, "g4_nms_ties.npz"))
    maps = torch.from_numpy(g["maps"])                      # [2,9,16,20]
    B, _, h, w = maps.shape
    heat = torch.full((B, h, w, 10), 123.0)
    heat[..., 0] = maps[:, 0]
    heat[..., 2:] = maps[:, 1:].permute(0, 2, 3, 1)
    ar = Arena()
    o = ar.put(heat)
    ar.materialize()
    run_op(dict(kind=L.OP_NMS, B=B, H=h, W=w, Ho=h, Wo=w, Cout_total=10, out=o), ar)
    out = ar.read(o, (B, h, w, 10), torch.float32).permute(0, 3, 1, 2).numpy()
    assert np.array_equal(out, g["heatmap"])


@pytest.mark.parametrize("x3", [False, True], ids=["f32", "f32x3"])
@pytest.mark.parametrize("G,B,H,W,Cin,Cout", [(1, 2, 20, 12, 64, 1), (6, 1, 33, 17, 192, 1), (3, 2, 16, 48, 32, 2), (1, 1, 40, 24, 96, 4)],
                         ids=["1x1ch", "6heads", "3x2ch", "4ch"])
def test_thin_top_convolution(G, B, H, W, Cin, Cout, x3):
    """3x3 convolutions with 1..4 output channels on fp32 tensors (no activation) run on thin_conv3x3_kernel (vector units) instead of a
    padded MFMA tile: G heads with their own inputs / weights / biases writing consecutive channel slices of one [B,H,W,10] map;
    fp16x3 plans hand it pre-split weights."""
    g = torch.Generator().manual_seed(7 + Cin + Cout)
    x = torch.randn(G, B, H, W, Cin, generator=g)
    w = torch.randn(G, Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(G, Cout, generator=g) * 0.3
    CoutT, coff = 10, 1
    ref = torch.stack([F.conv2d(x[i].permute(0, 3, 1, 2), w[i], bias[i], 1, 1).permute(0, 2, 3, 1) for i in range(G)])
    wk = w.permute(0, 1, 3, 4, 2).reshape(G, Cout, 9, Cin)
    ar = Arena()
    o_in, o_b = ar.put(x), ar.put(bias)
    o_w = ar.put(presplit_f16x3(wk) if x3 else wk)
    o_out = ar.put(torch.full((B, H, W, CoutT), 7.0))
    ar.materialize()
    op = dict(kind=L.OP_CONV, flags=(L.FLAG_GROUP_OUT_SLICE if G > 1 else 0) | (L.FLAG_SPLIT16 if x3 else 0), act=L.ACT_NONE, in_dtype=L.F32, out_dtype=L.F32,
              w_dtype=L.F32, B=B, H=H, W=W, Ho=H, Wo=W, Cin=Cin, Cin_total=Cin, Cout=Cout, Cout_total=CoutT, cout_off=coff, ksize=3, stride=1,
              groups=G if G > 1 else 0, in_=o_in, out=o_out, w=o_w, bias=o_b)
    lib = L.load()
    buf = C.create_string_buffer(128)
    o = L.Op()
    for k_, v_ in op.items():
        if not isinstance(v_, tuple) and k_ not in ("in_", "out", "w", "bias"):
            setattr(o, k_, int(v_))
    lib.ftc_op_kernel_label(C.byref(o), buf, 128)
    assert buf.value.decode().startswith("thin_conv3x3<"), buf.value
    run_op(op, ar)
    full = ar.read(o_out, (B, H, W, CoutT), torch.float32)
    out = torch.stack([full[..., coff + i * Cout:coff + (i + 1) * Cout] for i in range(G)])
    err = _rel(out, ref)
    _log(f"thin conv G={G} Cin={Cin} Cout={Cout} x3={x3} rel_err {err:.3e}")
    assert err < (3e-6 if not x3 else 3e-5)                         # fp32 FMAs; the pre-split weights keep 22 bits
    assert float((full[..., 0] - 7.0).abs().max()) == 0.0 and float((full[..., coff + G * Cout:] - 7.0).abs().max()) == 0.0      # other channels untouched


@pytest.mark.parametrize("top", [False, True], ids=["plain", "top_fuse"])
@pytest.mark.parametrize("x3", [False, True], ids=["f32", "f32x3"])
@pytest.mark.parametrize("shape", [(2, 2, 16, 24, 192, 64), (1, 1, 22, 10, 64, 32), (9, 1, 32, 32, 192, 64), (8, 2, 32, 32, 192, 64), (3, 2, 24, 16, 192, 96)], ids=lambda s: "x".join(map(str, s)))
def test_upcat_in_conv_fp32_and_fp16x3(shape, x3, top):
    """FTC_FLAG_UPCAT_IN on fp32 tensors (round 3: the fp32 / fp16x3 plans' last FPN level): four fp32 channels per halo chunk, the
    fp16x3 halo written pre-split; against F.interpolate(align_corners=True) + cat + conv2d.  top_fuse (round 5): the same launch with the
    fp32 FMA form of FTC_FLAG_TOP_FUSE -- T = y . tap matrix instead of y -- followed by FTC_OP_TAPSUM, against a 3x3 top convolution of y."""
    G, B, H, W, Cy, Ct = shape
    g = torch.Generator().manual_seed(59)
    prev = torch.randn(G, B, H // 2, W // 2, Cy, generator=g)
    tap = torch.randn(G, B, H, W, Ct, generator=g)
    Cin, Cout = Cy + Ct, 192
    w = torch.randn(G, Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(G, Cout, generator=g) * 0.3
    ref = []
    for i in range(G):
        up = F.interpolate(prev[i].permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
        ref.append(F.gelu(F.conv2d(torch.cat([up, tap[i].permute(0, 3, 1, 2)], 1), w[i], bias[i], 1, 1)).permute(0, 2, 3, 1))
    ref = torch.stack(ref)
    wk = w.permute(0, 1, 3, 4, 2).reshape(G, Cout, 9, Cin)
    ar = Arena()
    o_prev, o_tap, o_b = ar.put(prev), ar.put(tap), ar.put(bias)
    o_w = ar.put(presplit_f16x3(wk) if x3 else wk)
    flags = L.FLAG_UPCAT_IN | (L.FLAG_SPLIT16 if x3 else 0)
    common = dict(kind=L.OP_CONV, act=L.ACT_GELU, in_dtype=L.F32, out_dtype=L.F32, w_dtype=L.F32, B=B, H=H, W=W, Ho=H, Wo=W, Cin=Cin, Cin_total=Cy, Cout=Cout,
                  Cout_total=Cout, ksize=3, stride=1, aux0=65, groups=G if G > 1 else 0)
    if not top:
        o_out = ar.reserve(G * B * H * W * Cout * 4)
        ar.materialize()
        run_op(dict(common, flags=flags, in_=o_prev, in2=o_tap, out=o_out, w=o_w, bias=o_b), ar)
        out = ar.read(o_out, (G, B, H, W, Cout), torch.float32)
        err = _rel(out, ref)
        _log(f"upcat_in fp32 conv {shape} x3={x3} rel_err {err:.3e}")
        assert err < 2e-5
        return
    if Cy % 64 or G > 10:
        pytest.skip("TOP_FUSE: one 192-channel tile per head, at most 10 one-channel heads in this test")
    TW = 12                                                  # one output channel per head: 9 taps, padded to 12 floats per pixel
    wt = torch.randn(G, 1, Cout, 3, 3, generator=g) / (Cout * 9) ** 0.5
    bt = torch.randn(G, generator=g) * 0.2
    want = torch.stack([F.conv2d(ref[i].permute(0, 3, 1, 2).double(), wt[i].double(), bt[i:i + 1].double(), 1, 1)[:, 0] for i in range(G)], -1).float()    # [B,H,W,G]
    wt_mat = torch.zeros(G, 32, Cout)
    for i in range(G):
        for t9 in range(9):
            wt_mat[i, t9] = wt[i, 0, :, t9 // 3, t9 % 3]
    omap = torch.tensor([(i, 0, 1, i) for i in range(G)], dtype=torch.int32)
    o_wt, o_map, o_ob = ar.put(wt_mat), ar.put(omap), ar.put(bt)
    o_T = ar.reserve(G * B * H * W * TW * 4)
    o_out = ar.reserve(B * H * W * 10 * 4)
    ar.materialize()
    run_op(dict(common, flags=flags | L.FLAG_TOP_FUSE, aux1=TW, in_=o_prev, in2=o_tap, out=o_T, w=o_w, bias=o_b, w2=o_wt), ar)
    run_op(dict(kind=L.OP_TAPSUM, B=B, H=H, W=W, Ho=H, Wo=W, Cout_total=10, aux0=TW, aux1=G, groups=G, in_=o_T, out=o_out, w=o_map, bias=o_ob), ar)
    out = ar.read(o_out, (B, H, W, 10), torch.float32)[..., :G]
    err = _rel(out, want)
    _log(f"upcat_in + top_fuse fp32 conv {shape} x3={x3} rel_err {err:.3e}")
    assert err < 2e-5


@pytest.mark.parametrize("dt", [L.BF16, L.F16, 3], ids=["bf16", "f16", "f16x3"])
@pytest.mark.parametrize("shape", [(2, 24, 24, 64, 256, 64, True), (1, 19, 13, 96, 384, 96, True), (3, 16, 16, 32, 256, 32, True), (1, 12, 20, 64, 384, 128, False),
                                   (2, 33, 7, 96, 256, 64, True), (8, 48, 48, 64, 256, 64, True)],
                         ids=["24x24_64_256", "19x13_96_384_ragged", "16x16_32_256", "12x20_64_384_to128_nores", "33x7_96_256", "b8_48x48_64_256"])
def test_fused_mbconv_block_in_one_launch(shape, dt):
    """FTC_OP_FMBCONV (csrc/fused_mbconv.hip, round 6): 3x3 expand + BN + SiLU -> 1x1 project + BN + residual of a Fused-MBConv block
    (torchvision FusedMBConv as instantiated by /root/reference/models/detector.py:14-16) in ONE launch -- against the same chain on the CPU
    (fp32 with the expanded tensor rounded to the 16-bit type, i.e. what the two-launch form stores; float64 for the fp16x3 form, whose fp32
    tensors are held to 2e-5); and against that two-launch form itself run on the GPU through FTC_OP_CONV (same rounding points and K order:
    bit-identical).  Pixel counts that are not multiples of the 128-pixel tile, maps narrower than the tile, both K steps."""
    B, H, W, Cin, E, Cout, has_res = shape
    x3 = dt == 3
    if x3 and E != 256:
        pytest.skip("the fp16x3 form holds E = 256 (its activated tile is 128 KB of LDS)")
    sdt = L.F32 if x3 else dt
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + E)
    r16 = lambda t: round16(t, sdt)
    x = r16(torch.randn(B, H, W, Cin, generator=g))
    w1 = r16(torch.randn(E, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5 * 1.5)
    b1 = torch.randn(E, generator=g) * 0.3
    w2 = r16(torch.randn(Cout, E, generator=g) / E ** 0.5)
    b2 = torch.randn(Cout, generator=g) * 0.3
    res = torch.randn(B, H, W, Cout, generator=g) if has_res else None
    cd = torch.float64 if x3 else torch.float32
    e = r16(F.silu(F.conv2d(x.to(cd).permute(0, 3, 1, 2), w1.to(cd), b1.to(cd), 1, 1)).float()).permute(0, 2, 3, 1) if not x3 else \
        F.silu(F.conv2d(x.to(cd).permute(0, 3, 1, 2), w1.to(cd), b1.to(cd), 1, 1)).permute(0, 2, 3, 1)
    ref = e.reshape(-1, E).to(cd) @ w2.to(cd).t() + b2.to(cd)
    if has_res:
        ref = ref + res.reshape(-1, Cout).to(cd)
    ref = ref.reshape(B, H, W, Cout).float()
    ar = Arena()
    o_x = ar.put(to_dev_bytes(x, sdt))
    w1k = w1.permute(0, 2, 3, 1).contiguous()                                                            # [E][9][Cin]
    o_w1 = ar.put(presplit_f16x3(w1k) if x3 else to_dev_bytes(w1k, sdt))
    o_b1, o_w2, o_b2 = ar.put(b1), ar.put(presplit_f16x3(w2) if x3 else to_dev_bytes(w2, sdt)), ar.put(b2)
    o_res = ar.put(res) if has_res else None
    esz = 4 if x3 else 2
    o_out, o_out2 = ar.reserve(B * H * W * Cout * 4), ar.reserve(B * H * W * Cout * esz)
    o_e, o_out_b = ar.reserve(B * H * W * E * esz), ar.reserve(B * H * W * Cout * 4)
    ar.materialize()
    fl = (L.FLAG_RESIDUAL if has_res else 0) | (L.FLAG_SPLIT16 if x3 else 0)
    run_op(dict(kind=L.OP_FMBCONV, flags=fl, act=L.ACT_SILU, in_dtype=sdt, out_dtype=L.F32, w_dtype=sdt, res_dtype=L.F32, B=B, H=H, W=W,
                Ho=H, Wo=W, Cin=Cin, Cin_total=Cin, Cout=Cout, Cout_total=Cout, ksize=3, stride=1, aux1=E, in_=o_x, in2=o_res, w2=o_w1, bias2=o_b1, w=o_w2, bias=o_b2,
                out=o_out, out2=o_out2), ar)
    out = ar.read(o_out, (B, H, W, Cout), torch.float32)
    err = _rel(out, ref)
    # the two-launch form on the GPU
    run_op(dict(kind=L.OP_CONV, flags=L.FLAG_SPLIT16 if x3 else 0, act=L.ACT_SILU, in_dtype=sdt, out_dtype=sdt, w_dtype=sdt, B=B, H=H, W=W, Ho=H, Wo=W, Cin=Cin, Cin_total=Cin,
                Cout=E, Cout_total=E, ksize=3, stride=1, in_=o_x, w=o_w1, bias=o_b1, out=o_e), ar)
    run_op(dict(kind=L.OP_CONV, flags=fl, act=L.ACT_NONE, in_dtype=sdt, out_dtype=L.F32, w_dtype=sdt, res_dtype=L.F32, B=B, H=H, W=W, Ho=H,
                Wo=W, Cin=E, Cin_total=E, Cout=Cout, Cout_total=Cout, ksize=1, stride=1, in_=o_e, in2=o_res, w=o_w2, bias=o_b2, out=o_out_b), ar)
    two = ar.read(o_out_b, (B, H, W, Cout), torch.float32)
    err_two = _rel(out, two)
    _log(f"fmbconv {shape} dt={dt}: rel_err vs CPU chain {err:.3e}, vs the two-launch form on the GPU {err_two:.3e}")
    # an element of e whose fp32 pre-image sits within summation noise of a rounding boundary lands one 16-bit ulp away: 2^-8 (bf16) / 2^-11 (fp16)
    # of one of E terms -- the same budget as the MBHEAD test, which carries the same kind of intermediate
    lim = 2e-5 if x3 else 6e-3 if dt == L.BF16 else 1e-3
    assert err < lim and err_two < lim
    if x3:
        raw = ar.buf[o_out2:o_out2 + B * H * W * Cout * 4].cpu()
        assert float((_unsplit_f16x3(raw, (B, H, W, Cout)) - out).abs().max()) <= 3e-6 * float(out.abs().max())
    else:
        out2 = ar.read(o_out2, (B, H, W, Cout), tdtype(dt)).float()
        assert float((out2 - round16(out, dt)).abs().max()) == 0.0                   # the 16-bit copy IS the rounded fp32 output
    # nothing outside the outputs was written
    tail = ar.buf[ar.size:ar.size + 256]
    assert bool((tail == 0xCD).all())


@pytest.mark.parametrize("dt", [L.BF16, L.F16, 3], ids=["bf16", "f16", "f16x3"])
@pytest.mark.parametrize("shape", [(2, 32, 48, True, L.ACT_SILU), (1, 21, 19, True, L.ACT_SILU), (3, 16, 16, False, L.ACT_NONE), (1, 7, 50, True, L.ACT_GELU), (8, 96, 96, True, L.ACT_SILU)],
                         ids=["32x48", "21x19_ragged", "16x16_nores_noact", "7x50", "b8_96x96"])
def test_conv3x3_c32_resident_kernel(shape, dt):
    """csrc/conv3x3_c32.hip (round 6): the 32 -> 32 channel 3x3 convolution of the stage-1 Fused-MBConv blocks (/root/reference/models/detector.py:14) with halo AND
    weights resident in LDS -- against PyTorch on the CPU (float64 for the fp16x3 form: fp32 tensors, three fp16 MFMAs per product, held to 2e-5), and against the
    implicit-GEMM kernel on the same data (reached by presenting the input as a 32-channel slice of a 64-channel buffer, which the resident kernel does not take):
    same K order, bit-identical.  Ragged maps, maps smaller than a tile."""
    B, H, W, has_res, act = shape
    x3 = dt == 3
    sdt = L.F32 if x3 else dt                                                   # storage / compute type of the op
    g = torch.Generator().manual_seed(B * 100 + H)
    x = round16(torch.randn(B, H, W, 32, generator=g), sdt)
    w = round16(torch.randn(32, 32, 3, 3, generator=g) / (9 * 32) ** 0.5 * 1.5, sdt)
    bias = torch.randn(32, generator=g) * 0.3
    res = torch.randn(B, H, W, 32, generator=g) if has_res else None
    actf = {L.ACT_NONE: lambda v: v, L.ACT_SILU: F.silu, L.ACT_GELU: F.gelu}[act]
    ref = actf(F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), 1, 1)).permute(0, 2, 3, 1)
    if has_res:
        ref = ref + res.double()
    ref = ref.float()
    ar = Arena()
    o_x = ar.put(to_dev_bytes(x, sdt))
    xw = torch.zeros(B, H, W, 64)
    xw[..., :32] = x
    o_xw = ar.put(to_dev_bytes(xw, sdt))
    wk = w.permute(0, 2, 3, 1).contiguous()                                     # [Cout][9][Cin]
    o_w, o_b = ar.put(presplit_f16x3(wk) if x3 else to_dev_bytes(wk, sdt)), ar.put(bias)
    o_res = ar.put(res) if has_res else None
    o_out, o_out2, o_gen = ar.reserve(B * H * W * 32 * 4), ar.reserve(B * H * W * 32 * 4), ar.reserve(B * H * W * 32 * 4)
    ar.materialize()
    common = dict(kind=L.OP_CONV, flags=(L.FLAG_RESIDUAL if has_res else 0) | (L.FLAG_SPLIT16 if x3 else 0), act=act, in_dtype=sdt, out_dtype=L.F32, w_dtype=sdt,
                  res_dtype=L.F32, B=B, H=H, W=W, Ho=H, Wo=W, Cin=32, Cout=32, Cout_total=32, ksize=3, stride=1, in2=o_res, w=o_w, bias=o_b)
    lib = L.load()
    op = (L.Op * 1)()
    for k, v in dict(common, Cin_total=32).items():
        if k not in ("in2", "w", "bias"):
            setattr(op[0], k, int(v))
    buf = C.create_string_buffer(128)
    lib.ftc_op_kernel_label(C.byref(op[0]), buf, 128)
    assert buf.value.decode().startswith("conv3x3_c32<" + ("f16x3" if x3 else "")), buf.value      # the shape of the plan's stage-1 ops takes the resident kernel
    run_op(dict(common, Cin_total=32, in_=o_x, out=o_out, out2=o_out2), ar)
    run_op(dict(common, Cin_total=64, in_=o_xw, out=o_gen), ar)                          # the implicit-GEMM kernel (channel slice of a wider buffer)
    out, gen = ar.read(o_out, (B, H, W, 32), torch.float32), ar.read(o_gen, (B, H, W, 32), torch.float32)
    err = _rel(out, ref)
    _log(f"conv3x3_c32 {shape} dt={dt}: rel_err vs CPU {err:.3e}, max |resident - implicit GEMM| {float((out - gen).abs().max()):.3e}")
    assert err < (2e-5 if x3 else 3e-3 if dt == L.BF16 else 5e-4)
    assert torch.equal(out, gen)
    if x3:                                                                               # out2 = the PRE-SPLIT copy: [hi x4 | lo x4] halves per 16-byte chunk, hi + lo == out to 2^-22
        raw = ar.buf[o_out2:o_out2 + B * H * W * 32 * 4].cpu()
        assert float((_unsplit_f16x3(raw, (B, H, W, 32)) - out).abs().max()) <= 3e-6 * float(out.abs().max())
    else:
        out2 = ar.read(o_out2, (B, H, W, 32), tdtype(dt)).float()
        assert float((out2 - round16(out, dt)).abs().max()) == 0.0
    assert bool((ar.buf[ar.size:ar.size + 256] == 0xCD).all())
