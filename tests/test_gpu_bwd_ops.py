"""-m gpu: every BACKWARD kernel of the train step (csrc/bwd_ops.hip, csrc/wgrad.hip), called through the C ABI as a single-op plan,
against torch autograd of the same op on the CPU in fp32 -- the per-kernel floating-point reference; the end-to-end comparison with
the reference's own ``loss.backward()`` is tests/test_gpu_train_step.py.

Tolerances: fp32 kernels 2e-5 .. 2e-4 of max|ref| (summation order); 16-bit MFMA operands are compared with the CPU result on the same
ROUNDED operands (1.5e-2 bf16 / 2.5e-3 fp16).  Parameter-gradient outputs ACCUMULATE: every test pre-loads them with a known value.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from findtextcenternet_amd import _lib as L
from gpu_harness import Arena, round16, run_op

pytestmark = pytest.mark.gpu
ACT = {L.ACT_NONE: lambda v: v, L.ACT_SILU: F.silu, L.ACT_GELU: F.gelu}
TOL16 = {L.F32: 2e-4, L.BF16: 1.5e-2, L.F16: 2.5e-3}


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _fbits(v):
    import struct
    return struct.unpack("<i", struct.pack("<f", v))[0]


def _chunks(M, cap=512):
    return max(1, min(cap, -(-M // 64)))


@pytest.mark.parametrize("act", [L.ACT_NONE, L.ACT_SILU, L.ACT_GELU], ids=["none", "silu", "gelu"])
@pytest.mark.parametrize("variant", ["plain", "keep", "gate", "slice_accum"])
@pytest.mark.parametrize("shape", [(2, 6, 5, 96), (3, 9, 7, 24), (2, 16, 16, 256)], ids=["6x5x96", "9x7x24", "16x16x256"])
@pytest.mark.parametrize("wd", [L.F32, L.BF16], ids=["precise", "fast"])
def test_bnbwd(shape, variant, act, wd):
    """FTC_OP_BNBWD vs autograd of act(batch_norm(z, training=True)) (* keep[b]) with the incoming gradient optionally gated per
    (image, channel) (the MBConv SE path), taken from a channel slice, accumulated into the output."""
    B, H, W, C = shape
    g = torch.Generator().manual_seed(C + B)
    z = torch.randn(B, H, W, C, generator=g) * 1.3 + 0.2
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    eps = 1e-3
    M = B * H * W
    Ct, off = (C + 32, 32) if variant == "slice_accum" else (C, 0)
    gy_full = torch.randn(B, H, W, Ct, generator=g)
    gy = gy_full[..., off:off + C]
    keep = torch.tensor([0.0 if b % 2 else 1.25 for b in range(B)]) if variant == "keep" else None
    ga = torch.rand(B, C, generator=g) if variant == "gate" else None
    gb = torch.randn(B, C, generator=g) * 0.1 if variant == "gate" else None
    zz = z.clone().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = ACT[act](F.batch_norm(zz.permute(0, 3, 1, 2), None, None, gm, bt, True, 0.1, eps)).permute(0, 2, 3, 1)
    if keep is not None:
        y = y * keep[:, None, None, None]
    gin = gy if ga is None else gy * ga[:, None, None, :] + gb[:, None, None, :]
    y.backward(gin)
    mean = z.reshape(M, C).double().mean(0)
    var = z.reshape(M, C).double().var(0, unbiased=False)
    istd = 1.0 / torch.sqrt(var + eps)
    ss = torch.stack([gamma.double() * istd, beta.double() - mean * gamma.double() * istd, mean, istd]).float()
    pre = torch.randn(B, H, W, C, generator=g)
    ar = Arena()
    o_gy, o_z, o_ss = ar.put(gy_full), ar.put(z), ar.put(ss)
    o_keep = ar.put(keep) if keep is not None else None
    o_ga = ar.put(ga) if ga is not None else None
    o_gb = ar.put(gb) if gb is not None else None
    o_out = ar.put(pre)
    o_gg, o_gbeta = ar.put(torch.full((C,), 0.5)), ar.put(torch.full((C,), -0.25))
    o_aux = ar.reserve(_chunks(M) * 2 * C * 8 + 2 * C * 4)
    ar.materialize()
    run_op(dict(kind=L.OP_BNBWD, flags=L.FLAG_ACCUM if variant == "slice_accum" else 0, act=act, w_dtype=wd, B=B, H=H, W=W, Cin=C,
                Cin_total=Ct if off else 0, cin_off=off, in_=o_gy, in2=o_z, scale=o_ss, w2=o_keep, bias=o_ga, bias2=o_gb, out=o_out, w=o_gg,
                shift=o_gbeta, aux=o_aux), ar)
    dz = ar.read(o_out, (B, H, W, C), torch.float32)
    want = zz.grad + (pre if variant == "slice_accum" else 0)
    tol = 3e-5 if wd == L.F32 else 2e-4                               # "fast": exp2 / rcp activations derivatives (1.5e-7 absolute erf)
    assert _rel(dz, want) < tol
    assert _rel(ar.read(o_gg, (C,), torch.float32) - 0.5, gm.grad) < tol
    assert _rel(ar.read(o_gbeta, (C,), torch.float32) + 0.25, bt.grad) < tol


WG_CASES = [  # name, B,H,W, Cin,CinT,cin_off, Cout,CoutT,cout_off, k, stride, se
    ("pw_192_768", 2, 12, 12, 192, 192, 0, 768, 768, 0, 1, 1, False),
    ("pw_project_se", 3, 6, 6, 1536, 1536, 0, 256, 256, 0, 1, 1, True),
    ("c3_32_32", 1, 40, 36, 32, 32, 0, 32, 32, 0, 3, 1, False),
    ("c3_s2_64_256", 2, 16, 20, 64, 64, 0, 256, 256, 0, 3, 2, False),
    ("c3_fpn_288_192", 1, 24, 24, 288, 288, 0, 192, 192, 0, 3, 1, False),
    ("top_1_of_9", 2, 16, 24, 192, 192, 0, 1, 9, 4, 3, 1, False),
    ("top_2_of_9", 1, 16, 16, 192, 192, 0, 2, 9, 1, 3, 1, False),
    ("top_100_of_128", 1, 12, 12, 192, 192, 0, 100, 128, 0, 3, 1, False),
    ("linear_100_of_128", 1, 700, 1, 100, 128, 0, 2048, 2048, 0, 1, 1, False),
    ("linear_logits", 1, 300, 1, 256, 256, 0, 1091, 1104, 0, 1, 1, False),
    ("odd_hw", 1, 15, 17, 24, 24, 0, 48, 48, 0, 3, 1, False),
    # Wo % 32 == 0: with 16-bit copies of both operands these run on the nine-taps-in-one-workgroup kernel (wgrad3_kernel)
    ("c3n_64_256_w64", 2, 6, 64, 64, 64, 0, 256, 256, 0, 3, 1, False),
    ("c3n_288_192_w32", 1, 5, 32, 288, 288, 0, 192, 192, 0, 3, 1, False),
    ("c3n_slices_w96", 1, 4, 96, 40, 72, 32, 100, 128, 8, 3, 1, False),
    ("c3n_many_rows", 3, 33, 32, 32, 32, 0, 64, 64, 0, 3, 1, False),
    # one or two output channels: the thin (column-sum) kernel when the output gradient is fp32
    ("thin_1_of_9", 2, 12, 16, 192, 192, 0, 1, 9, 3, 3, 1, False),
    ("thin_2_of_9", 3, 9, 11, 64, 72, 8, 2, 9, 5, 3, 1, False),
    ("thin_1x1", 2, 40, 40, 96, 96, 0, 2, 2, 0, 1, 1, False),
    # SE-gated input with one or two splits per image: the gate multiplies the partial tile's columns instead of the staged elements
    ("se_image_splits", 4, 16, 16, 96, 96, 0, 40, 40, 0, 1, 1, True),
    ("se_image_splits_slices", 3, 16, 8, 72, 80, 8, 136, 136, 0, 1, 1, True),
]


@pytest.mark.parametrize("wd,io", [(L.F32, "f32"), (L.BF16, "f32"), (L.F16, "f32"), (L.BF16, "x16"), (L.BF16, "x16d16"), (L.F16, "x16d16"), (L.F16, "d16")],
                         ids=["f32", "bf16", "f16", "bf16_x16", "bf16_x16d16", "f16_x16d16", "f16_d16"])
@pytest.mark.parametrize("case", WG_CASES, ids=[c[0] for c in WG_CASES])
def test_wgrad(case, wd, io):
    """FTC_OP_WGRAD vs autograd of F.conv2d with respect to the weight, accumulated into a pre-loaded OIHW gradient; operands stored
    fp32 or as the 16-bit copies the BatchNorm passes of the train step write (in_dtype: layer input, res_dtype: output gradient)."""
    from gpu_harness import to_dev_bytes
    xdt = wd if "x16" in io else L.F32
    ddt = wd if "d16" in io else L.F32
    _, B, H, W, Cin, CinT, cio, Cout, CoutT, coo, k, stride, se = case
    g = torch.Generator().manual_seed(Cin + Cout)
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x_full = torch.randn(B, H, W, CinT, generator=g)
    dz_full = torch.randn(B, Ho, Wo, CoutT, generator=g)
    sc = torch.rand(B, Cin, generator=g) + 0.25 if se else None
    x = x_full[..., cio:cio + Cin]
    dz = dz_full[..., coo:coo + Cout]
    if xdt != L.F32:
        x = round16(x, xdt)
    xe = x * sc[:, None, None, :] if se else x
    w = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    F.conv2d(round16(xe, wd).permute(0, 3, 1, 2), w, None, stride, pad).backward(round16(dz, wd).permute(0, 3, 1, 2))
    lib = L.load()
    S = int(lib.ftc_wgrad_splits(B, Ho, Wo, Cout, Cin, k))
    if case[0].startswith("se_image_splits"):                  # (what TrainStep._G.wgrad asks for when the input is gated)
        S = B * (2 if (Ho * Wo) % 128 == 0 and case[0].endswith("slices") is False else 1)
        assert (Ho * Wo) % (S // B * 64) == 0
    pre = torch.randn(Cout, Cin, k, k, generator=g)
    ar = Arena()
    o_x, o_dz, o_out = ar.put(to_dev_bytes(x_full, xdt)), ar.put(to_dev_bytes(dz_full, ddt)), ar.put(pre)
    o_sc = ar.put(sc) if se else None
    o_aux = ar.reserve(S * k * k * Cout * Cin * 4)
    ar.materialize()
    run_op(dict(kind=L.OP_WGRAD, flags=L.FLAG_SE_SCALE if se else 0, w_dtype=wd, in_dtype=xdt, res_dtype=ddt, B=B, H=H, W=W, Ho=Ho, Wo=Wo, Cin=Cin, Cin_total=CinT, cin_off=cio,
                Cout=Cout, Cout_total=CoutT, cout_off=coo, ksize=k, stride=stride, aux0=S, in_=o_x, in2=o_dz, scale=o_sc, out=o_out, aux=o_aux), ar)
    got = ar.read(o_out, (Cout, Cin, k, k), torch.float32) - pre
    # (with SE the kernel rounds x * s, the reference rounds the same product)
    assert _rel(got, w.grad) < TOL16[wd], (case[0], S)


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("shape", [(2, 12, 10, 96), (1, 9, 9, 24), (3, 8, 8, 384)], ids=["12x10x96", "9x9x24", "8x8x384"])
def test_dwbwd(shape, stride):
    B, H, W, C = shape
    if stride == 2 and (H % 2 or W % 2):
        pytest.skip("the network's stride-2 depthwise layers see even sizes")
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, H, W, C, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dz = torch.randn(B, Ho, Wo, C, generator=g)
    xx, ww = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv2d(xx.permute(0, 3, 1, 2), ww, None, stride, 1, 1, C).backward(dz.permute(0, 3, 1, 2))
    pre = torch.randn(C, 1, 3, 3, generator=g)
    ar = Arena()
    o_x, o_dz, o_w = ar.put(x), ar.put(dz), ar.put(w.reshape(C, 9).t().contiguous())
    o_out, o_gw = ar.reserve(B * H * W * C * 4), ar.put(pre)
    o_aux = ar.reserve(_chunks(B * Ho * Wo) * 9 * C * 8)
    ar.materialize()
    run_op(dict(kind=L.OP_DWBWD, B=B, H=H, W=W, Ho=Ho, Wo=Wo, Cin=C, stride=stride, in_=o_x, in2=o_dz, w=o_w, out=o_out, out2=o_gw, aux=o_aux), ar)
    assert _rel(ar.read(o_out, (B, H, W, C), torch.float32), xx.grad) < 2e-5
    assert _rel(ar.read(o_gw, (C, 1, 3, 3), torch.float32) - pre, ww.grad) < 2e-5


@pytest.mark.parametrize("shape", [(2, 6, 6, 384, 24, 3), (3, 12, 12, 768, 48, 2), (8, 4, 4, 3072, 128, 1)], ids=["384_24", "768_48", "3072_128"])
def test_sebwd(shape):
    """FTC_OP_SEBWD vs autograd of y * sigmoid(fc2(silu(fc1(mean_hw y)))): the four parameter gradients, and the two per-(image,
    channel) operands it hands to the following FTC_OP_BNBWD (gate s and d mean / HW), checked through d y = g * s + d mean / HW."""
    B, H, W, C, S, P = shape
    g = torch.Generator().manual_seed(C)
    y = torch.randn(B, H, W, C, generator=g)
    w1, b1 = torch.randn(S, C, generator=g) * 0.05, torch.randn(S, generator=g) * 0.1
    w2, b2 = torch.randn(C, S, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    gout = torch.randn(B, H, W, C, generator=g)
    yy = y.clone().requires_grad_(True)
    ps = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    m = yy.mean((1, 2))
    s = torch.sigmoid(F.linear(F.silu(F.linear(m, ps[0], ps[1])), ps[2], ps[3]))
    (yy * s[:, None, None, :]).backward(gout)
    HW = H * W
    rows = -(-HW // P)
    sums = torch.stack([y.reshape(B, HW, C)[:, p * rows:(p + 1) * rows].sum(1) for p in range(P)], 1)      # forward partial sums [B][P][C]
    pre = torch.randn(2 * S * C + S + C, generator=g)
    ar = Arena()
    o_g, o_y, o_s, o_sums = ar.put(gout), ar.put(y), ar.put(s.detach()), ar.put(sums)
    o_w1, o_w2t, o_b1, o_b2 = ar.put(w1), ar.put(w2.t().contiguous()), ar.put(b1), ar.put(b2)
    o_scr = ar.reserve((36 * B * C + 2 * B * S) * 4)
    o_grads = ar.put(pre)
    ar.materialize()
    run_op(dict(kind=L.OP_SEBWD, B=B, H=H, W=W, Cin=C, aux0=S, aux1=P, in_=o_g, in2=o_y, scale=o_s, aux=o_sums, w=o_w1, w2=o_w2t, bias=o_b1, bias2=o_b2,
                out=o_scr, out2=o_grads), ar)
    got = ar.read(o_grads, (2 * S * C + S + C,), torch.float32) - pre
    want = torch.cat([ps[0].grad.reshape(-1), ps[1].grad, ps[2].grad.reshape(-1), ps[3].grad])
    for name, lo, hi in (("fc1.w", 0, S * C), ("fc1.b", S * C, S * C + S), ("fc2.w", S * C + S, 2 * S * C + S), ("fc2.b", 2 * S * C + S, 2 * S * C + S + C)):
        assert _rel(got[lo:hi], want[lo:hi]) < 5e-5, name
    gbv = ar.read(o_scr + 3 * B * C * 4, (B, C), torch.float32)
    dy = gout * s.detach()[:, None, None, :] + gbv[:, None, None, :]
    assert _rel(dy, yy.grad) < 5e-5


@pytest.mark.parametrize("shape", [(2, 6, 5, 192, 96), (1, 12, 12, 64, 32), (3, 4, 4, 192, 1280)], ids=["6x5", "12x12", "4x4"])
def test_upcatbwd(shape):
    B, Hi, Wi, Cy, Ct = shape
    g = torch.Generator().manual_seed(Cy + Hi)
    y = torch.randn(B, Cy, Hi, Wi, generator=g, requires_grad=True)
    gcat = torch.randn(B, 2 * Hi, 2 * Wi, Cy + Ct, generator=g)
    F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True).backward(gcat[..., :Cy].permute(0, 3, 1, 2))
    ar = Arena()
    o_g, o_out = ar.put(gcat), ar.reserve(B * Hi * Wi * Cy * 4)
    ar.materialize()
    run_op(dict(kind=L.OP_UPCATBWD, B=B, H=Hi, W=Wi, Ho=2 * Hi, Wo=2 * Wi, Cin_total=Cy + Ct, aux0=Cy, in_=o_g, out=o_out), ar)
    assert _rel(ar.read(o_out, (B, Hi, Wi, Cy), torch.float32), y.grad.permute(0, 2, 3, 1)) < 2e-5


@pytest.mark.parametrize("wd", [L.F32, L.BF16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 16, 16, 32, 128, 3), (1, 12, 20, 64, 256, 3)], ids=["32_128", "64_256"])
def test_stride2_data_gradient_as_dilate_plus_flipped_conv(case, wd):
    """d input of a stride-2 3x3 convolution = FTC_OP_DILATE + the forward implicit-GEMM kernel on the flipped / transposed weights that
    ftc_pack_train_weights writes -- both through the ABI, against autograd."""
    B, H, W, Cin, Cout, k = case
    g = torch.Generator().manual_seed(Cin)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    Ho, Wo = H // 2, W // 2
    dz = torch.randn(B, Ho, Wo, Cout, generator=g)
    F.conv2d(x, round16(w, wd), None, 2, 1).backward(round16(dz, wd).permute(0, 3, 1, 2))
    lib = L.load()
    esz = 4 if wd == L.F32 else 2
    wdev = w.cuda().contiguous()
    dg = torch.zeros(Cin * 9 * Cout * esz, dtype=torch.uint8, device="cuda")
    ent = (L.PackEntry * 1)()
    ent[0].src, ent[0].fwd, ent[0].dgrad = wdev.data_ptr(), None, dg.data_ptr()
    ent[0].Cout, ent[0].Cin, ent[0].kk, ent[0].cin_pad, ent[0].cout_pad, ent[0].dtype = Cout, Cin, 9, Cin, Cout, wd
    ent_dev = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).cuda()
    L.check(lib.ftc_pack_train_weights(ent_dev.data_ptr(), 1, Cout * 9 * Cin * 2, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pack")
    torch.cuda.synchronize()
    ar = Arena()
    o_dz, o_dil, o_dx = ar.put(dz), ar.reserve(B * H * W * Cout * 4), ar.reserve(B * H * W * Cin * 4)
    o_w, o_zero = ar.put(dg.cpu()), ar.put(torch.zeros(Cin))
    ar.materialize()
    run_op(dict(kind=L.OP_DILATE, B=B, H=Ho, W=Wo, Ho=H, Wo=W, Cin=Cout, in_=o_dz, out=o_dil), ar)
    run_op(dict(kind=L.OP_CONV, act=L.ACT_NONE, in_dtype=L.F32, out_dtype=L.F32, w_dtype=wd, B=B, H=H, W=W, Ho=H, Wo=W, Cin=Cout, Cin_total=Cout, Cout=Cin,
                Cout_total=Cin, ksize=3, stride=1, res_dtype=L.F32, in_=o_dil, out=o_dx, w=o_w, bias=o_zero), ar)
    assert _rel(ar.read(o_dx, (B, H, W, Cin), torch.float32), x.grad.permute(0, 2, 3, 1)) < TOL16[wd]


@pytest.mark.parametrize("co,off", [(1, 0), (2, 1), (1, 8)])
def test_topdgrad_and_colsum(co, off):
    B, H, W, Ci, CoT = 2, 14, 10, 192, 9
    g = torch.Generator().manual_seed(co + off)
    y = torch.randn(B, Ci, H, W, generator=g, requires_grad=True)
    w = torch.randn(co, Ci, 3, 3, generator=g) * 0.1
    gm = torch.randn(B, H, W, CoT, generator=g)
    F.conv2d(y, w, None, 1, 1).backward(gm[..., off:off + co].permute(0, 3, 1, 2))
    ar = Arena()
    o_g, o_w = ar.put(gm), ar.put(w.permute(0, 2, 3, 1).reshape(co, 9, Ci).contiguous())
    o_out, o_b = ar.reserve(B * H * W * Ci * 4), ar.put(torch.full((co,), 2.0))
    o_aux = ar.reserve(_chunks(B * H * W) * co * 8)
    ar.materialize()
    run_op(dict(kind=L.OP_TOPDGRAD, w_dtype=L.F32, B=B, H=H, W=W, Cin=co, Cin_total=CoT, cin_off=off, Cout=Ci, in_=o_g, w=o_w, out=o_out), ar)
    run_op(dict(kind=L.OP_COLSUM, B=B, H=H, W=W, Cin=co, Cin_total=CoT, cin_off=off, in_=o_g, out=o_b, aux=o_aux), ar)
    assert _rel(ar.read(o_out, (B, H, W, Ci), torch.float32), y.grad.permute(0, 2, 3, 1)) < 2e-5
    assert _rel(ar.read(o_b, (co,), torch.float32) - 2.0, gm[..., off:off + co].sum((0, 1, 2))) < 2e-5


@pytest.mark.parametrize("C0", [32, 24])
def test_stemwgrad(C0):
    B, H, W = 2, 20, 28
    g = torch.Generator().manual_seed(C0)
    img = torch.rand(B, H, W, 3, generator=g)
    dz = torch.randn(B, H // 2, W // 2, C0, generator=g)
    w = torch.zeros(C0, 3, 3, 3, requires_grad=True)
    F.conv2d((img * 2 - 1).permute(0, 3, 1, 2), w, None, 2, 1).backward(dz.permute(0, 3, 1, 2))
    pre = torch.randn(C0, 3, 3, 3, generator=g)
    ar = Arena()
    o_img, o_dz, o_out = ar.put(img), ar.put(dz), ar.put(pre)
    o_aux = ar.reserve(max(1, min(2048, -(-(B * (H // 2) * (W // 2)) // 256))) * 27 * C0 * 8)
    ar.materialize()
    run_op(dict(kind=L.OP_STEMWGRAD, B=B, H=H, W=W, Ho=H // 2, Wo=W // 2, Cout=C0, in_=o_img, in2=o_dz, out=o_out, aux=o_aux), ar)
    assert _rel(ar.read(o_out, (C0, 3, 3, 3), torch.float32) - pre, w.grad) < 2e-5


def test_losses_and_loss_bwd_match_autograd_of_the_oracle():
    """FTC_OP_LOSSES + FTC_OP_LOSS_BWD vs autograd through oracle/loss_oracle.py (pinned by g7 against the reference's loss_func.py):
    d(sum alpha_i loss_i * scale) / d maps and / d decoder logits, with gather / scatter of the selected rows around it."""
    import synth
    from oracle import loss_oracle
    B, h, w = 2, 64, 64
    label, idmap = synth.train_labels(77, B, h, w)
    lab_t, id_t = torch.from_numpy(label), torch.from_numpy(idmap).long()
    fmask = loss_oracle.get_fmask(lab_t)
    n = int(fmask.sum())
    tgt = id_t[:, 0].flatten()[fmask].numpy()
    hm, dec = synth.loss_case(78, B, h, w, tgt, (1091, 1093, 1097))
    hm_t = torch.from_numpy(hm).clone().requires_grad_(True)
    dec_t = [torch.from_numpy(d).clone().requires_grad_(True) for d in dec]
    raw = loss_oracle.loss_function(fmask, lab_t, id_t, hm_t, dec_t)
    alphas = torch.tensor([0.3, 0.05, 0.1, 0.02, 0.2, 0.08, 0.1, 0.1, 0.05])
    keys = ["keymap_loss", "size_loss", "textline_loss", "separator_loss", "id_loss", "code1_loss", "code2_loss", "code4_loss", "code8_loss"]
    (sum(a * raw[k] for a, k in zip(alphas, keys)) * 0.5).backward()
    sel = torch.nonzero(fmask).flatten().to(torch.int32)
    lib = L.load()
    PAD = 1104
    ar = Arena()
    o_maps = ar.put(torch.from_numpy(hm).permute(0, 2, 3, 1).contiguous())
    o_lab, o_id, o_sel = ar.put(lab_t), ar.put(torch.from_numpy(idmap).to(torch.int32)), ar.put(sel)
    o_d = [ar.put(torch.from_numpy(d)) for d in dec]
    o_lossv, o_scr = ar.reserve(64), ar.reserve(int(lib.ftc_losses_scratch_bytes()))
    o_al, o_gm, o_gd = ar.put(alphas), ar.reserve(B * h * w * 9 * 4), ar.reserve(3 * n * PAD * 4)
    ar.materialize()
    common = dict(B=B, H=h, W=w, aux0=n, in_=o_maps, in2=o_lab, w=o_id, w2=o_d[0], bias=o_d[1], bias2=o_d[2], scale=o_sel)
    run_op(dict(kind=L.OP_LOSSES, out=o_lossv, aux=o_scr, **common), ar)
    lossv = ar.read(o_lossv, (16,), torch.float32)
    for i, k in enumerate(keys):
        assert abs(float(lossv[1 + i]) - float(raw[k])) < 2e-5 * max(1.0, abs(float(raw[k]))), k
    run_op(dict(kind=L.OP_LOSS_BWD, aux1=PAD, Cout=_fbits(0.5), shift=o_al, aux=o_lossv, out=o_gm, out2=o_gd, **common), ar)
    gm = ar.read(o_gm, (B, h, w, 9), torch.float32).permute(0, 3, 1, 2)
    assert _rel(gm, hm_t.grad) < 2e-5
    gd = ar.read(o_gd, (3, n, PAD), torch.float32)
    for j, m in enumerate((1091, 1093, 1097)):
        assert _rel(gd[j, :, :m], dec_t[j].grad) < 2e-5
        assert float(gd[j, :, m:].abs().max()) == 0.0
    # gather / scatter of the selected rows
    feat = torch.randn(B * h * w, 100)
    ar2 = Arena()
    o_f, o_s2 = ar2.put(feat), ar2.put(sel)
    o_rows, o_back = ar2.reserve(n * 128 * 4), ar2.reserve(B * h * w * 128 * 4)
    ar2.materialize()
    run_op(dict(kind=L.OP_GATHER_ROWS, B=B, H=h, W=w, Cin=100, Cout_total=128, aux0=n, in_=o_f, in2=o_s2, out=o_rows), ar2)
    rows = ar2.read(o_rows, (n, 128), torch.float32)
    assert torch.equal(rows[:, :100], feat[fmask]) and float(rows[:, 100:].abs().max()) == 0.0
    run_op(dict(kind=L.OP_SCATTER_ROWS, B=B, H=h, W=w, Cout_total=128, aux0=n, in_=o_rows, in2=o_s2, out=o_back), ar2)
    back = ar2.read(o_back, (B * h * w, 128), torch.float32)
    assert torch.equal(back[fmask][:, :100], feat[fmask]) and float(back[~fmask].abs().max()) == 0.0
