"""-m gpu: seeded fuzz of the convolution dispatcher -- random shapes (ragged H/W, channel counts off the tile sizes, slices of
wider buffers, stride 2, residual, SE scale) x EVERY kernel variant the tuner may pick for the shape (tile, staging, K step,
split-K, halo) -- against PyTorch fp32 on the CPU.  Guards the legality checks: a variant ftc_plan_create accepts must compute
the right answer."""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from findtextcenternet_amd import _lib as L
from findtextcenternet_amd import tuning as T
from gpu_harness import Arena, bf16_round, presplit_f16x3, run_op, tdtype, to_dev_bytes

pytestmark = pytest.mark.gpu
ACT = {L.ACT_NONE: lambda v: v, L.ACT_SILU: F.silu, L.ACT_GELU: F.gelu}


def _case(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    bf16 = bool(rng.integers(0, 4))                      # 3 of 4 cases in the bf16 mode (more variants there)
    k = int(rng.choice([1, 3]))
    stride = int(rng.choice([1, 1, 2]))
    unit = 8 if bf16 else 4
    Cin = int(rng.choice([unit * m for m in (1, 2, 3, 4, 8, 12, 16, 24, 36, 48, 64, 96)]))
    Cout = int(rng.choice([1, 2, 8, 24, 32, 64, 96, 100, 128, 192, 200, 256, 384]))
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(5, 41)), int(rng.integers(5, 41))
    slice_in = bool(rng.integers(0, 3) == 0)
    CinT = Cin + (unit * int(rng.integers(1, 4)) if slice_in else 0)
    cin_off = unit * int(rng.integers(0, (CinT - Cin) // unit + 1)) if slice_in else 0
    slice_out = bool(rng.integers(0, 4) == 0)
    CoutT = Cout + (4 * int(rng.integers(1, 3)) if slice_out else 0)
    cout_off = 4 * int(rng.integers(0, (CoutT - Cout) // 4 + 1)) if slice_out else 0
    act = int(rng.choice([L.ACT_NONE, L.ACT_SILU, L.ACT_GELU]))
    residual = bool(rng.integers(0, 3) == 0) and not slice_out
    se = k == 1 and bool(rng.integers(0, 4) == 0)
    if bf16:
        idt = L.BF16 if not se or rng.integers(0, 2) else L.BF16
        odt = int(rng.choice([L.BF16, L.F32]))
        wdt = L.BF16
        if rng.integers(0, 5) == 0:
            idt = L.F32                                   # fp32 trunk read by a bf16 GEMM
    else:
        idt = odt = wdt = L.F32
    return dict(B=B, H=H, W=W, Cin=Cin, CinT=CinT, cin_off=cin_off, Cout=Cout, CoutT=CoutT, cout_off=cout_off, k=k, stride=stride,
                act=act, residual=residual, se=se, idt=idt, odt=odt, wdt=wdt)


def _case_f32(seed):
    for i in range(64):
        c = _case(seed + 1000 * i)
        if c["wdt"] == L.F32:
            return c
    raise AssertionError("no fp32 case found")


@pytest.mark.parametrize("seed", range(10))
def test_every_legal_kernel_variant_of_a_random_conv_fp16x3(seed):
    """The same fuzz for the fp16x3 arithmetic (FTC_FLAG_SPLIT16: fp32 tensors, pre-split weights, three fp16 MFMAs per product) on the
    fp32 cases, held to 2e-5 (the exact-fp32 kernels: 2e-4 budget, 1e-6 measured)."""
    _run_fuzz_case(_case_f32(7000 + seed), x3=True)


@pytest.mark.parametrize("seed", range(24))
def test_every_legal_kernel_variant_of_a_random_conv(seed):
    _run_fuzz_case(_case(9000 + seed), x3=False)


def _run_fuzz_case(c, x3):
    B, H, W, Cin, k, stride = c["B"], c["H"], c["W"], c["Cin"], c["k"], c["stride"]
    g = torch.Generator().manual_seed(zlib.crc32(str(c).encode()) % 100000)
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x_full = torch.randn(B, H, W, c["CinT"], generator=g)
    w = torch.randn(c["Cout"], Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(c["Cout"], generator=g) * 0.3
    res = torch.randn(B, Ho, Wo, c["Cout"], generator=g) if c["residual"] else None
    sc = torch.rand(B, Cin, generator=g) + 0.25 if c["se"] else None
    if c["idt"] == L.BF16:
        x_full = bf16_round(x_full)
    wq = bf16_round(w) if c["wdt"] == L.BF16 else w
    xin = x_full[..., c["cin_off"]:c["cin_off"] + Cin]
    if c["se"]:
        xin = xin * sc[:, None, None, :]
    if c["wdt"] == L.BF16 and (c["se"] or c["idt"] == L.F32):
        xin = bf16_round(xin)
    ref = ACT[c["act"]](F.conv2d(xin.permute(0, 3, 1, 2), wq, None, stride, pad).permute(0, 2, 3, 1) + bias)
    if c["residual"]:
        ref = ref + res
    ar = Arena()
    o_in = ar.put(to_dev_bytes(x_full, c["idt"]))
    wk = w.permute(0, 2, 3, 1).reshape(c["Cout"], k * k, Cin)
    o_w = ar.put(presplit_f16x3(wk) if x3 else to_dev_bytes(wk, c["wdt"]))
    o_b = ar.put(bias)
    o_res = ar.put(res) if c["residual"] else None
    o_sc = ar.put(sc) if c["se"] else None
    esz = 4 if c["odt"] == L.F32 else 2
    o_out = ar.reserve(B * Ho * Wo * c["CoutT"] * esz)
    ar.materialize()
    fields = dict(kind=L.OP_CONV, flags=(L.FLAG_RESIDUAL if c["residual"] else 0) | (L.FLAG_SE_SCALE if c["se"] else 0) | (L.FLAG_SPLIT16 if x3 else 0), act=c["act"],
                  in_dtype=c["idt"], out_dtype=c["odt"], w_dtype=c["wdt"], B=B, H=H, W=W, Ho=Ho, Wo=Wo, Cin=Cin, Cin_total=c["CinT"],
                  cin_off=c["cin_off"], Cout=c["Cout"], Cout_total=c["CoutT"], cout_off=c["cout_off"], ksize=k, stride=stride,
                  res_dtype=L.F32, in_=o_in, in2=o_res, out=o_out, w=o_w, bias=o_b, scale=o_sc)
    probe = L.Op()
    for name in ("w_dtype", "in_dtype", "out_dtype", "Cin", "Cout", "ksize", "stride", "groups"):
        setattr(probe, name, fields.get(name, 0))
    tol = 2e-5 if x3 else 2e-4 if c["wdt"] == L.F32 else 1.5e-2
    ran = 0
    for aux0 in [0] + T.candidates(probe):
        try:
            run_op(dict(fields, aux0=aux0), ar)
        except L.FtcError:
            continue                                       # variant not legal for this op: refused at plan creation
        out = ar.read(o_out, (B, Ho, Wo, c["CoutT"]), tdtype(c["odt"]))[..., c["cout_off"]:c["cout_off"] + c["Cout"]].float()
        err = float((out - ref).abs().max() / (ref.abs().max() + 1e-12))
        assert err < tol, (c, T.describe(aux0), err)
        ran += 1
    assert ran >= 3, (c, ran)
