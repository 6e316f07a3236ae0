"""Host-side construction of the HIP execution plan for the detector forward pass.

Mirrors, as data, what the reference expresses as nn.Module composition:
``CenterNetDetection.forward`` (``/root/reference/models/detector.py:217-230``) = stem + 100
Fused-MBConv/MBConv blocks with taps (``BackboneModel.forward`` ``:139-146``) + nine ``Leafmap`` heads
(``:192-201``), followed by the NMS of ``CenterNetDetector.forward`` (``:289-296``).

Two steps:

* ``pack_weights``  -- once per checkpoint: folds every eval-mode BatchNorm that directly follows a
  convolution into that convolution (scale into the weights, shift into a bias; fp64 arithmetic),
  re-lays weights out K-major ``[Cout][kh*kw][Cin]`` for the NHWC implicit GEMM, converts to the
  MFMA compute type, and concatenates everything into one blob that is uploaded to HBM once.
  The per-head input BatchNorm (``Leafmap.in_bn``) is NOT folded into the following zero-padded
  3x3 conv (its shift would leak into the padding ring); it is applied by the UPCAT kernel.
* ``build_plan``    -- once per input shape: the op list (``ftc_op`` records) with activation
  buffers placed in one workspace arena by a liveness-based first-fit allocator.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from .schema import (BACKBONE_BN_EPS, FPN_DIM, HEAD_BN_EPS, HEADS, LAST_CHANNEL, STAGES, TAP_DIMS, backbone_blocks,
                     feature_dim)

ALIGN = 256


def _align(n: int, a: int = ALIGN) -> int:
    return (n + a - 1) // a * a


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
@dataclass
class PackedWeights:
    blob: np.ndarray                      # uint8
    table: Dict[str, int]                 # name -> byte offset
    mode: str                             # "fp32" | "bf16"
    model_size: str

    @property
    def nbytes(self) -> int:
        return int(self.blob.nbytes)


class _Blob:
    def __init__(self):
        self.parts: List[Tuple[int, np.ndarray]] = []
        self.table: Dict[str, int] = {}
        self.size = 0

    def add(self, name: str, arr: np.ndarray) -> None:
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        self.table[name] = self.size
        self.parts.append((self.size, raw))
        self.size = _align(self.size + raw.nbytes)

    def finish(self) -> np.ndarray:
        out = np.zeros(self.size, np.uint8)
        for off, raw in self.parts:
            out[off:off + raw.nbytes] = raw
        return out


def _to_compute(a: np.ndarray, mode: str) -> np.ndarray:
    if mode == "fp32":
        return a.astype(np.float32)
    t = torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16)      # round-to-nearest-even
    return t.view(torch.int16).numpy()


def _fold(sd, conv_key: str, bn_prefix: str, eps: float):
    """conv weight [O,I,kh,kw] and eval BN -> (W*s [O,I,kh,kw] f64, b f64)."""
    w = sd[conv_key].detach().cpu().double().numpy()
    g = sd[bn_prefix + ".weight"].detach().cpu().double().numpy()
    b = sd[bn_prefix + ".bias"].detach().cpu().double().numpy()
    m = sd[bn_prefix + ".running_mean"].detach().cpu().double().numpy()
    v = sd[bn_prefix + ".running_var"].detach().cpu().double().numpy()
    s = g / np.sqrt(v + eps)
    return w * s[:, None, None, None], b - m * s


def _kmajor(w: np.ndarray) -> np.ndarray:
    """[O,I,kh,kw] -> [O, kh*kw, I]."""
    o, i, kh, kw = w.shape
    return np.ascontiguousarray(w.transpose(0, 2, 3, 1)).reshape(o, kh * kw, i)


TOP6 = ("textline", "sepatator", "code1", "code2", "code4", "code8")      # heads[2..7]: one output channel each, channels 4..9


def pack_weights(sd: Dict[str, torch.Tensor], mode: str = "fp32", model_size: str = "xl") -> PackedWeights:
    """sd: ``CenterNetDetection`` state_dict (keys without the ``detector.`` prefix)."""
    assert mode in ("fp32", "bf16")
    bmode = os.environ.get("FTC_EXP_BACKBONE", mode)      # experiment switches: numeric mode of the backbone / of the heads
    hmode = os.environ.get("FTC_EXP_HEADS", mode)
    mode = bmode
    if any(k.startswith("detector.") for k in sd):
        sd = {k[len("detector."):]: v for k, v in sd.items() if k.startswith("detector.")}
    bl = _Blob()

    def conv_bn(name: str, conv_key: str, bn_prefix: str, eps: float):
        w, b = _fold(sd, conv_key, bn_prefix, eps)
        bl.add(name + ".w", _to_compute(_kmajor(w), mode))
        bl.add(name + ".b", b.astype(np.float32))

    # stem: [C0,3,3,3] -> [(r*3+s)*3+c][C0] fp32 (VALU kernel, always fp32)
    w, b = _fold(sd, "backbone.features.0.0.weight", "backbone.features.0.1", BACKBONE_BN_EPS)
    bl.add("stem.w", np.ascontiguousarray(w.transpose(2, 3, 1, 0)).reshape(27, -1).astype(np.float32))
    bl.add("stem.b", b.astype(np.float32))
    for stage in backbone_blocks(model_size):
        for blk in stage:
            p = blk.prefix + ".block"
            if blk.kind == "fused":
                conv_bn(p + ".0", p + ".0.0.weight", p + ".0.1", BACKBONE_BN_EPS)
                if blk.exp != blk.cin:
                    conv_bn(p + ".1", p + ".1.0.weight", p + ".1.1", BACKBONE_BN_EPS)
            else:
                conv_bn(p + ".0", p + ".0.0.weight", p + ".0.1", BACKBONE_BN_EPS)
                w, b = _fold(sd, p + ".1.0.weight", p + ".1.1", BACKBONE_BN_EPS)          # [C,1,3,3]
                bl.add(p + ".1.w", np.ascontiguousarray(w.reshape(w.shape[0], 9).T).astype(np.float32))   # [9][C]
                bl.add(p + ".1.b", b.astype(np.float32))
                w1 = sd[p + ".2.fc1.weight"].detach().cpu().float().numpy()
                w2 = sd[p + ".2.fc2.weight"].detach().cpu().float().numpy()
                bl.add(p + ".2.w1", w1.reshape(w1.shape[0], w1.shape[1]))                 # [S][C]
                bl.add(p + ".2.b1", sd[p + ".2.fc1.bias"].detach().cpu().float().numpy())
                bl.add(p + ".2.w2t", np.ascontiguousarray(w2.reshape(w2.shape[0], w2.shape[1]).T))   # [S][C]
                bl.add(p + ".2.b2", sd[p + ".2.fc2.bias"].detach().cpu().float().numpy())
                conv_bn(p + ".3", p + ".3.0.weight", p + ".3.1", BACKBONE_BN_EPS)
    nfeat = len(STAGES[model_size]) + 1
    hp = f"backbone.features.{nfeat}"
    conv_bn(hp, hp + ".0.weight", hp + ".1", BACKBONE_BN_EPS)
    mode = hmode
    ntap = len(TAP_DIMS[model_size])
    # FPN level 0 of all nine heads as ONE convolution over the shared 1/32 tap: each head's input
    # BatchNorm is folded in exactly -- scale into the weights, shift into a 16-entry border-case bias
    # table (the shift does not see the zero padding ring, so its contribution depends on which taps of
    # the 3x3 window fall inside the image).  Leafmap.forward i=0, models/detector.py:194-197.
    wm_all, bias16_all = [], []
    for name, out_dim, _ in HEADS:
        q = f"{name}.in_bn.{ntap - 1}"
        g = sd[q + ".weight"].detach().cpu().double().numpy()
        si = g / np.sqrt(sd[q + ".running_var"].detach().cpu().double().numpy() + HEAD_BN_EPS)
        ti = sd[q + ".bias"].detach().cpu().double().numpy() - sd[q + ".running_mean"].detach().cpu().double().numpy() * si
        wf, bo = _fold(sd, f"{name}.upsamplers.0.0.weight", f"{name}.upsamplers.0.1", HEAD_BN_EPS)      # [192,C,3,3]
        wm_all.append(_kmajor(wf * si[None, :, None, None]))
        tmap = np.einsum("ncrs,c->nrs", wf, ti)                                                      # [192,3,3]
        b16 = np.zeros((16, wf.shape[0]))
        for idx in range(16):
            rows = [r for r in range(3) if not (r == 0 and idx & 1) and not (r == 2 and idx & 2)]
            cols = [c for c in range(3) if not (c == 0 and idx & 4) and not (c == 2 and idx & 8)]
            b16[idx] = bo + tmap[np.ix_(range(wf.shape[0]), rows, cols)].sum(axis=(1, 2))
        bias16_all.append(b16)
    bl.add("heads.L0.w", _to_compute(np.concatenate(wm_all, axis=0), mode))
    bl.add("heads.L0.b", np.concatenate(bias16_all, axis=1).astype(np.float32))                      # [16][9*192]
    # FPN levels 1.. and the input BatchNorms of the nine heads are stored head-major ([9][...]) so that one grouped
    # launch (ftc_op.groups = 9) covers all heads of a level; so are the six one-channel top convolutions whose heat-map
    # channels are consecutive (textline, separator, code1/2/4/8 -> channels 4..9).
    def bn_affine(q):
        g = sd[q + ".weight"].detach().cpu().double().numpy()
        s_ = g / np.sqrt(sd[q + ".running_var"].detach().cpu().double().numpy() + HEAD_BN_EPS)
        t_ = sd[q + ".bias"].detach().cpu().double().numpy() - sd[q + ".running_mean"].detach().cpu().double().numpy() * s_
        return s_.astype(np.float32), t_.astype(np.float32)

    for i in range(ntap - 1):
        st = [bn_affine(f"{name}.in_bn.{i}") for name, _, _ in HEADS]
        bl.add(f"heads.in_bn.{i}.scale", np.stack([a for a, _ in st]))
        bl.add(f"heads.in_bn.{i}.shift", np.stack([b for _, b in st]))
    for i in range(1, ntap):
        wb = [_fold(sd, f"{name}.upsamplers.{i}.0.weight", f"{name}.upsamplers.{i}.1", HEAD_BN_EPS) for name, _, _ in HEADS]
        bl.add(f"heads.L{i}.w", _to_compute(np.stack([_kmajor(w) for w, _ in wb]), mode))
        bl.add(f"heads.L{i}.b", np.stack([b for _, b in wb]).astype(np.float32))

    if mode == "bf16" and ntap >= 2:
        # Last level with the input BatchNorm of the backbone tap folded in exactly (as level 0 above): scale into the tap
        # columns of the weights, shift into a 16-case border bias table -- so the convolution reads the shared bf16 trunk
        # copy of the tap instead of nine batch-normed copies (FTC_FLAG_GROUP_IN2_SHARED + FTC_FLAG_BORDER_BIAS).
        i = ntap - 1
        ws, bs = [], []
        for name, _, _ in HEADS:
            q = f"{name}.in_bn.0"
            g = sd[q + ".weight"].detach().cpu().double().numpy()
            si = g / np.sqrt(sd[q + ".running_var"].detach().cpu().double().numpy() + HEAD_BN_EPS)
            ti = sd[q + ".bias"].detach().cpu().double().numpy() - sd[q + ".running_mean"].detach().cpu().double().numpy() * si
            wf, bo = _fold(sd, f"{name}.upsamplers.{i}.0.weight", f"{name}.upsamplers.{i}.1", HEAD_BN_EPS)      # [192, 192+tc, 3, 3]
            wm = wf.copy()
            wm[:, FPN_DIM:] *= si[None, :, None, None]
            tmap = np.einsum("ncrs,c->nrs", wf[:, FPN_DIM:], ti)
            b16 = np.zeros((16, wf.shape[0]))
            for idx in range(16):
                rows = [r for r in range(3) if not (r == 0 and idx & 1) and not (r == 2 and idx & 2)]
                cols = [c for c in range(3) if not (c == 0 and idx & 4) and not (c == 2 and idx & 8)]
                b16[idx] = bo + tmap[np.ix_(range(wf.shape[0]), rows, cols)].sum(axis=(1, 2))
            ws.append(_kmajor(wm))
            bs.append(b16)
        bl.add(f"heads.L{i}f.w", _to_compute(np.stack(ws), mode))
        bl.add(f"heads.L{i}f.b", np.stack(bs).astype(np.float32))                               # [9][16][192]

    def top(name):
        w = sd[f"{name}.top_conv.0.weight"].detach().cpu().double().numpy()
        return _kmajor(w), sd[f"{name}.top_conv.0.bias"].detach().cpu().float().numpy()

    for name in ("keyheatmap", "sizes", "feature"):
        w, b = top(name)
        bl.add(f"{name}.top_conv.w", _to_compute(w, mode))
        bl.add(f"{name}.top_conv.b", b)
    tops = [top(name) for name in TOP6]
    bl.add("heads.top6.w", _to_compute(np.stack([w for w, _ in tops]), mode))
    bl.add("heads.top6.b", np.stack([b for _, b in tops]).astype(np.float32))
    if mode == "bf16":
        # The eight map heads' top convolutions as per-pixel tap matrices for the fused last-level epilogue
        # (FTC_FLAG_TOP_FUSE + FTC_OP_TAPSUM): row tap*Co + o of head g = top_conv weight [o, :, r, s], 32 rows zero padded.
        wt = np.zeros((len(HEADS) - 1, 32, FPN_DIM))
        bias, omap = [], []
        for g, (name, co, ch0) in enumerate(HEADS[:-1]):
            w, b = top(name)                                    # [co][9][192]
            for o in range(co):
                wt[g, np.arange(9) * co + o, :] = w[o]
                bias.append(b[o])
                omap.append((g, o, co, (0 if ch0 == 0 else ch0 + 1) + o))
        bl.add("heads.top8.wt", _to_compute(wt, mode))
        bl.add("heads.top8.b", np.asarray(bias, np.float32))
        bl.add("heads.top8.map", np.asarray(omap, np.int32))
    pw = PackedWeights(bl.finish(), bl.table, bmode, model_size)
    pw.hmode = hmode
    return pw


# ------------------------------------------------------------------------------------------------
# plan
# ------------------------------------------------------------------------------------------------
@dataclass
class OpMeta:
    name: str
    kind: str
    flops: float = 0.0          # 2*MACs of the convolution (bias/activation excluded)
    bytes: float = 0.0          # algorithmic bytes: inputs + outputs + weights, each once


@dataclass
class Plan:
    ops: "C.Array"
    meta: List[OpMeta]
    workspace_bytes: int
    B: int
    H: int
    W: int
    h: int
    w: int
    mode: str
    handle: Optional[int] = None
    peak_live_bytes: int = 0
    total_buffer_bytes: int = 0


@dataclass
class _Buf:
    nbytes: int
    first: int = 10 ** 9
    last: int = -1
    offset: int = -1


class _Builder:
    def __init__(self, pw: PackedWeights, B: int, H: int, W: int, nchw: bool):
        self.pw, self.B, self.H, self.W, self.nchw = pw, B, H, W, nchw
        self.mode = pw.mode
        self.act = L.F32 if self.mode == "fp32" else L.BF16       # expanded / FPN activations
        self.trunk = L.F32                                         # residual trunk + taps stay fp32
        self.cdt = L.F32 if self.mode == "fp32" else L.BF16       # MFMA compute type
        self.ops: List[dict] = []
        self.meta: List[OpMeta] = []
        self.bufs: List[_Buf] = []

    @staticmethod
    def esize(dt: int) -> int:
        return 4 if dt == L.F32 else 2

    def buf(self, nelem: int, dt: int) -> int:
        self.bufs.append(_Buf(_align(nelem * self.esize(dt))))
        return len(self.bufs) - 1

    def wref(self, name: str):
        return ("w", self.pw.table[name])

    def emit(self, meta: OpMeta, **f) -> None:
        idx = len(self.ops)
        for k in ("in_", "in2", "out", "aux", "scale", "out2", "w", "w2"):
            r = f.get(k)
            if isinstance(r, tuple) and r[0] == "buf":
                b = self.bufs[r[1]]
                b.first, b.last = min(b.first, idx), max(b.last, idx)
        self.ops.append(f)
        self.meta.append(meta)

    # --- op helpers ---------------------------------------------------------------------------
    def conv(self, name, x, xdt, H, W, cin, cin_total, cin_off, wname, cout, k, stride, act, out, odt, cout_total=None,
             cout_off=0, residual=None, res_dt=0, se=None, out2=None, extra_flags=0, wsets=None, groups=1, w_off=0, b_off=0):
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        flags = (L.FLAG_RESIDUAL if residual is not None else 0) | (L.FLAG_SE_SCALE if se is not None else 0) | extra_flags
        macs = groups * self.B * Ho * Wo * cout * cin * k * k
        byt = groups * (self.B * H * W * cin * self.esize(xdt) + self.B * Ho * Wo * cout * self.esize(odt)
                        + cout * cin * k * k * self.esize(self.cdt))
        if residual is not None:
            byt += self.B * Ho * Wo * cout * self.esize(res_dt)
        if out2 is not None:
            byt += self.B * Ho * Wo * cout * 2
        if wsets is not None:                     # one weight set per image (SE excitation folded in by the SE op)
            flags |= L.FLAG_W_PER_IMAGE
            byt += (self.B - 1) * cout * cin * k * k * self.esize(self.cdt)
        self.emit(OpMeta(name, f"conv{k}x{k}", 2.0 * macs, byt), kind=L.OP_CONV, flags=flags, act=act, in_dtype=xdt,
                  out_dtype=odt, w_dtype=self.cdt, B=self.B, H=H, W=W, Ho=Ho, Wo=Wo, Cin=cin, Cin_total=cin_total,
                  cin_off=cin_off, Cout=cout, Cout_total=cout_total or cout, cout_off=cout_off, ksize=k, stride=stride,
                  res_dtype=res_dt, in_=x, in2=residual, out=out,
                  w=wsets if wsets is not None else ("w", self.pw.table[wname + ".w"] + w_off),
                  bias=("w", self.pw.table[wname + ".b"] + b_off), scale=se, out2=out2, groups=groups if groups > 1 else 0)
        return Ho, Wo

    def build(self) -> Plan:
        B, H, W = self.B, self.H, self.W
        ms = self.pw.model_size
        stages = backbone_blocks(ms)
        c0 = STAGES[ms][0][4]
        T, A = self.trunk, self.act
        # In bf16 mode every trunk tensor (fp32, feeds the residual adds and the FPN taps) is written
        # together with a bf16 copy by the producing epilogue; the next GEMM reads the copy.
        dual = self.mode == "bf16"
        G = A if dual else T                      # dtype the GEMMs read the trunk in

        def trunk(nelem):
            return ("buf", self.buf(nelem, T)), (("buf", self.buf(nelem, L.BF16)) if dual else None)

        # stem
        h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        x, xb = trunk(B * h * w * c0)
        self.emit(OpMeta("backbone.features.0", "stem", 2.0 * B * h * w * c0 * 27,
                         B * H * W * 3 * 4 + B * h * w * c0 * (self.esize(T) + (2 if dual else 0))),
                  kind=L.OP_STEM, flags=L.FLAG_IN_NCHW if self.nchw else 0, act=L.ACT_SILU, in_dtype=L.F32, out_dtype=T,
                  B=B, H=H, W=W, Ho=h, Wo=w, Cin=3, Cout=c0, ksize=3, stride=2, in_=("input", 0), out=x, out2=xb,
                  w=self.wref("stem.w"), bias=self.wref("stem.b"))
        taps = []
        tap_copies = []                           # bf16 trunk copies of the backbone taps (bf16 mode), same order
        for si, stage in enumerate(stages):
            for blk in stage:
                p = blk.prefix + ".block"
                ho, wo = (h - 1) // blk.stride + 1, (w - 1) // blk.stride + 1
                res = x if blk.residual else None
                gin = xb if dual else x           # GEMM-side view of the block input
                y, yb = trunk(B * ho * wo * blk.cout)
                if blk.kind == "fused" and blk.exp == blk.cin:
                    self.conv(p + ".0", gin, G, h, w, blk.cin, blk.cin, 0, p + ".0", blk.cout, 3, blk.stride, L.ACT_SILU, y, T,
                              residual=res, res_dt=T, out2=yb)
                elif blk.kind == "fused":
                    e = ("buf", self.buf(B * ho * wo * blk.exp, A))
                    self.conv(p + ".0", gin, G, h, w, blk.cin, blk.cin, 0, p + ".0", blk.exp, 3, blk.stride, L.ACT_SILU, e, A)
                    self.conv(p + ".1", e, A, ho, wo, blk.exp, blk.exp, 0, p + ".1", blk.cout, 1, 1, L.ACT_NONE, y, T,
                              residual=res, res_dt=T, out2=yb)
                else:
                    e = ("buf", self.buf(B * h * w * blk.exp, A))
                    self.conv(p + ".0", gin, G, h, w, blk.cin, blk.cin, 0, p + ".0", blk.exp, 1, 1, L.ACT_SILU, e, A)
                    th = 8 if blk.stride == 1 else 4
                    P = ((ho + th - 1) // th) * ((wo + 7) // 8)
                    d = ("buf", self.buf(B * ho * wo * blk.exp, A))
                    part = ("buf", self.buf(B * P * blk.exp, L.F32))
                    self.emit(OpMeta(p + ".1", "dwconv3x3", 2.0 * B * ho * wo * blk.exp * 9,
                                     B * (h * w + ho * wo) * blk.exp * self.esize(A) + blk.exp * 40),
                              kind=L.OP_DWCONV, act=L.ACT_SILU, in_dtype=A, out_dtype=A, B=B, H=h, W=w, Ho=ho, Wo=wo,
                              Cin=blk.exp, Cout=blk.exp, ksize=3, stride=blk.stride, aux0=P, in_=e, out=d,
                              w=self.wref(p + ".1.w"), bias=self.wref(p + ".1.b"), aux=part)
                    sc = ("buf", self.buf(B * blk.exp, L.F32))
                    hid = ("buf", self.buf(B * blk.squeeze, L.F32))
                    # bf16 mode: the SE op also writes the project weights scaled per image, so that the project
                    # convolution streams both operands by DMA instead of rescaling activations while staging them.
                    # Needs a 64-pixel tile that divides the image (true for every 768x768 stage).
                    fold = dual and (ho * wo) % 64 == 0 and blk.exp % 8 == 0
                    wb = ("buf", self.buf(B * blk.cout * blk.exp, L.BF16)) if fold else None
                    se_bytes = 8.0 * blk.exp * blk.squeeze + B * P * blk.exp * 4 + ((B + 1) * blk.cout * blk.exp * 2 if fold else 0)
                    self.emit(OpMeta(p + ".2", "se", 4.0 * B * blk.exp * blk.squeeze, se_bytes),
                              kind=L.OP_SE, flags=L.FLAG_SE_FOLD if fold else 0, w_dtype=L.BF16 if fold else 0, B=B, H=ho, W=wo,
                              Cin=blk.exp, Cout=blk.exp, Cout_total=blk.cout if fold else 0, aux0=blk.squeeze, aux1=P, aux=part,
                              out=sc, in2=hid, w=self.wref(p + ".2.w1"), w2=self.wref(p + ".2.w2t"), bias=self.wref(p + ".2.b1"),
                              bias2=self.wref(p + ".2.b2"), in_=self.wref(p + ".3.w") if fold else None, out2=wb)
                    self.conv(p + ".3", d, A, ho, wo, blk.exp, blk.exp, 0, p + ".3", blk.cout, 1, 1, L.ACT_NONE, y, T,
                              residual=res, res_dt=T, se=None if fold else sc, out2=yb, wsets=wb)
                x, xb, h, w = y, yb, ho, wo
            if (si + 1) in (2, 3, 5):
                taps.append((x, stage[-1].cout, h, w, T))
                tap_copies.append(xb)
        nfeat = len(stages) + 1
        hp = f"backbone.features.{nfeat}"
        hmode = getattr(self.pw, "hmode", self.mode)
        if hmode != self.mode:                    # experiment: heads in the other numeric mode
            x4dt = L.F32
        else:
            x4dt = A
        x4 = ("buf", self.buf(B * h * w * LAST_CHANNEL, x4dt))
        self.conv(hp, xb if dual else x, G, h, w, stages[-1][-1].cout, stages[-1][-1].cout, 0, hp, LAST_CHANNEL, 1, 1, L.ACT_SILU, x4, x4dt)
        taps.append((x4, LAST_CHANNEL, h, w, x4dt))
        if hmode != self.mode:
            self.mode = hmode
            A = self.act = L.F32 if hmode == "fp32" else L.BF16
            self.cdt = A
            dual = False
            tap_copies = []
        mh, mw = taps[0][2], taps[0][3]
        # heads.  Level 0 of all nine heads is one convolution (see pack_weights); levels 1.. per head.
        ntap = len(taps)
        nh = len(HEADS)
        t4, c4, h4, w4, dt4 = taps[ntap - 1]
        y0 = ("buf", self.buf(B * h4 * w4 * nh * FPN_DIM, A))
        self.conv("heads.upsamplers.0", t4, dt4, h4, w4, c4, c4, 0, "heads.L0", nh * FPN_DIM, 3, 1, L.ACT_GELU, y0, A,
                  extra_flags=L.FLAG_BORDER_BIAS)
        # Levels 1.. : the nine heads have identical shapes, so each level is ONE grouped launch (upsample+concat, then the
        # 3x3 convolution) over head-major stacked tensors [9][B,h,w,C] -- 2592 instead of 288 workgroups for the 96x96
        # level, no 1.1-round tails on 256 CUs.
        y, yh, yw = y0, h4, w4
        nmap = nh - 1                                          # the map heads (all but `feature`)
        fuse_top = dual and taps[0][1] + FPN_DIM == 256 and not os.environ.get("FTC_NO_TOPFUSE")
        TW = 20                                                # floats per pixel of the tap tensor T (9 * 2 outputs, padded)
        fuse_up = dual and not os.environ.get("FTC_NO_UPFUSE")
        for i in range(1, ntap):
            tbuf, tc, th_, tw_, tdt = taps[ntap - 1 - i]
            cy = FPN_DIM
            cin = cy + tc
            M = B * th_ * tw_
            last = i == ntap - 1
            bn_s, bn_t = self.wref(f"heads.in_bn.{ntap - 1 - i}.scale"), self.wref(f"heads.in_bn.{ntap - 1 - i}.shift")
            wsz = FPN_DIM * cin * 9 * self.esize(self.cdt)
            # bf16 mode, levels whose upsampled source is a stacked tensor (2..): the concatenated input is never
            # materialised -- the convolution upsamples while it stages its halo (FTC_FLAG_UPCAT_IN) and reads the
            # batch-normed backbone tap (one small grouped elementwise launch) as its second channel source.
            # (measured: with 32-channel K blocks the per-block upsampling work outweighs the saved pass -- level 2, Cin 288,
            # keeps the two-kernel form: 1032 us fused vs 643 + 194 us)
            up_in = fuse_up and i >= 2 and th_ == 2 * yh and tw_ == 2 * yw and cy % 64 == 0 and tc % 64 == 0
            # ... and on the last level the tap's BatchNorm is folded into the weights + a border bias table, so that all heads read
            # the ONE bf16 trunk copy of the tap (no batch-normed copies at all)
            tap_copy = tap_copies[ntap - 1 - i] if ntap - 1 - i < len(tap_copies) else None
            bn_fold = up_in and last and tap_copy is not None and f"heads.L{i}f.w" in self.pw.table and not os.environ.get("FTC_NO_BNFOLD")
            if bn_fold:
                tapbn = tap_copy
                src_bytes = B * yh * yw * cy * 2 + M * tc * 2 // nh
            elif up_in:
                tapbn = ("buf", self.buf(nh * M * tc, A))
                self.emit(OpMeta(f"heads.tapbn{i}", "upcat", 0.0, nh * M * tc * self.esize(A) + M * tc * self.esize(tdt)),
                          kind=L.OP_UPCAT, in_dtype=A, out_dtype=A, res_dtype=tdt, B=B, H=th_, W=tw_, Ho=th_, Wo=tw_, Cin=tc, Cout=tc,
                          aux0=0, aux1=tc, groups=nh, in2=tbuf, out=tapbn, scale=bn_s, shift=bn_t)
                src_bytes = B * yh * yw * cy * 2 + M * tc * 2
            else:
                cat = ("buf", self.buf(nh * M * cin, A))
                self.emit(OpMeta(f"heads.cat{i}", "upcat", 0.0,
                                 nh * (M * (cin * self.esize(A) + tc * self.esize(tdt)) + B * yh * yw * cy * self.esize(A))),
                          kind=L.OP_UPCAT, flags=L.FLAG_GROUP_IN_SLICE if i == 1 else 0, in_dtype=A, out_dtype=A, res_dtype=tdt, B=B,
                          H=yh, W=yw, Ho=th_, Wo=tw_, Cin=cin, Cin_total=nh * FPN_DIM if i == 1 else FPN_DIM, cin_off=0, Cout=cin,
                          aux0=cy, aux1=tc, groups=nh, in_=y, in2=tbuf, out=cat, scale=bn_s, shift=bn_t)
                src_bytes = M * cin * 2

            def level_conv(name, g0, ng, out, top):
                """groups [g0, g0+ng) of level i; `top`: fused top convolution (out = T) instead of the 192-channel output."""
                wname = f"heads.L{i}f" if bn_fold else f"heads.L{i}"
                brows = 16 if bn_fold else 1
                f = dict(kind=L.OP_CONV, act=L.ACT_GELU, in_dtype=A, out_dtype=A, w_dtype=self.cdt, B=B, H=th_, W=tw_, Ho=th_, Wo=tw_,
                         Cin=cin, Cout=FPN_DIM, Cout_total=FPN_DIM, ksize=3, stride=1, groups=ng if ng > 1 else 0, out=out,
                         w=("w", self.pw.table[wname + ".w"] + g0 * wsz), bias=("w", self.pw.table[wname + ".b"] + g0 * brows * FPN_DIM * 4))
                flags = 0
                if bn_fold:
                    flags |= L.FLAG_UPCAT_IN | L.FLAG_BORDER_BIAS | L.FLAG_GROUP_IN2_SHARED
                    f.update(Cin_total=cy, aux0=65, in_=("buf", y[1], g0 * B * yh * yw * cy * 2), in2=tapbn)
                elif up_in:
                    flags |= L.FLAG_UPCAT_IN
                    f.update(Cin_total=cy, aux0=65, in_=("buf", y[1], g0 * B * yh * yw * cy * 2), in2=("buf", tapbn[1], g0 * M * tc * 2))
                else:
                    f.update(Cin_total=cin, in_=("buf", cat[1], g0 * M * cin * self.esize(A)))
                flops = 2.0 * ng * M * FPN_DIM * cin * 9
                byt = ng * (src_bytes + FPN_DIM * cin * 9 * self.esize(self.cdt))
                if top:
                    flags |= L.FLAG_TOP_FUSE
                    nout = sum(co for _, co, _ in HEADS[:-1])
                    f.update(aux0=65, aux1=TW, w2=self.wref("heads.top8.wt"))
                    flops += 2.0 * M * FPN_DIM * 9 * nout
                    byt += ng * M * TW * 4
                else:
                    byt += ng * M * FPN_DIM * self.esize(A)
                self.emit(OpMeta(name, "conv3x3", flops, byt), flags=flags, **f)

            if last and fuse_top:
                # Last level, bf16: the eight map heads never store their 192-channel output -- the epilogue multiplies the
                # tile by the head's top-convolution taps and stores 20 floats per pixel; TAPSUM does the 9-point sum into
                # the heat-map channels.  The feature head (100 output channels) keeps the two-kernel form.
                T = ("buf", self.buf(nmap * M * TW, L.F32))
                nout = sum(co for _, co, _ in HEADS[:-1])
                level_conv(f"heads.upsamplers.{i}+top", 0, nmap, T, True)
                self.emit(OpMeta("heads.top8.tapsum", "tapsum", 0.0, nmap * M * TW * 4 + M * nout * 4),
                          kind=L.OP_TAPSUM, B=B, H=th_, W=tw_, Ho=th_, Wo=tw_, Cout_total=10, aux0=TW, aux1=nout, groups=nmap,
                          in_=T, out=("heatmap", 0), w=self.wref("heads.top8.map"), bias=self.wref("heads.top8.b"))
                yf = ("buf", self.buf(M * FPN_DIM, A))
                level_conv(f"feature.upsamplers.{i}", nh - 1, 1, yf, False)
                self.conv("feature.top_conv", yf, A, th_, tw_, FPN_DIM, FPN_DIM, 0, "feature.top_conv", feature_dim, 3, 1, L.ACT_NONE,
                          ("features", 0), L.F32, cout_total=feature_dim, cout_off=0)
                y = None
                break
            ynew = ("buf", self.buf(nh * M * FPN_DIM, A))
            level_conv(f"heads.upsamplers.{i}", 0, nh, ynew, False)
            y = ynew
            yh, yw = th_, tw_
        gs = B * yh * yw * FPN_DIM * self.esize(A)            # bytes between the heads' last-level tensors
        for hi, (name, out_dim, ch0) in enumerate(HEADS if y is not None else []):
            yi = ("buf", y[1], hi * gs)
            if name in TOP6:
                if name != TOP6[0]:
                    continue                                   # covered by the grouped launch below
                self.conv("heads.top6", yi, A, yh, yw, FPN_DIM, FPN_DIM, 0, "heads.top6", 1, 3, 1, L.ACT_NONE, ("heatmap", 0), L.F32,
                          cout_total=10, cout_off=ch0 + 1, groups=len(TOP6), extra_flags=L.FLAG_GROUP_OUT_SLICE)
            elif ch0 >= 0:     # map heads write straight into their channel slice; channel 1 is the NMS slot
                off = 0 if ch0 == 0 else ch0 + 1
                self.conv(f"{name}.top_conv", yi, A, yh, yw, FPN_DIM, FPN_DIM, 0, f"{name}.top_conv", out_dim, 3, 1, L.ACT_NONE,
                          ("heatmap", 0), L.F32, cout_total=10, cout_off=off)
            else:
                self.conv(f"{name}.top_conv", yi, A, yh, yw, FPN_DIM, FPN_DIM, 0, f"{name}.top_conv", out_dim, 3, 1, L.ACT_NONE,
                          ("features", 0), L.F32, cout_total=feature_dim, cout_off=0)
        self.emit(OpMeta("nms", "nms", 0.0, B * mh * mw * 8.0), kind=L.OP_NMS, B=B, H=mh, W=mw, Ho=mh, Wo=mw, Cout_total=10,
                  out=("heatmap", 0))
        return self.finish(mh, mw)

    # --- arena allocation + ctypes ---------------------------------------------------------------
    def finish(self, mh: int, mw: int) -> Plan:
        order = sorted(range(len(self.bufs)), key=lambda i: self.bufs[i].first)
        live: List[Tuple[int, int, int]] = []        # (offset, end, last_use)
        top = peak = 0
        for bi in order:
            b = self.bufs[bi]
            if b.last < 0:
                raise RuntimeError("buffer never used")
            live = [iv for iv in live if iv[2] >= b.first]
            live.sort()
            off = 0
            for (o, e, _) in live:
                if off + b.nbytes <= o:
                    break
                off = max(off, e)
            b.offset = off
            live.append((off, off + b.nbytes, b.last))
            top = max(top, off + b.nbytes)
            peak = max(peak, sum(e - o for o, e, _ in live))
        ws = _align(top)
        arr = (L.Op * len(self.ops))()
        base_of = {"buf": L.BASE_WORKSPACE, "w": L.BASE_WEIGHTS, "input": L.BASE_INPUT, "heatmap": L.BASE_HEATMAP,
                   "features": L.BASE_FEATURES}
        for i, f in enumerate(self.ops):
            o = arr[i]
            for k, v in f.items():
                if k in ("in_", "in2", "out", "w", "w2", "bias", "bias2", "scale", "shift", "aux", "out2"):
                    if v is None:
                        continue
                    r = getattr(o, k)
                    r.base = base_of[v[0]]
                    r.offset = self.bufs[v[1]].offset + (v[2] if len(v) > 2 else 0) if v[0] == "buf" else v[1]
                else:
                    setattr(o, k, int(v))
        from . import tuning
        tuning.apply(arr)                       # measured kernel choice per conv shape (ftc_op.aux0)
        return Plan(arr, self.meta, ws, self.B, self.H, self.W, mh, mw, self.mode, None, peak,
                    sum(b.nbytes for b in self.bufs))


def build_plan(pw: PackedWeights, B: int, H: int, W: int, nchw_input: bool = False) -> Plan:
    if H % 32 or W % 32:
        raise ValueError("H and W must be multiples of 32 (the reference always uses 768)")
    return _Builder(pw, B, H, W, nchw_input).build()


def create_handle(plan: Plan, weights_bytes: int) -> int:
    """ftc_plan_create (host-only: validates shapes/offsets, copies the op list)."""
    lib = L.load()
    h = C.c_void_p()
    L.check(lib.ftc_plan_create(plan.ops, len(plan.ops), plan.workspace_bytes, weights_bytes, C.byref(h)), "ftc_plan_create")
    plan.handle = h.value
    return plan.handle
