"""Training-mode forward WITHOUT gradients of ``TextDetectorModel`` on MI355X -- the reference's end-of-epoch BN-refresh pass
(``train1.py:203-211``: ``train_step`` under ``torch.no_grad()`` with the model in ``train()``).

In ``train()`` every BatchNorm normalises with the statistics of the batch and moves its running statistics (momentum 0.1,
unbiased variance), and the residual branches pass torchvision's ``StochasticDepth("row")``.  Nothing can be folded, so this path
is a different op list over the same HIP kernels (``ftc_plan_create`` / ``ftc_plan_run``, ``include/ftc.h``):

    conv (raw weights, MFMA, no bias / activation)  ->  FTC_OP_BNSTAT (float64 batch statistics, running-stat update)
                                                    ->  FTC_OP_BNACT  (normalise, SiLU / GELU, keep-scale * branch + residual, SE sums)

Activations stay fp32 between the ops (the convolutions narrow their operands to the module's precision while staging them); the
op list is built here from the module's ``state_dict`` (the graph knowledge the reference keeps in Python as well), the arithmetic is
all in the library.  There is no backward pass: with gradients enabled ``TextDetectorModel.forward`` raises in ``train()`` mode.
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from .model import PRECISIONS, TORCH_DTYPE

BACKBONE_EPS, HEAD_EPS = 1e-3, 1e-5          # models/detector.py:27; nn.BatchNorm2d default (:161-184)
HEAD_NAMES = ["keyheatmap", "sizes", "textline", "sepatator", "code1", "code2", "code4", "code8"]
_STAGE_STRIDE = {1: 1, 2: 2, 3: 2, 4: 2, 5: 1, 6: 2, 7: 1}    # first-block stride per stage (models/detector.py:14-20, tv s/m/l tables)


def _fbits(v: float) -> int:
    return struct.unpack("<i", struct.pack("<f", v))[0]


def _align(n: int, a: int = 256) -> int:
    return (n + a - 1) // a * a


class _Buf:
    def __init__(self, nbytes: int):
        self.nbytes, self.first, self.last, self.offset = _align(nbytes), 1 << 30, -1, -1


class TrainForward:
    """One per (module, precision): packs the raw parameters once per parameter version, builds one plan per input shape."""

    def __init__(self, module, precision: str):
        self.module, self.precision = module, precision
        self.cdt = PRECISIONS["fp32" if precision == "fp16x3" else precision]      # (fp16x3 is an inference mode: the BN-refresh pass runs fp32)
        self.fingerprint = None
        self.wdev: Optional[torch.Tensor] = None
        self.table: Dict[str, int] = {}
        self.plans: Dict[Tuple[int, int, int], dict] = {}
        self.dec_plans: Dict[int, dict] = {}
        self.workspace: Optional[torch.Tensor] = None

    # ---- parameters -----------------------------------------------------------------------------------------------------
    def _fp(self):
        ts = list(self.module.parameters()) + list(self.module.buffers())
        return (sum(t._version for t in ts), sum(t.data_ptr() for t in ts))

    def _pack(self, dev) -> None:
        fp = self._fp()
        if self.wdev is not None and self.fingerprint == fp and self.wdev.device == dev:
            return
        sd = {k: v.detach() for k, v in self.module.state_dict().items()}
        self.sd_shapes = {k: tuple(v.shape) for k, v in sd.items()}
        tdt = TORCH_DTYPE[self.precision]                                           # (fp16x3 -> float32)
        items: List[Tuple[str, torch.Tensor]] = []
        cmax = 0
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                continue
            v = v.to(device=dev, dtype=torch.float32)
            if k.endswith(".running_mean"):
                p = k[: -len(".running_mean")]
                items.append((p + ".running", torch.stack([v, sd[p + ".running_var"].to(device=dev, dtype=torch.float32)]).contiguous()))
                cmax = max(cmax, v.numel())
            elif k.endswith(".running_var"):
                continue
            elif v.ndim == 4 and k == "detector.backbone.features.0.0.weight":
                items.append((k, v.permute(2, 3, 1, 0).reshape(27, -1).contiguous()))                    # stem: [(r*3+s)*3+c][C0] fp32
            elif v.ndim == 4 and v.shape[1] == 1 and v.shape[2] == 3 and ".block.1.0." in k:
                items.append((k, v.reshape(v.shape[0], 9).t().contiguous()))                              # depthwise: [9][C] fp32
            elif ".fc1.weight" in k:
                items.append((k, v.reshape(v.shape[0], v.shape[1]).contiguous()))                         # SE fc1: [S][C]
            elif ".fc2.weight" in k:
                items.append((k, v.reshape(v.shape[0], v.shape[1]).t().contiguous()))                     # SE fc2 transposed: [S][C]
            elif v.ndim == 4:
                items.append((k, v.permute(0, 2, 3, 1).reshape(v.shape[0], -1).to(tdt).contiguous()))     # conv: K-major, compute dtype
            elif v.ndim == 2:                                                                             # Linear: K padded to 128
                w = torch.zeros((v.shape[0], _align(v.shape[1], 128)), dtype=torch.float32, device=dev)
                w[:, : v.shape[1]] = v
                items.append((k, w.to(tdt).contiguous()))
            else:
                items.append((k, v.contiguous()))
        items.append(("zeros", torch.zeros(max(cmax, 4096), dtype=torch.float32, device=dev)))
        off, table = 0, {}
        for k, t in items:
            table[k] = off
            off = _align(off + t.numel() * t.element_size())
        blob = torch.zeros(off + 256, dtype=torch.uint8, device=dev)
        for k, t in items:
            raw = t.view(torch.int16).view(torch.uint8).reshape(-1) if t.dtype in (torch.bfloat16, torch.float16) else t.view(torch.uint8).reshape(-1)
            blob[table[k]: table[k] + raw.numel()] = raw
        self.wdev, self.table, self.fingerprint = blob, table, fp
        self._drop_plans()

    def _drop_plans(self) -> None:
        """The plans hold native memory (ftc_plan_create): destroy the handles, do not just forget them."""
        lib = L.load()
        for d in (self.plans, self.dec_plans):
            for pl in d.values():
                if pl.get("handle") is not None:
                    lib.ftc_plan_destroy(pl["handle"])
                    pl["handle"] = None
            d.clear()

    def __del__(self):
        try:
            self._drop_plans()
        except Exception:
            pass

    def _unpack_running_stats(self) -> None:
        """The kernels moved the running statistics inside the packed blob: copy them back into the module's buffers."""
        with torch.no_grad():
            dsts, srcs, counters = [], [], []
            for k, buf in self.module.named_buffers():
                if k.endswith("running_mean") or k.endswith("running_var"):
                    p = k.rsplit(".", 1)[0]
                    c = buf.numel()
                    o = self.table[p + ".running"] + (0 if k.endswith("running_mean") else 4 * c)
                    dsts.append(buf)
                    srcs.append(self.wdev[o: o + 4 * c].view(torch.float32).view_as(buf))
                elif k.endswith("num_batches_tracked"):
                    counters.append(buf)
            torch._foreach_copy_(dsts, srcs)                # one multi-tensor copy instead of ~1100 small ones
            torch._foreach_add_(counters, 1)
        self.fingerprint = self._fp()                      # the blob already holds these values

    # ---- op-list builder --------------------------------------------------------------------------------------------------
    class _G:
        def __init__(self, tf: "TrainForward", B: int):
            self.tf, self.B, self.ops, self.bufs = tf, B, [], []

        def buf(self, nbytes: int) -> _Buf:
            b = _Buf(nbytes)
            self.bufs.append(b)
            return b

        def w(self, name: str, extra: int = 0):
            return ("w", self.tf.table[name] + extra)

        def emit(self, **f) -> None:
            idx = len(self.ops)
            for v in f.values():
                if isinstance(v, tuple) and v[0] == "ws":
                    v[1].first, v[1].last = min(v[1].first, idx), max(v[1].last, idx)
            self.ops.append(f)

        # conv (raw weights) -> z fp32 [B,ho,wo,cout]
        def conv(self, x, h, w, cin, wname, cout, k, stride=1, se=None, bias=None, out=None, cout_total=None, cout_off=0):
            ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
            z = out if out is not None else ("ws", self.buf(self.B * ho * wo * cout * 4), 0)
            self.emit(kind=L.OP_CONV, flags=L.FLAG_SE_SCALE if se is not None else 0, act=L.ACT_NONE, in_dtype=L.F32, out_dtype=L.F32,
                      w_dtype=self.tf.cdt, B=self.B, H=h, W=w, Ho=ho, Wo=wo, Cin=cin, Cin_total=cin, Cout=cout, Cout_total=cout_total or cout,
                      cout_off=cout_off, ksize=k, stride=stride, res_dtype=L.F32, in_=x, out=z, w=self.w(wname), bias=bias or self.w("zeros"), scale=se)
            return z, ho, wo

        # BatchNorm with batch statistics (+ activation, keep-scale * branch + residual, SE sums) on z [B,h,w,c] fp32
        def bn(self, z, h, w, c, bn_name, eps, act, residual=None, keep=None, sums_p=0):
            M = self.B * h * w
            nchunk = max(1, min(512, -(-M // 64)))
            ss = ("ws", self.buf(4 * c * 4), 0)                 # scale | shift | mean | 1/std (FTC_OP_BNSTAT)
            part = ("ws", self.buf(nchunk * 2 * c * 8), 0)
            self.emit(kind=L.OP_BNSTAT, in_dtype=L.F32, B=self.B, H=h, W=w, Cin=c, aux0=_fbits(eps), aux1=_fbits(0.1), in_=z, w=self.w(bn_name + ".weight"),
                      bias=self.w(bn_name + ".bias"), aux=self.w(bn_name + ".running"), out=ss, in2=part)
            y = ("ws", self.buf(M * c * 4), 0)
            sums = ("ws", self.buf(self.B * sums_p * c * 4), 0) if sums_p else None
            # row chunks per image: few when the SE op has to read the partial sums, else enough workgroups to fill the GPU
            rows_p = sums_p if sums_p else max(1, min(2048, (h * w) // 64))
            self.emit(kind=L.OP_BNACT, flags=L.FLAG_RESIDUAL if residual is not None else 0, act=act, in_dtype=L.F32, out_dtype=L.F32,
                      w_dtype=L.F16 if self.tf.cdt == L.F16 else L.BF16, res_dtype=L.F32, B=self.B, H=h, W=w, Cin=c, aux0=rows_p, in_=z, scale=ss,
                      shift=("ws", ss[1], c * 4), in2=residual, w2=keep, out=y, aux=sums)
            return y, sums, ss

    def _build_detector(self, B: int, H: int, W: int) -> dict:
        g = TrainForward._G(self, B)
        sh = self.sd_shapes
        pre = "detector."
        P = pre + "backbone.features"
        res_names: List[str] = []
        keep_buf = g.buf(4096 * 4)                                   # [n residual blocks][B] fp32 keep-scales, filled before every run
        c0 = sh[P + ".0.0.weight"][0]
        h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        z = ("ws", g.buf(B * h * w * c0 * 4), 0)
        g.emit(kind=L.OP_STEM, act=L.ACT_NONE, in_dtype=L.F32, out_dtype=L.F32, B=B, H=H, W=W, Ho=h, Wo=w, Cin=3, Cout=c0, ksize=3, stride=2,
               in_=("in",), out=z, w=g.w(P + ".0.0.weight"), bias=g.w("zeros"))
        x, _, _ = g.bn(z, h, w, c0, P + ".0.1", BACKBONE_EPS, L.ACT_SILU)
        c = c0
        taps = []
        i = 1
        while f"{P}.{i}.0.block.0.0.weight" in sh:
            j = 0
            while f"{P}.{i}.{j}.block.0.0.weight" in sh:
                p = f"{P}.{i}.{j}"
                b = p + ".block"
                stride = _STAGE_STRIDE[i] if j == 0 else 1
                mb = f"{b}.2.fc1.weight" in sh
                fused4 = (not mb) and f"{b}.1.0.weight" in sh
                last = ".3" if mb else (".1" if fused4 else ".0")
                cout = sh[b + last + ".0.weight"][0]
                residual = x if (stride == 1 and c == cout) else None
                keep = None
                if residual is not None:
                    keep = ("ws", keep_buf, len(res_names) * _align(B, 4) * 4)       # rows padded to 16 bytes (operand alignment)
                    res_names.append(p[len(pre):])
                if mb:
                    e = sh[b + ".0.0.weight"][0]
                    z, _, _ = g.conv(x, h, w, c, b + ".0.0.weight", e, 1)
                    y, _, _ = g.bn(z, h, w, e, b + ".0.1", BACKBONE_EPS, L.ACT_SILU)
                    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
                    th = 8 if stride == 1 else 4
                    pdw = -(-ho // th) * -(-wo // 8)
                    zd = ("ws", g.buf(B * ho * wo * e * 4), 0)
                    g.emit(kind=L.OP_DWCONV, act=L.ACT_NONE, in_dtype=L.F32, out_dtype=L.F32, B=B, H=h, W=w, Ho=ho, Wo=wo, Cin=e, Cout=e, ksize=3,
                           stride=stride, aux0=pdw, in_=y, out=zd, w=g.w(b + ".1.0.weight"), bias=g.w("zeros"), aux=("ws", g.buf(B * pdw * e * 4), 0))
                    pse = max(1, min(16, (ho * wo) // 64))
                    y, sums, _ = g.bn(zd, ho, wo, e, b + ".1.1", BACKBONE_EPS, L.ACT_SILU, sums_p=pse)
                    s = sh[b + ".2.fc1.weight"][0]
                    sc = ("ws", g.buf(B * e * 4), 0)
                    g.emit(kind=L.OP_SE, B=B, H=ho, W=wo, Cin=e, Cout=e, aux0=s, aux1=pse, aux=sums, out=sc, in2=("ws", g.buf(B * s * 4), 0),
                           w=g.w(b + ".2.fc1.weight"), w2=g.w(b + ".2.fc2.weight"), bias=g.w(b + ".2.fc1.bias"), bias2=g.w(b + ".2.fc2.bias"))
                    z, _, _ = g.conv(y, ho, wo, e, b + ".3.0.weight", cout, 1, se=sc)
                    x, _, _ = g.bn(z, ho, wo, cout, b + ".3.1", BACKBONE_EPS, L.ACT_NONE, residual=residual, keep=keep)
                    h, w = ho, wo
                elif fused4:
                    e = sh[b + ".0.0.weight"][0]
                    z, h, w = g.conv(x, h, w, c, b + ".0.0.weight", e, 3, stride)
                    y, _, _ = g.bn(z, h, w, e, b + ".0.1", BACKBONE_EPS, L.ACT_SILU)
                    z, _, _ = g.conv(y, h, w, e, b + ".1.0.weight", cout, 1)
                    x, _, _ = g.bn(z, h, w, cout, b + ".1.1", BACKBONE_EPS, L.ACT_NONE, residual=residual, keep=keep)
                else:
                    z, h, w = g.conv(x, h, w, c, b + ".0.0.weight", cout, 3, stride)
                    x, _, _ = g.bn(z, h, w, cout, b + ".0.1", BACKBONE_EPS, L.ACT_SILU, residual=residual, keep=keep)
                c = cout
                j += 1
            if i in (2, 3, 5):
                taps.append((x, c, h, w))
            i += 1
        cl = sh[f"{P}.{i}.0.weight"][0]
        z, _, _ = g.conv(x, h, w, c, f"{P}.{i}.0.weight", cl, 1)
        x, _, _ = g.bn(z, h, w, cl, f"{P}.{i}.1", BACKBONE_EPS, L.ACT_SILU)
        taps.append((x, cl, h, w))
        mh, mw = taps[0][2], taps[0][3]
        maps = ("ws", g.buf(B * mh * mw * 9 * 4), 0)
        feats = ("ws", g.buf(B * mh * mw * 100 * 4), 0)
        ch = 0
        n = len(taps)
        for name in HEAD_NAMES + ["feature"]:
            hp = pre + name
            y, cy, yh, yw = None, 0, 0, 0
            for lvl, (tx, tc, th_, tw_) in enumerate(reversed(taps)):
                ti = n - 1 - lvl
                # the head's input BatchNorm of the tap: statistics here, the affine applied by the upsample+concat kernel
                M = B * th_ * tw_
                nchunk = max(1, min(512, -(-M // 64)))
                ss = ("ws", g.buf(4 * tc * 4), 0)
                g.emit(kind=L.OP_BNSTAT, in_dtype=L.F32, B=B, H=th_, W=tw_, Cin=tc, aux0=_fbits(HEAD_EPS), aux1=_fbits(0.1), in_=tx,
                       w=g.w(f"{hp}.in_bn.{ti}.weight"), bias=g.w(f"{hp}.in_bn.{ti}.bias"), aux=g.w(f"{hp}.in_bn.{ti}.running"), out=ss,
                       in2=("ws", g.buf(nchunk * 2 * tc * 8), 0))
                cat = ("ws", g.buf(M * (cy + tc) * 4), 0)
                g.emit(kind=L.OP_UPCAT, in_dtype=L.F32, out_dtype=L.F32, res_dtype=L.F32, B=B, H=yh if y is not None else th_, W=yw if y is not None else tw_,
                       Ho=th_, Wo=tw_, Cin=cy + tc, Cout=cy + tc, aux0=cy, aux1=tc, in_=y, in2=tx, out=cat, scale=ss, shift=("ws", ss[1], tc * 4))
                cm = sh[f"{hp}.upsamplers.{lvl}.0.weight"][0]
                z, _, _ = g.conv(cat, th_, tw_, cy + tc, f"{hp}.upsamplers.{lvl}.0.weight", cm, 3)
                y, _, _ = g.bn(z, th_, tw_, cm, f"{hp}.upsamplers.{lvl}.1", HEAD_EPS, L.ACT_GELU)
                cy, yh, yw = cm, th_, tw_
            co = sh[f"{hp}.top_conv.0.weight"][0]
            if name == "feature":
                g.conv(y, yh, yw, cy, f"{hp}.top_conv.0.weight", co, 3, bias=g.w(f"{hp}.top_conv.0.bias"), out=feats)
            else:
                g.conv(y, yh, yw, cy, f"{hp}.top_conv.0.weight", co, 3, bias=g.w(f"{hp}.top_conv.0.bias"), out=maps, cout_total=9, cout_off=ch)
                ch += co
        # keep the outputs and the keep-scales alive over the whole plan
        for bb in (maps[1], feats[1], keep_buf):
            bb.first, bb.last = 0, len(g.ops)
        plan = self._finish(g)
        plan.update(maps=maps[1], feats=feats[1], keep=keep_buf, res_names=res_names, mh=mh, mw=mw)
        return plan

    def _build_decoder(self, n: int) -> dict:
        g = TrainForward._G(self, 1)
        sh = self.sd_shapes
        rows = ("ws", g.buf(n * 128 * 4), 0)
        outs = []
        j = 0
        while f"decoder.blocks.{j}.0.weight" in sh:
            b = f"decoder.blocks.{j}"
            y, c = rows, 128
            for li, bi in ((0, 1), (3, 4)):
                co = sh[f"{b}.{li}.weight"][0]
                z, _, _ = g.conv(y, n, 1, c, f"{b}.{li}.weight", co, 1)
                y, _, _ = g.bn(z, n, 1, co, f"{b}.{bi}", HEAD_EPS, L.ACT_GELU)
                c = co
            co = sh[f"{b}.6.weight"][0]
            o = ("ws", g.buf(n * co * 4), 0)
            g.conv(y, n, 1, c, f"{b}.6.weight", co, 1, bias=g.w(f"{b}.6.bias"), out=o)
            outs.append((o[1], co))
            j += 1
        for bb in [rows[1]] + [o for o, _ in outs]:
            bb.first, bb.last = 0, len(g.ops)
        plan = self._finish(g)
        plan.update(rows=rows[1], outs=outs)
        return plan

    def _finish(self, g: "TrainForward._G") -> dict:
        """Liveness-based first-fit arena (as csrc/model.hip does for the inference plan), then ftc_plan_create."""
        order = sorted((b for b in g.bufs if b.last >= 0), key=lambda b: b.first)
        live: List[_Buf] = []
        top = 0
        for b in order:
            live = [x for x in live if x.last >= b.first]
            off = 0
            for x in sorted(live, key=lambda x: x.offset):
                if off + b.nbytes <= x.offset:
                    break
                off = max(off, x.offset + x.nbytes)
            b.offset = off
            live.append(b)
            top = max(top, off + b.nbytes)
        ops = (L.Op * len(g.ops))()
        for i, f in enumerate(g.ops):
            for k, v in f.items():
                if k in ("in_", "in2", "out", "w", "w2", "bias", "bias2", "scale", "shift", "aux", "out2"):
                    if v is None:
                        continue
                    r = getattr(ops[i], k)
                    if v[0] == "ws":
                        r.base, r.offset = L.BASE_WORKSPACE, v[1].offset + v[2]
                    elif v[0] == "w":
                        r.base, r.offset = L.BASE_WEIGHTS, v[1]
                    else:
                        r.base, r.offset = L.BASE_INPUT, 0
                else:
                    setattr(ops[i], k, int(v))
        h = C.c_void_p()
        L.check(L.load().ftc_plan_create(ops, len(g.ops), top + 256, self.wdev.numel(), C.byref(h)), "ftc_plan_create (training-mode forward)")
        return {"handle": h, "workspace_bytes": top + 256, "n_ops": len(g.ops)}

    def _run(self, plan: dict, x_ptr) -> None:
        dev = self.wdev.device
        if self.workspace is None or self.workspace.device != dev or self.workspace.numel() < plan["workspace_bytes"]:
            self.workspace = torch.empty(plan["workspace_bytes"], dtype=torch.uint8, device=dev)
        bases = (C.c_void_p * L.NUM_BASES)(None, self.workspace.data_ptr(), self.wdev.data_ptr(), x_ptr, None, None)
        L.check(L.load().ftc_plan_run(plan["handle"], bases, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), 0, -1), "ftc_plan_run (training-mode forward)")

    def _view(self, b: _Buf, shape) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        return self.workspace[b.offset: b.offset + 4 * n].view(torch.float32).reshape(shape)

    # ---- the forward --------------------------------------------------------------------------------------------------------
    def stochastic_depth_probs(self) -> Dict[str, float]:
        """torchvision EfficientNet.__init__: sd_prob = 0.2 * block_id / total_blocks over ALL blocks of the backbone."""
        blocks = []
        i = 1
        while f"detector.backbone.features.{i}.0.block.0.0.weight" in self.sd_shapes:
            j = 0
            while f"detector.backbone.features.{i}.{j}.block.0.0.weight" in self.sd_shapes:
                blocks.append(f"backbone.features.{i}.{j}")
                j += 1
            i += 1
        return {p: 0.2 * k / len(blocks) for k, p in enumerate(blocks)}

    def forward(self, x: torch.Tensor, fmask: torch.Tensor, keep: Optional[Dict[str, torch.Tensor]] = None, generator=None):
        """x [B,3,H,W] fp32 0..1 (NHWC memory behind the NCHW view, or NCHW-contiguous), fmask the boolean mask of get_fmask.
        keep: block prefix ("backbone.features.i.j") -> [B] keep-scales of StochasticDepth (missing blocks / None: drawn here with
        torch.rand(generator), as torchvision draws them).  Returns (maps [B,9,h,w], [three [n,modulo] logits]) and updates the module's
        BatchNorm running statistics."""
        from .loss_func import mask_to_index
        if not x.is_cuda:
            raise RuntimeError("findtextcenternet_amd: the training-mode forward runs on MI355X (gfx950) only (there is no CPU fallback)")
        dev = x.device
        x = x.float()
        B, _, H, W = x.shape
        if x.permute(0, 2, 3, 1).is_contiguous():
            xn = x.permute(0, 2, 3, 1)
        else:
            xn = x.permute(0, 2, 3, 1).contiguous()
        lib = L.load()
        with torch.cuda.device(dev):
            self._pack(dev)
            key = (B, H, W)
            if key not in self.plans:
                self.plans[key] = self._build_detector(B, H, W)
            plan = self.plans[key]
            n_res = len(plan["res_names"])
            Bp = _align(B, 4)
            if n_res * Bp > 4096:
                raise ValueError("batch too large for the keep-scale table")
            probs = self.stochastic_depth_probs()
            ks = torch.ones((n_res, B), dtype=torch.float32, device=dev)
            for r, name in enumerate(plan["res_names"]):
                kv = None if keep is None else keep.get(name, keep.get("detector." + name))
                if kv is not None:
                    ks[r] = kv.to(device=dev, dtype=torch.float32)
                elif keep is None:
                    surv = 1.0 - probs[name]
                    ks[r] = (torch.rand(B, device=dev, generator=generator) < surv).float() / surv
            if self.workspace is None or self.workspace.device != dev or self.workspace.numel() < plan["workspace_bytes"]:
                self.workspace = torch.empty(plan["workspace_bytes"], dtype=torch.uint8, device=dev)
            self._view(plan["keep"], (n_res, Bp))[:, :B].copy_(ks)
            self._run(plan, xn.data_ptr())
            mh, mw = plan["mh"], plan["mw"]
            maps = self._view(plan["maps"], (B, mh, mw, 9)).clone()
            feat = self._view(plan["feats"], (B, mh, mw, 100)).clone()
            # decoder on the masked rows (features.permute(0,2,3,1).flatten(0,-2)[fmask], models/detector.py:265-266)
            sel, cnt = mask_to_index(fmask)
            n = int(cnt.item())
            outs = [torch.empty((0, m), dtype=torch.float32, device=dev) for m in (1091, 1093, 1097)]
            if n > 1:                                                       # (BatchNorm1d in train() needs more than one row)
                if n not in self.dec_plans:
                    self.dec_plans[n] = self._build_decoder(n)
                dp = self.dec_plans[n]
                if self.workspace.numel() < dp["workspace_bytes"]:
                    self.workspace = torch.empty(dp["workspace_bytes"], dtype=torch.uint8, device=dev)
                rows = self._view(dp["rows"], (n, 128))
                L.check(lib.ftc_gather_rows(feat.data_ptr(), sel.data_ptr(), cnt.data_ptr(), n, 100, 128, rows.data_ptr(), L.F32,
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "ftc_gather_rows")
                self._run(dp, None)
                outs = [self._view(o, (n, co)).clone() for o, co in dp["outs"]]
            self._unpack_running_stats()
        return maps.permute(0, 3, 1, 2), outs
