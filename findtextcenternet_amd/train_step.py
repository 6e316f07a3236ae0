"""The reference's TRAIN step on MI355X: forward in ``train()`` mode, ``loss_function`` + ``CoVWeightingLoss``, and the backward pass that
the reference gets from ``loss.backward()`` (``/root/reference/train1.py:125-131, 170-179``; BASELINE configs[4]).

There is no autograd here.  The step is ONE static op list over the library's plan interface (``include/ftc.h``):

    detector forward (conv -> FTC_OP_BNSTAT -> FTC_OP_BNACT ...; every intermediate stays resident: 288 GB of HBM make recomputation
    pointless)  ->  FTC_OP_GATHER_ROWS  ->  SimpleDecoder forward  ->  FTC_OP_LOSSES
    [host: ftc_cov_weighting_step -- the alphas are DETACHED weights, loss_func.py:69-71]
    FTC_OP_LOSS_BWD  ->  decoder backward  ->  FTC_OP_SCATTER_ROWS  ->  nine heads backward  ->  backbone backward

with, per layer class:  data gradient of a dense conv = the forward implicit-GEMM kernel on flipped / transposed weights (stride 2 via
FTC_OP_DILATE), weight gradient = FTC_OP_WGRAD (MFMA, pixels are the contraction), batch-statistics BatchNorm + SiLU / GELU +
StochasticDepth = FTC_OP_BNBWD, depthwise = FTC_OP_DWBWD, SqueezeExcitation = FTC_OP_SEBWD, bilinear upsample + concat + per-head
input BatchNorm = FTC_OP_UPCATBWD + FTC_OP_BNBWD on a channel slice, top convolutions = FTC_OP_TOPDGRAD / FTC_OP_COLSUM.

Memory layout: every parameter of the module is re-pointed into ONE flat fp32 buffer and every ``.grad`` into a second one with the
same offsets (what a bucketed gradient all-reduce and the multi-tensor optimizer want); the BatchNorm running statistics live in the
same buffer as [mean | var] pairs so the kernels update them in place.  The layouts the MFMA kernels read (K-major forward weights,
flipped data-gradient weights, in the compute type) are re-derived from the flat parameters by ONE multi-tensor launch per step
(``ftc_pack_train_weights``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from .model import PRECISIONS
from .train_forward import BACKBONE_EPS, HEAD_EPS, HEAD_NAMES, _STAGE_STRIDE, _align, _Buf, _fbits

LOSS_KEYS = ["loss", "keymap_loss", "size_loss", "textline_loss", "separator_loss", "id_loss", "code1_loss", "code2_loss", "code4_loss",
             "code8_loss", "correct", "total"]
COV_KEYS = ["keymap_loss", "size_loss", "textline_loss", "separator_loss", "id_loss", "code1_loss", "code2_loss", "code4_loss", "code8_loss"]   # train1.py:107-114
DEC_PAD = 1104                    # padded logit row of the decoder gradients (>= 1097, multiple of 8)


class TrainStep:
    """``ts = TrainStep(model)``; per iteration ``loss, raw = ts.forward_backward(image, labelmap, idmap, fmask)`` then
    ``optimizer.step(); ts.zero_grad()`` -- the reference's loop body (train1.py:170-179) with ``train_step`` + ``backward`` fused."""

    def __init__(self, module, precision: Optional[str] = None, cov=None, two_streams: bool = True, decoder_only: bool = False,
                 z16: Optional[bool] = None):
        """two_streams: the backward's weight-gradient ops run on a second HIP stream beside the chain that produces their operands
        (ftc_plan_run_streams; same kernels, same results -- False keeps everything on the caller's stream).
        decoder_only: the reference's ``decoder_only`` switch (train1.py:98-101, 163-164): the detector is frozen and runs in eval mode
        (``module.detector.eval()``: the inference engine, running statistics, no StochasticDepth), only the SimpleDecoder trains -- the
        step is detector forward (eval) -> gather rows -> decoder forward (train) -> loss_function -> backward through the decoder;
        gradients of the detector's parameters stay zero (the reference sets requires_grad_(False) on them)."""
        self.module = module
        self.decoder_only = bool(decoder_only)
        # z16 (opt-in; FTC_TRAIN_Z16=1): the convolution outputs (the inputs of the batch-statistics BatchNorms) are STORED in the 16-bit
        # compute type, as the reference's autocast stores them (train1.py:127), so the statistics / normalise / backward passes of
        # every BatchNorm read half the bytes of z.  Measured (round 4, batch 8 x 768x768, bf16): 121.7 -> 119.8 ms per step -- those
        # passes are not bound by the z stream -- while the gradients move further from the fp32 reference (cosine min / p10 / median
        # 0.61 / 0.64 / 0.93 -> 0.16 / 0.53 / 0.91; still inside the envelope of the reference's OWN bf16 autocast, tests/golden/g11:
        # 0.07 / 0.32 / 0.86).  Not worth it by default.
        self._z16_arg = z16
        self.two_streams = bool(two_streams) and os.environ.get("FTC_TRAIN_ONE_STREAM") != "1"      # (env: A/B measurements)
        self.side_stream = None
        self.precision = precision or module.detector.precision
        if self.precision == "fp16x3":
            raise ValueError("TrainStep: 'fp16x3' is an inference mode; train in 'bf16' (what the reference's autocast does), 'fp16' or 'fp32'")
        self.cdt = PRECISIONS[self.precision]
        self.esz = 4 if self.cdt == L.F32 else 2
        env_z = os.environ.get("FTC_TRAIN_Z16")
        self.z16 = self.cdt != L.F32 and (self._z16_arg if self._z16_arg is not None else (env_z == "1"))
        self.cov = cov
        self.plans: Dict[Tuple[int, int, int], dict] = {}
        self.workspace: Optional[torch.Tensor] = None
        self.blob: Optional[torch.Tensor] = None
        self._flatten()

    # ---- flat parameters / gradients / derived layouts -----------------------------------------------------------------------
    def _flatten(self) -> None:
        params = list(self.module.named_parameters())
        dev = params[0][1].device          # (a CPU module can only BUILD plans -- ftc_plan_create validates without a GPU; the step itself raises)
        self.dev = dev
        sd_shapes = {k: tuple(v.shape) for k, v in self.module.state_dict().items()}
        self.sd_shapes = sd_shapes
        off = 0
        ptable: Dict[str, int] = {}
        for n, p in params:
            if p.dtype != torch.float32:
                raise TypeError("TrainStep keeps fp32 master parameters (the reference trains fp32 parameters under bf16 autocast)")
            ptable[n] = off
            off = _align(off + p.numel() * 4, 16)
        self.n_param_bytes = off
        stats: Dict[str, int] = {}
        cmax = 0
        for k, shp in sd_shapes.items():
            if k.endswith(".running_mean"):
                stats[k[: -len(".running_mean")] + ".running"] = off
                off = _align(off + 2 * shp[0] * 4, 16)
                cmax = max(cmax, shp[0])
        table = dict(ptable)
        table.update(stats)
        # derived layouts
        self.pack_specs: List[tuple] = []                     # (param name, fwd off | None, dgrad off | None, Cout, Cin, kk, cin_pad, cout_pad, dtype)
        off = _align(off, 256)
        for n, p in params:
            shp = tuple(p.shape)
            if n == "detector.backbone.features.0.0.weight":
                table[n + "#stem"] = off
                off = _align(off + p.numel() * 4, 256)
            elif p.ndim == 4 and shp[1] == 1 and shp[2] == 3 and ".block.1.0." in n:          # depthwise [C][1][3][3] -> [9][C] fp32
                self.pack_specs.append((n, None, off, shp[0], 9, 1, 9, shp[0], L.F32))
                table[n + "#dw"] = off
                off = _align(off + p.numel() * 4, 256)
            elif ".fc2.weight" in n:                                                          # [C][S] -> [S][C] fp32
                self.pack_specs.append((n, None, off, shp[0], shp[1], 1, shp[1], shp[0], L.F32))
                table[n + "#t"] = off
                off = _align(off + p.numel() * 4, 256)
            elif ".fc1.weight" in n:                                                          # [S][C][1][1] is the layout the kernels read
                q = n[: -len("fc1.weight")]
                S_, C_ = shp[0], shp[1]
                if not (ptable[q + "fc1.bias"] == ptable[n] + S_ * C_ * 4 and ptable[q + "fc2.weight"] == ptable[q + "fc1.bias"] + S_ * 4
                        and ptable[q + "fc2.bias"] == ptable[q + "fc2.weight"] + S_ * C_ * 4):
                    raise RuntimeError("TrainStep: the SqueezeExcitation parameters must be contiguous in the flat buffer (FTC_OP_SEBWD)")
                continue
            elif p.ndim == 4 or p.ndim == 2:
                co, ci = shp[0], shp[1]
                kk = shp[2] * shp[3] if p.ndim == 4 else 1
                cin_pad = _align(ci, 128) if p.ndim == 2 else ci                               # Linear: K padded to 128 (100 -> 128)
                cout_pad = _align(co, 32) if co % 8 else co                                     # feature top_conv: 100 -> 128
                if p.ndim == 2 and co in (1091, 1093, 1097):
                    cout_pad = DEC_PAD
                drows = cin_pad if p.ndim == 2 else ci
                f_off = off
                off = _align(off + co * kk * cin_pad * self.esz, 256)
                d_off = off
                off = _align(off + drows * kk * cout_pad * self.esz, 256)
                self.pack_specs.append((n, f_off, d_off, co, ci, kk, cin_pad, cout_pad, self.cdt))
                table[n + "#f"], table[n + "#d"] = f_off, d_off
        table["zeros"] = off
        off = _align(off + max(cmax, 4096) * 4, 256)
        table["ones_zeros"] = off                             # [ones | zeros]: an identity (scale, shift) pair (FTC_TRAIN_EMULATE_Z16 experiment)
        self._ones_n = max(cmax, 4096)
        off = _align(off + 2 * self._ones_n * 4, 256)
        self.table, self.ptable = table, ptable
        blob = torch.zeros(off + 256, dtype=torch.uint8, device=dev)
        grads = torch.zeros(self.n_param_bytes // 4, dtype=torch.float32, device=dev)
        fl = blob.view(torch.float32) if blob.numel() % 4 == 0 else blob[: blob.numel() // 4 * 4].view(torch.float32)
        fl[table["ones_zeros"] // 4: table["ones_zeros"] // 4 + self._ones_n] = 1.0
        with torch.no_grad():
            for n, p in params:
                o = ptable[n] // 4
                v = fl[o: o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                p.grad = grads[o: o + p.numel()].view(p.shape)
            mods = dict(self.module.named_modules())
            for k, o in stats.items():
                m = mods[k[: -len(".running")]]
                c = m.running_mean.numel()
                blk = fl[o // 4: o // 4 + 2 * c].view(2, c)
                blk[0].copy_(m.running_mean)
                blk[1].copy_(m.running_var)
                m.running_mean, m.running_var = blk[0], blk[1]
        self.blob, self.flat, self.grads = blob, fl, grads
        self.params = params
        self.counters = [b for k, b in self.module.named_buffers() if k.endswith("num_batches_tracked")]
        self.decoder_counters = [b for k, b in self.module.named_buffers() if k.endswith("num_batches_tracked") and k.startswith("decoder.")]
        self.stat_buffers = [b for k, b in self.module.named_buffers() if k.endswith("running_mean") or k.endswith("running_var")]
        # pack-entry table on the device
        ents = (L.PackEntry * len(self.pack_specs))()
        base = blob.data_ptr()
        pmap = dict(params)
        mx = 1
        for i, (n, f_off, d_off, co, ci, kk, cin_pad, cout_pad, dt) in enumerate(self.pack_specs):
            e = ents[i]
            e.src = pmap[n].data_ptr()
            e.fwd = base + f_off if f_off is not None else None
            e.dgrad = base + d_off if d_off is not None else None
            e.Cout, e.Cin, e.kk, e.cin_pad, e.cout_pad, e.dtype = co, ci, kk, cin_pad, cout_pad, dt
            mx = max(mx, co * kk * cin_pad + (ci * kk * cout_pad if d_off is not None else 0))
        raw = bytes(ents)
        self.pack_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.pack_max = mx
        self._stem_src = pmap["detector.backbone.features.0.0.weight"]
        self._param_ptrs = [p.data_ptr() for _, p in params]

    def __del__(self):
        try:
            lib = L.load()
            for pl in self.plans.values():
                if pl.get("handle") is not None:
                    lib.ftc_plan_destroy(pl["handle"])
                    pl["handle"] = None
        except Exception:
            pass

    def check_views(self) -> None:
        """The parameters must still live in the flat buffer (``module.to()`` / ``load_state_dict(assign=True)`` would move them)."""
        for (n, p), ptr in zip(self.params, self._param_ptrs):
            if p.data_ptr() != ptr:
                raise RuntimeError(f"TrainStep: parameter {n} no longer lives in the flat buffer; build a new TrainStep")

    def zero_grad(self) -> None:
        """``optimizer.zero_grad()`` as one memset; re-attaches the ``.grad`` views when an optimizer set them to None."""
        self.grads.zero_()
        self._grads_synced = False
        for n, p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + self.ptable[n]:
                o = self.ptable[n] // 4
                p.grad = self.grads[o: o + p.numel()].view(p.shape)

    def pack(self) -> None:
        """Derived layouts from the current flat parameters (after every optimizer step)."""
        s = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        L.check(L.load().ftc_pack_train_weights(self.pack_dev.data_ptr(), len(self.pack_specs), self.pack_max, s), "ftc_pack_train_weights")
        o = self.table["detector.backbone.features.0.0.weight#stem"] // 4
        w = self._stem_src
        with torch.no_grad():
            self.flat[o: o + w.numel()].view(27, -1).copy_(w.detach().permute(2, 3, 1, 0).reshape(27, -1))

    # ---- op-list builder -------------------------------------------------------------------------------------------------------
    class _G:
        """Op-list builder.  An activation is a pair (fp32 tensor | None, 16-bit copy | None): in the 16-bit modes the BatchNorm passes
        write a copy in the compute type next to (or instead of) the fp32 tensor, the GEMMs (forward conv, data gradient, weight gradient)
        read the copies -- half the operand bytes, DMA-staged kernels -- and fp32 stays where something other than a GEMM reads it
        (residual trunk, FPN taps, depthwise input, SE)."""

        def __init__(self, ts: "TrainStep", B: int):
            self.ts, self.B, self.ops, self.bufs, self.names = ts, B, [], [], []
            self.lib = L.load()
            self.h16 = ts.cdt != L.F32
            self.cdt = ts.cdt
            # weight gradients on the side stream (ftc_plan_run_streams): nothing on the backward chain reads them, so they overlap its
            # HBM-bound BatchNorm / depthwise passes.  An op's buffers stay allocated until the FTC_OP_JOIN after it (join()).
            self.side = ts.two_streams
            self.side_pending: List[int] = []

        def buf(self, nbytes: int) -> tuple:
            b = _Buf(nbytes)
            self.bufs.append(b)
            return ("ws", b, 0)

        def w(self, name: str, extra: int = 0):
            return ("w", self.ts.table[name] + extra)

        def g(self, name: str):
            return ("g", self.ts.ptable[name])

        def emit(self, _name: str = "", **f) -> None:
            idx = len(self.ops)
            for v in f.values():
                if isinstance(v, tuple) and v[0] == "ws":
                    v[1].first, v[1].last = min(v[1].first, idx), max(v[1].last, idx)
            self.ops.append(f)
            self.names.append(_name)

        def join(self) -> None:
            """FTC_OP_JOIN: the main stream waits for the side stream; every buffer a pending side op touches lives until here."""
            if not self.side_pending:
                return
            j = len(self.ops)
            self.ops.append(dict(kind=L.OP_JOIN, B=1, H=1, W=1, Ho=1, Wo=1))
            self.names.append("join")
            for i in self.side_pending:
                for v in self.ops[i].values():
                    if isinstance(v, tuple) and v[0] == "ws":
                        v[1].last = max(v[1].last, j)
            self.side_pending = []

        def side_op(self) -> None:
            """The op emitted next goes to the side stream: registered for the next join (emitted first when the oldest pending op is far behind)."""
            if self.side_pending and len(self.ops) - self.side_pending[0] >= int(os.environ.get("FTC_TRAIN_JOIN_EVERY", "32")):      # bounds how long operands outlive their last main-stream use
                self.join()
            self.side_pending.append(len(self.ops))

        def pin(self, ref) -> None:
            ref[1].first, ref[1].last = 0, 1 << 29

        def pick(self, act):
            """(operand, dtype) a GEMM reads for activation `act` = (fp32, copy16)."""
            return (act[1], self.cdt) if act[1] is not None else (act[0], L.F32)

        # ---- forward pieces
        def conv(self, x, h, w, cin, wname, cout, k, stride=1, se=None, bias=None, out=None, cout_total=None, cout_off=0, cin_total=None, B=None):
            B = B or self.B
            ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
            zdt = self.cdt if (self.ts.z16 and out is None and cout_total is None) else L.F32
            z = out if out is not None else self.buf(B * ho * wo * cout * (2 if zdt != L.F32 else 4))
            if out is None:
                z[1].dt = zdt                                     # the dtype travels with the buffer: BNSTAT / BNACT / BNBWD read it
            xin, xdt = self.pick(x)
            self.emit(wname, kind=L.OP_CONV, flags=L.FLAG_SE_SCALE if se is not None else 0, act=L.ACT_NONE, in_dtype=xdt, out_dtype=zdt,
                      w_dtype=self.cdt, B=B, H=h, W=w, Ho=ho, Wo=wo, Cin=cin, Cin_total=cin_total or cin, Cout=cout, Cout_total=cout_total or cout,
                      cout_off=cout_off, ksize=k, stride=stride, res_dtype=L.F32, in_=xin, out=z, w=self.w(wname + "#f"), bias=bias or self.w("zeros"), scale=se)
            if self.h16 and zdt == L.F32 and out is None and cout_total is None and os.environ.get("FTC_TRAIN_EMULATE_Z16") == "1":
                # EXPERIMENT (what would storing the conv outputs in 16 bits, as the reference's autocast does, cost in gradient agreement?):
                # round z to the compute type and back in place -- two extra identity passes, numerics of a 16-bit z, storage unchanged
                z16 = self.buf(B * ho * wo * cout * 2)
                idn = self.w("ones_zeros")
                ids = ("w", idn[1] + self.ts._ones_n * 4)
                rows_p = max(1, min(2048, (ho * wo) // 64))
                tc = L.F16 if self.cdt == L.F16 else L.BF16
                self.emit("z16:" + wname, kind=L.OP_BNACT, act=L.ACT_NONE, in_dtype=L.F32, out_dtype=self.cdt, w_dtype=tc, res_dtype=L.F32, B=B, H=ho, W=wo, Cin=cout,
                          aux0=rows_p, in_=z, scale=idn, shift=ids, out=z16)
                self.emit("z32:" + wname, kind=L.OP_BNACT, act=L.ACT_NONE, in_dtype=self.cdt, out_dtype=L.F32, w_dtype=tc, res_dtype=L.F32, B=B, H=ho, W=wo, Cin=cout,
                          aux0=rows_p, in_=z16, scale=idn, shift=ids, out=z)
            return z, ho, wo

        def bnstat(self, z, h, w, c, bn_name, eps, B=None):
            B = B or self.B
            M = B * h * w
            nchunk = max(1, min(512, -(-M // 64)))
            ss = self.buf(4 * c * 4)
            self.emit(bn_name, kind=L.OP_BNSTAT, in_dtype=getattr(z[1], "dt", L.F32), B=B, H=h, W=w, Cin=c, aux0=_fbits(eps), aux1=_fbits(0.1), in_=z, w=self.w(bn_name + ".weight"),
                      bias=self.w(bn_name + ".bias"), aux=self.w(bn_name + ".running"), out=ss, in2=self.buf(nchunk * 2 * c * 8))
            return ss

        def bn(self, z, h, w, c, bn_name, eps, act, residual=None, keep=None, sums_p=0, B=None, want32=True, want16=True):
            """-> ((y fp32 | None, y 16-bit | None), SE partial sums, statistics block)"""
            B = B or self.B
            ss = self.bnstat(z, h, w, c, bn_name, eps, B)
            want16 = want16 and self.h16
            want32 = want32 or not want16
            M = B * h * w
            y32 = self.buf(M * c * 4) if want32 else None
            y16 = self.buf(M * c * 2) if want16 else None
            sums = self.buf(B * sums_p * c * 4) if sums_p else None
            rows_p = sums_p if sums_p else max(1, min(2048, (h * w) // 64))
            self.emit(bn_name, kind=L.OP_BNACT, flags=L.FLAG_RESIDUAL if residual is not None else 0, act=act, in_dtype=getattr(z[1], "dt", L.F32),
                      out_dtype=L.F32 if want32 else self.cdt, w_dtype=L.F16 if self.cdt == L.F16 else L.BF16, res_dtype=L.F32, B=B, H=h, W=w, Cin=c,
                      aux0=rows_p, in_=z, scale=ss, shift=("ws", ss[1], c * 4), in2=residual, w2=keep, out=y32 if want32 else y16,
                      out2=y16 if (want32 and want16) else None, aux=sums)
            return (y32, y16), sums, ss

        # ---- backward pieces
        def bn_bwd(self, gy, z, ss, h, w, c, bn_name, act, keep=None, ga=None, gb=None, gy_total=0, gy_off=0, out=None, accum=False, B=None,
                   want32=False, want16=True):
            """-> (dz fp32 | None, dz 16-bit | None): dz of a convolution's BatchNorm is read by that convolution's two GEMMs only."""
            B = B or self.B
            M = B * h * w
            nchunk = max(1, min(512, -(-M // 64)))
            want16 = want16 and self.h16 and out is None
            want32 = want32 or not want16 or out is not None
            d32 = out if out is not None else (self.buf(M * c * 4) if want32 else None)
            d16 = self.buf(M * c * 2) if want16 else None
            self.emit("bwd:" + bn_name, kind=L.OP_BNBWD, flags=L.FLAG_ACCUM if accum else 0, act=act, w_dtype=self.cdt, in_dtype=getattr(z[1], "dt", L.F32), B=B, H=h, W=w, Cin=c,
                      Cin_total=gy_total, cin_off=gy_off, in_=gy, in2=z, scale=ss, w2=keep, bias=ga, bias2=gb, out=d32, out2=d16,
                      w=self.g(bn_name + ".weight"), shift=self.g(bn_name + ".bias"), aux=self.buf(nchunk * 2 * c * 8 + 2 * c * 4))
            return (d32, d16)

        def wgrad(self, x, dz, h, w, cin, cout, k, stride, wname, se=None, cin_total=0, cin_off=0, cout_total=0, cout_off=0, B=None):
            B = B or self.B
            ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
            S = int(self.lib.ftc_wgrad_splits(B, ho, wo, cout, cin, k))
            if se is not None and k == 1 and stride == 1:
                # SE-gated input (the project convolution): splits that each lie inside ONE image let the kernel apply the gate to the
                # columns of the partial tile instead of to every staged element (FTC_OP_WGRAD does so when Ho*Wo % chunk == 0)
                kk = max(1, round(S / B))
                while kk > 1 and (ho * wo) % (kk * 64):
                    kk -= 1
                if (ho * wo) % (kk * 64) == 0:
                    S = B * kk
            xin, xdt = self.pick(x)
            din, ddt = self.pick(dz)
            if self.side:
                self.side_op()
            self.emit("wgrad:" + wname, kind=L.OP_WGRAD, flags=(L.FLAG_SE_SCALE if se is not None else 0) | (L.FLAG_SIDE_STREAM if self.side else 0), w_dtype=self.cdt, in_dtype=xdt, res_dtype=ddt, B=B, H=h,
                      W=w, Ho=ho, Wo=wo, Cin=cin, Cin_total=cin_total, cin_off=cin_off, Cout=cout, Cout_total=cout_total, cout_off=cout_off, ksize=k,
                      stride=stride, aux0=S, in_=xin, in2=din, scale=se, out=self.g(wname), aux=self.buf(S * k * k * cout * cin * 4))

        def dgrad(self, dz, ho, wo, cout, wname, cin, k, stride, h, w, add=None, cout_pad=None, B=None):
            """d input fp32 [B,h,w,cin] of a convolution whose d output is dz = (fp32, 16-bit) [B,ho,wo,cout] (+ add)."""
            B = B or self.B
            src, sdt = self.pick(dz)
            if stride == 2:                                            # (the two stride-2 dense convs: fp32 through the dilation)
                assert dz[0] is not None
                src, sdt = self.buf(B * h * w * cout * 4), L.F32
                self.emit("dilate:" + wname, kind=L.OP_DILATE, B=B, H=ho, W=wo, Ho=h, Wo=w, Cin=cout, in_=dz[0], out=src)
            dx = self.buf(B * h * w * cin * 4)
            cp = cout_pad or cout
            self.emit("dgrad:" + wname, kind=L.OP_CONV, flags=L.FLAG_RESIDUAL if add is not None else 0, act=L.ACT_NONE, in_dtype=sdt, out_dtype=L.F32,
                      w_dtype=self.cdt, B=B, H=h, W=w, Ho=h, Wo=w, Cin=cp, Cin_total=cp, Cout=cin, Cout_total=cin, ksize=k, stride=1, res_dtype=L.F32,
                      in_=src, out=dx, w=self.w(wname + "#d"), bias=self.w("zeros"), in2=add)
            return dx

    def _build(self, B: int, H: int, W: int, loss_scale: float = 1.0) -> dict:
        g = TrainStep._G(self, B)
        sh = self.sd_shapes
        pre = "detector."
        P = pre + "backbone.features"
        n_rows = min(1024 * B, B * (H // 4) * (W // 4))
        F32ONLY = dict(want32=True, want16=False)
        GEMM_ONLY = dict(want32=False, want16=True)          # read by convolutions only: the 16-bit copy suffices (fp32 mode: fp32)
        # ---------------- forward ----------------
        res_names: List[str] = []
        keep_buf = g.buf(4096 * 4)
        g.pin(keep_buf)
        adt = g.cdt if g.h16 else L.F32                              # dtype of the FPN level tensors (read by convolutions / the upsampler only)
        aes = 2 if g.h16 else 4
        tape: List[dict] = []
        taps = []
        heads = []
        if self.decoder_only:
            # the detector ran in eval mode through the inference engine: its maps and features are INPUTS of this plan
            mh, mw = H // 4, W // 4
            maps = g.buf(B * mh * mw * 9 * 4)
            feats = g.buf(B * mh * mw * 100 * 4)
            g.pin(maps)
            g.pin(feats)
        else:
            c0 = sh[P + ".0.0.weight"][0]
            h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            z = g.buf(B * h * w * c0 * 4)
            g.emit("stem", kind=L.OP_STEM, act=L.ACT_NONE, in_dtype=L.F32, out_dtype=L.F32, B=B, H=H, W=W, Ho=h, Wo=w, Cin=3, Cout=c0, ksize=3, stride=2,
                   in_=("in",), out=z, w=g.w(P + ".0.0.weight#stem"), bias=g.w("zeros"))
            x, _, ss = g.bn(z, h, w, c0, P + ".0.1", BACKBONE_EPS, L.ACT_SILU)
            tape.append(dict(kind="stem", z=z, ss=ss, h=h, w=w, c=c0, out=x))
            c = c0
            pending_tap = None
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
                    residual = x[0] if (stride == 1 and c == cout) else None
                    keep = None
                    if residual is not None:
                        keep = ("ws", keep_buf[1], len(res_names) * _align(B, 4) * 4)
                        res_names.append(p[len(pre):])
                    rec = dict(b=b, xin=x, h=h, w=w, c=c, cout=cout, stride=stride, residual=residual is not None, keep=keep)
                    if pending_tap is not None:                              # this block reads a tap: the heads' gradient of the tap joins its data gradient
                        rec["xin_tap"], pending_tap = pending_tap, None
                    if mb:
                        e = sh[b + ".0.0.weight"][0]
                        z0, _, _ = g.conv(x, h, w, c, b + ".0.0.weight", e, 1)
                        y0, _, ss0 = g.bn(z0, h, w, e, b + ".0.1", BACKBONE_EPS, L.ACT_SILU, **F32ONLY)       # read by the fp32 depthwise kernels
                        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
                        th = 8 if stride == 1 else 4
                        pdw = -(-ho // th) * -(-wo // 8)
                        zd = g.buf(B * ho * wo * e * 4)
                        g.emit(b + ".1.0", kind=L.OP_DWCONV, act=L.ACT_NONE, in_dtype=L.F32, out_dtype=L.F32, B=B, H=h, W=w, Ho=ho, Wo=wo, Cin=e, Cout=e, ksize=3,
                               stride=stride, aux0=pdw, in_=y0[0], out=zd, w=g.w(b + ".1.0.weight#dw"), bias=g.w("zeros"), aux=g.buf(B * pdw * e * 4))
                        pse = max(1, min(16, (ho * wo) // 64))
                        y1, sums, ss1 = g.bn(zd, ho, wo, e, b + ".1.1", BACKBONE_EPS, L.ACT_SILU, sums_p=pse)                 # fp32 for the SE backward + copy
                        s = sh[b + ".2.fc1.weight"][0]
                        sc = g.buf(B * e * 4)
                        g.emit(b + ".2", kind=L.OP_SE, B=B, H=ho, W=wo, Cin=e, Cout=e, aux0=s, aux1=pse, aux=sums, out=sc, in2=g.buf(B * s * 4),
                               w=g.w(b + ".2.fc1.weight"), w2=g.w(b + ".2.fc2.weight#t"), bias=g.w(b + ".2.fc1.bias"), bias2=g.w(b + ".2.fc2.bias"))
                        z3, _, _ = g.conv(y1, ho, wo, e, b + ".3.0.weight", cout, 1, se=sc)
                        x, _, ss3 = g.bn(z3, ho, wo, cout, b + ".3.1", BACKBONE_EPS, L.ACT_NONE, residual=residual, keep=keep)
                        rec.update(kind="mb", e=e, z0=z0, y0=y0, ss0=ss0, zd=zd, y1=y1, ss1=ss1, sums=sums, pse=pse, s=s, sc=sc, z3=z3, ss3=ss3, ho=ho, wo=wo)
                        h, w = ho, wo
                    elif fused4:
                        e = sh[b + ".0.0.weight"][0]
                        z0, ho, wo = g.conv(x, h, w, c, b + ".0.0.weight", e, 3, stride)
                        y0, _, ss0 = g.bn(z0, ho, wo, e, b + ".0.1", BACKBONE_EPS, L.ACT_SILU, **GEMM_ONLY)
                        z1, _, _ = g.conv(y0, ho, wo, e, b + ".1.0.weight", cout, 1)
                        x, _, ss1 = g.bn(z1, ho, wo, cout, b + ".1.1", BACKBONE_EPS, L.ACT_NONE, residual=residual, keep=keep)
                        rec.update(kind="f4", e=e, z0=z0, y0=y0, ss0=ss0, z1=z1, ss1=ss1, ho=ho, wo=wo)
                        h, w = ho, wo
                    else:
                        z0, ho, wo = g.conv(x, h, w, c, b + ".0.0.weight", cout, 3, stride)
                        x, _, ss0 = g.bn(z0, ho, wo, cout, b + ".0.1", BACKBONE_EPS, L.ACT_SILU, residual=residual, keep=keep)
                        rec.update(kind="f1", z0=z0, ss0=ss0, ho=ho, wo=wo)
                        h, w = ho, wo
                    rec["out"] = x
                    tape.append(rec)
                    c = cout
                    j += 1
                if i in (2, 3, 5):
                    taps.append((x[0], c, h, w))
                    pending_tap = len(taps) - 1
                i += 1
            cl = sh[f"{P}.{i}.0.weight"][0]
            zl, _, _ = g.conv(x, h, w, c, f"{P}.{i}.0.weight", cl, 1)
            xl, _, ssl = g.bn(zl, h, w, cl, f"{P}.{i}.1", BACKBONE_EPS, L.ACT_SILU, **F32ONLY)               # the last tap: read by the heads' fp32 tap path only
            hc = dict(kind="headconv", name=f"{P}.{i}", xin=x, z=zl, ss=ssl, h=h, w=w, c=c, cout=cl, out=xl, tap=len(taps))
            if pending_tap is not None:
                hc["xin_tap"], pending_tap = pending_tap, None
            tape.append(hc)
            taps.append((xl[0], cl, h, w))
            mh, mw = taps[0][2], taps[0][3]
            maps = g.buf(B * mh * mw * 9 * 4)
            feats = g.buf(B * mh * mw * 100 * 4)
            g.pin(maps)
            ch = 0
            n = len(taps)
            for name in HEAD_NAMES + ["feature"]:
                hp = pre + name
                y, cy, yh, yw = None, 0, 0, 0
                levels = []
                for lvl, (tx, tc, th_, tw_) in enumerate(reversed(taps)):
                    ti = n - 1 - lvl
                    ssi = g.bnstat(tx, th_, tw_, tc, f"{hp}.in_bn.{ti}", HEAD_EPS)
                    catb = g.buf(B * th_ * tw_ * (cy + tc) * aes)
                    cat = (None, catb) if g.h16 else (catb, None)
                    g.emit(f"{hp}.upcat.{lvl}", kind=L.OP_UPCAT, in_dtype=adt, out_dtype=adt, res_dtype=L.F32, B=B, H=yh if y is not None else th_,
                           W=yw if y is not None else tw_, Ho=th_, Wo=tw_, Cin=cy + tc, Cout=cy + tc, aux0=cy, aux1=tc, in_=g.pick(y)[0] if y is not None else None,
                           in2=tx, out=catb, scale=ssi, shift=("ws", ssi[1], tc * 4))
                    cm = sh[f"{hp}.upsamplers.{lvl}.0.weight"][0]
                    zc, _, _ = g.conv(cat, th_, tw_, cy + tc, f"{hp}.upsamplers.{lvl}.0.weight", cm, 3)
                    yn, _, ssc = g.bn(zc, th_, tw_, cm, f"{hp}.upsamplers.{lvl}.1", HEAD_EPS, L.ACT_GELU, **GEMM_ONLY)
                    levels.append(dict(lvl=lvl, ti=ti, tx=tx, tc=tc, h=th_, w=tw_, cy=cy, yh=yh, yw=yw, ssi=ssi, cat=cat, cm=cm, z=zc, ss=ssc, y=yn))
                    y, cy, yh, yw = yn, cm, th_, tw_
                co = sh[f"{hp}.top_conv.0.weight"][0]
                if name == "feature":
                    g.conv(y, yh, yw, cy, f"{hp}.top_conv.0.weight", co, 3, bias=g.w(f"{hp}.top_conv.0.bias"), out=feats)
                    heads.append(dict(hp=hp, levels=levels, co=co, ch=None, y=y, cy=cy))
                else:
                    g.conv(y, yh, yw, cy, f"{hp}.top_conv.0.weight", co, 3, bias=g.w(f"{hp}.top_conv.0.bias"), out=maps, cout_total=9, cout_off=ch)
                    heads.append(dict(hp=hp, levels=levels, co=co, ch=ch, y=y, cy=cy))
                    ch += co
        # decoder on the selected rows
        sel = g.buf(n_rows * 4)
        lab = g.buf(B * 5 * mh * mw * 4)
        idm = g.buf(B * 2 * mh * mw * 4)
        lossv = g.buf(64)
        alphas = g.buf(64)
        for r in (sel, lab, idm, lossv, alphas):
            g.pin(r)
        rowsb = g.buf(n_rows * 128 * aes)
        rows = (None, rowsb) if g.h16 else (rowsb, None)
        g.emit("gather_rows", kind=L.OP_GATHER_ROWS, out_dtype=adt, B=B, H=mh, W=mw, Cin=100, Cout_total=128, aux0=n_rows, in_=feats, in2=sel, out=rowsb)
        dec = []
        jb = 0
        while f"decoder.blocks.{jb}.0.weight" in sh:
            bq = f"decoder.blocks.{jb}"
            yq, cq = rows, 128
            lay = []
            for li, bi in ((0, 1), (3, 4)):
                coq = sh[f"{bq}.{li}.weight"][0]
                zq, _, _ = g.conv(yq, n_rows, 1, cq, f"{bq}.{li}.weight", coq, 1, B=1)
                yn, _, ssq = g.bn(zq, n_rows, 1, coq, f"{bq}.{bi}", HEAD_EPS, L.ACT_GELU, B=1, **GEMM_ONLY)
                lay.append(dict(x=yq, cin=cq, z=zq, ss=ssq, y=yn, cout=coq, wname=f"{bq}.{li}.weight", bn=f"{bq}.{bi}"))
                yq, cq = yn, coq
            coq = sh[f"{bq}.6.weight"][0]
            oq = g.buf(n_rows * coq * 4)
            g.pin(oq)
            g.conv(yq, n_rows, 1, cq, f"{bq}.6.weight", coq, 1, bias=g.w(f"{bq}.6.bias"), out=oq, B=1)
            dec.append(dict(b=bq, lay=lay, x=yq, cin=cq, out=oq, cout=coq))
            jb += 1
        has_dec = len(dec) == 3
        g.emit("losses", kind=L.OP_LOSSES, B=B, H=mh, W=mw, aux0=n_rows if has_dec else 0, in_=maps, in2=lab, w=idm, w2=dec[0]["out"] if has_dec else None,
               bias=dec[1]["out"] if has_dec else None, bias2=dec[2]["out"] if has_dec else None, scale=sel if has_dec else None, out=lossv,
               aux=g.buf(int(g.lib.ftc_losses_scratch_bytes())))
        n_fwd = len(g.ops)
        # ---------------- backward ----------------
        gmaps = g.buf(B * mh * mw * 9 * 4)
        gdec = g.buf(3 * n_rows * DEC_PAD * 4) if has_dec else None
        lscale_slot = len(g.ops)
        g.emit("loss_bwd", kind=L.OP_LOSS_BWD, B=B, H=mh, W=mw, aux0=n_rows if has_dec else 0, aux1=DEC_PAD, Cout=_fbits(loss_scale), in_=maps, in2=lab, w=idm,
               w2=dec[0]["out"] if has_dec else None, bias=dec[1]["out"] if has_dec else None, bias2=dec[2]["out"] if has_dec else None,
               scale=sel if has_dec else None, shift=alphas, aux=lossv, out=gmaps, out2=gdec)
        gfeat = g.buf(B * mh * mw * 128 * 4)
        if has_dec:
            grows = None
            for jb, d in enumerate(dec):
                go = ("ws", gdec[1], jb * n_rows * DEC_PAD * 4)
                g.emit("bwd:" + d["b"] + ".6.bias", kind=L.OP_COLSUM, B=1, H=n_rows, W=1, Cin=d["cout"], Cin_total=DEC_PAD, in_=go, out=g.g(d["b"] + ".6.bias"),
                       aux=g.buf(max(1, min(512, -(-n_rows // 256))) * d["cout"] * 8))
                g.wgrad(d["x"], (go, None), n_rows, 1, d["cin"], d["cout"], 1, 1, d["b"] + ".6.weight", cout_total=DEC_PAD, B=1)
                gy = g.dgrad((go, None), n_rows, 1, d["cout"], d["b"] + ".6.weight", d["cin"], 1, 1, n_rows, 1, cout_pad=DEC_PAD, B=1)
                for li in (1, 0):
                    ly = d["lay"][li]
                    gz = g.bn_bwd(gy, ly["z"], ly["ss"], n_rows, 1, ly["cout"], ly["bn"], L.ACT_GELU, B=1)
                    if li == 1:
                        g.wgrad(ly["x"], gz, n_rows, 1, ly["cin"], ly["cout"], 1, 1, ly["wname"], B=1)
                        gy = g.dgrad(gz, n_rows, 1, ly["cout"], ly["wname"], ly["cin"], 1, 1, n_rows, 1, B=1)
                    else:
                        g.wgrad(ly["x"], gz, n_rows, 1, 100, ly["cout"], 1, 1, ly["wname"], cin_total=128, B=1)
                        grows = g.dgrad(gz, n_rows, 1, ly["cout"], ly["wname"], 128, 1, 1, n_rows, 1, add=grows, B=1)
            if not self.decoder_only:
                g.emit("scatter_rows", kind=L.OP_SCATTER_ROWS, B=B, H=mh, W=mw, Cout_total=128, aux0=n_rows, in_=grows, in2=sel, out=gfeat)
        elif not self.decoder_only:
            g.emit("fill", kind=L.OP_FILL, B=B, H=mh, W=mw, Cin=128, out=gfeat)
        # heads
        gtap = [g.buf(B * th_ * tw_ * tc * 4) for (_, tc, th_, tw_) in taps]
        tap_written = [False] * len(taps)
        nchunk_m = max(1, min(512, -(-(B * mh * mw) // 256)))
        for hd in reversed(heads):
            hp, co, cy = hd["hp"], hd["co"], hd["cy"]
            wn = f"{hp}.top_conv.0.weight"
            if hd["ch"] is None:
                g.emit("bwd:" + hp + ".top.bias", kind=L.OP_COLSUM, B=B, H=mh, W=mw, Cin=co, Cin_total=128, in_=gfeat, out=g.g(f"{hp}.top_conv.0.bias"),
                       aux=g.buf(nchunk_m * co * 8))
                g.wgrad(hd["y"], (gfeat, None), mh, mw, cy, co, 3, 1, wn, cout_total=128)
                gy = g.dgrad((gfeat, None), mh, mw, co, wn, cy, 3, 1, mh, mw, cout_pad=128)        # (_flatten pads the 100 feature channels to 128)
            else:
                g.emit("bwd:" + hp + ".top.bias", kind=L.OP_COLSUM, B=B, H=mh, W=mw, Cin=co, Cin_total=9, cin_off=hd["ch"], in_=gmaps,
                       out=g.g(f"{hp}.top_conv.0.bias"), aux=g.buf(nchunk_m * co * 8))
                g.wgrad(hd["y"], (gmaps, None), mh, mw, cy, co, 3, 1, wn, cout_total=9, cout_off=hd["ch"])
                gy = g.buf(B * mh * mw * cy * 4)
                g.emit("topdgrad:" + hp, kind=L.OP_TOPDGRAD, w_dtype=self.cdt, B=B, H=mh, W=mw, Cin=co, Cin_total=9, cin_off=hd["ch"], Cout=cy, in_=gmaps,
                       w=g.w(wn + "#f"), out=gy)
            for lv in reversed(hd["levels"]):
                lvl, ti, tc, lh, lw, lcy = lv["lvl"], lv["ti"], lv["tc"], lv["h"], lv["w"], lv["cy"]
                wn = f"{hp}.upsamplers.{lvl}.0.weight"
                gz = g.bn_bwd(gy, lv["z"], lv["ss"], lh, lw, lv["cm"], f"{hp}.upsamplers.{lvl}.1", L.ACT_GELU)
                g.wgrad(lv["cat"], gz, lh, lw, lcy + tc, lv["cm"], 3, 1, wn)
                gcat = g.dgrad(gz, lh, lw, lv["cm"], wn, lcy + tc, 3, 1, lh, lw)
                g.bn_bwd(gcat, lv["tx"], lv["ssi"], lh, lw, tc, f"{hp}.in_bn.{ti}", L.ACT_NONE, gy_total=lcy + tc, gy_off=lcy, out=gtap[ti], accum=tap_written[ti])
                tap_written[ti] = True
                if lcy > 0:
                    gy = g.buf(B * lv["yh"] * lv["yw"] * lcy * 4)
                    g.emit(f"upcatbwd:{hp}.{lvl}", kind=L.OP_UPCATBWD, B=B, H=lv["yh"], W=lv["yw"], Ho=lh, Wo=lw, Cin_total=lcy + tc, aux0=lcy, in_=gcat, out=gy)
        # backbone
        gx = None
        for rec in reversed(tape):
            kind = rec["kind"]
            tap_add = gtap[rec["xin_tap"]] if "xin_tap" in rec else None    # the layer's INPUT is a tap: the heads' gradient joins here
            if kind == "headconv":
                gout = gtap[rec["tap"]]                                    # its output is the last tap: only the heads read it
                gz = g.bn_bwd(gout, rec["z"], rec["ss"], rec["h"], rec["w"], rec["cout"], rec["name"] + ".1", L.ACT_SILU)
                g.wgrad(rec["xin"], gz, rec["h"], rec["w"], rec["c"], rec["cout"], 1, 1, rec["name"] + ".0.weight")
                gx = g.dgrad(gz, rec["h"], rec["w"], rec["cout"], rec["name"] + ".0.weight", rec["c"], 1, 1, rec["h"], rec["w"], add=tap_add)
                continue
            if kind == "stem":
                gz = g.bn_bwd(gx, rec["z"], rec["ss"], rec["h"], rec["w"], rec["c"], P + ".0.1", L.ACT_SILU, want32=True, want16=False)
                g.emit("stemwgrad", kind=L.OP_STEMWGRAD, B=B, H=H, W=W, Ho=rec["h"], Wo=rec["w"], Cout=rec["c"], in_=("in",), in2=gz[0], out=g.g(P + ".0.0.weight"),
                       aux=g.buf(max(1, min(2048, -(-(B * rec["h"] * rec["w"]) // 256))) * 27 * rec["c"] * 8))
                continue
            gout = gx
            b, h_, w_, c_, cout, stride, ho, wo = rec["b"], rec["h"], rec["w"], rec["c"], rec["cout"], rec["stride"], rec["ho"], rec["wo"]
            assert not (rec["residual"] and tap_add is not None)
            skip = gout if rec["residual"] else tap_add
            s2 = dict(want32=True) if stride == 2 else {}                # a stride-2 data gradient goes through the fp32 dilation
            if kind == "mb":
                e = rec["e"]
                gz3 = g.bn_bwd(gout, rec["z3"], rec["ss3"], ho, wo, cout, b + ".3.1", L.ACT_NONE, keep=rec["keep"])
                g.wgrad(rec["y1"], gz3, ho, wo, e, cout, 1, 1, b + ".3.0.weight", se=rec["sc"])
                gys = g.dgrad(gz3, ho, wo, cout, b + ".3.0.weight", e, 1, 1, ho, wo)
                s = rec["s"]
                scr = g.buf((36 * B * e + 2 * B * s) * 4)
                g.emit("sebwd:" + b, kind=L.OP_SEBWD, B=B, H=ho, W=wo, Cin=e, aux0=s, aux1=rec["pse"], in_=gys, in2=rec["y1"][0], scale=rec["sc"], aux=rec["sums"],
                       w=g.w(b + ".2.fc1.weight"), w2=g.w(b + ".2.fc2.weight#t"), bias=g.w(b + ".2.fc1.bias"), bias2=g.w(b + ".2.fc2.bias"), out=scr,
                       out2=g.g(b + ".2.fc1.weight"))
                gzd = g.bn_bwd(gys, rec["zd"], rec["ss1"], ho, wo, e, b + ".1.1", L.ACT_SILU, ga=rec["sc"], gb=("ws", scr[1], 3 * B * e * 4), want32=True, want16=False)
                gy0 = g.buf(B * h_ * w_ * e * 4)
                dw_aux = g.buf(max(1, min(512, -(-(B * ho * wo) // 256))) * 9 * e * 8)
                if g.side:
                    # the weight half on the side stream, like the dense weight gradients (nothing on the chain reads it); the data half stays
                    g.emit("dwbwd:" + b, kind=L.OP_DWBWD, B=B, H=h_, W=w_, Ho=ho, Wo=wo, Cin=e, stride=stride, in2=gzd[0], w=g.w(b + ".1.0.weight#dw"), out=gy0)
                    g.side_op()
                    g.emit("dwwgrad:" + b, kind=L.OP_DWBWD, flags=L.FLAG_SIDE_STREAM, B=B, H=h_, W=w_, Ho=ho, Wo=wo, Cin=e, stride=stride, in_=rec["y0"][0], in2=gzd[0],
                           out2=g.g(b + ".1.0.weight"), aux=dw_aux)
                else:
                    g.emit("dwbwd:" + b, kind=L.OP_DWBWD, B=B, H=h_, W=w_, Ho=ho, Wo=wo, Cin=e, stride=stride, in_=rec["y0"][0], in2=gzd[0], w=g.w(b + ".1.0.weight#dw"),
                           out=gy0, out2=g.g(b + ".1.0.weight"), aux=dw_aux)
                gz0 = g.bn_bwd(gy0, rec["z0"], rec["ss0"], h_, w_, e, b + ".0.1", L.ACT_SILU)
                g.wgrad(rec["xin"], gz0, h_, w_, c_, e, 1, 1, b + ".0.0.weight")
                gx = g.dgrad(gz0, h_, w_, e, b + ".0.0.weight", c_, 1, 1, h_, w_, add=skip)
            elif kind == "f4":
                e = rec["e"]
                gz1 = g.bn_bwd(gout, rec["z1"], rec["ss1"], ho, wo, cout, b + ".1.1", L.ACT_NONE, keep=rec["keep"])
                g.wgrad(rec["y0"], gz1, ho, wo, e, cout, 1, 1, b + ".1.0.weight")
                gy0 = g.dgrad(gz1, ho, wo, cout, b + ".1.0.weight", e, 1, 1, ho, wo)
                gz0 = g.bn_bwd(gy0, rec["z0"], rec["ss0"], ho, wo, e, b + ".0.1", L.ACT_SILU, **s2)
                g.wgrad(rec["xin"], gz0, h_, w_, c_, e, 3, stride, b + ".0.0.weight")
                gx = g.dgrad(gz0, ho, wo, e, b + ".0.0.weight", c_, 3, stride, h_, w_, add=skip)
            else:
                gz0 = g.bn_bwd(gout, rec["z0"], rec["ss0"], ho, wo, cout, b + ".0.1", L.ACT_SILU, keep=rec["keep"], **s2)
                g.wgrad(rec["xin"], gz0, h_, w_, c_, cout, 3, stride, b + ".0.0.weight")
                gx = g.dgrad(gz0, ho, wo, cout, b + ".0.0.weight", c_, 3, stride, h_, w_, add=skip)
        g.join()
        plan = self._finish(g)
        plan.update(maps=maps[1], feats=feats[1], keep=keep_buf[1], res_names=res_names, mh=mh, mw=mw, sel=sel[1], lab=lab[1], idm=idm[1], lossv=lossv[1], alphas=alphas[1],
                    n_rows=n_rows, n_fwd=n_fwd, loss_bwd_op=lscale_slot, names=g.names, dec_outs=[d["out"][1] for d in dec])
        return plan

    def _finish(self, g: "TrainStep._G") -> dict:
        order = sorted((b for b in g.bufs if b.last >= 0), key=lambda b: b.first)
        live: List[_Buf] = []
        top = 0
        n_ops = len(g.ops)
        for b in order:
            b.last = min(b.last, n_ops)
            live = [x for x in live if x.last >= b.first]
            off = 0
            for x in sorted(live, key=lambda x: x.offset):
                if off + b.nbytes <= x.offset:
                    break
                off = max(off, x.offset + x.nbytes)
            b.offset = off
            live.append(b)
            top = max(top, off + b.nbytes)
        ops = (L.Op * n_ops)()
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
                    elif v[0] == "g":
                        r.base, r.offset = L.BASE_GRADS, v[1]
                    else:
                        r.base, r.offset = L.BASE_INPUT, 0
                else:
                    setattr(ops[i], k, int(v))
        from . import tuning
        tuning.apply(ops)                                   # measured kernel choice per conv signature (tuning_gfx950.json; FTC_NO_TUNING=1: heuristics)
        h = C.c_void_p()
        L.check(L.load().ftc_plan_create(ops, n_ops, top + 256, self.blob.numel(), C.byref(h)), "ftc_plan_create (train step)")
        return {"handle": h, "workspace_bytes": top + 256, "n_ops": n_ops, "ops": ops}

    # ---- the step --------------------------------------------------------------------------------------------------------------
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

    def _view(self, b: _Buf, shape, dtype=torch.float32) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        return self.workspace[b.offset: b.offset + 4 * n].view(dtype).reshape(shape)

    def plan_for(self, B: int, H: int, W: int, loss_scale: float = 1.0) -> dict:
        key = (B, H, W, float(loss_scale))
        if key not in self.plans:
            self.plans[key] = self._build(B, H, W, float(loss_scale))
        return self.plans[key]

    def forward_backward(self, image: torch.Tensor, labelmap: torch.Tensor, idmap: torch.Tensor, fmask: Optional[torch.Tensor] = None,
                         keep: Optional[Dict[str, torch.Tensor]] = None, generator=None, loss_scale: float = 1.0, alphas: Optional[torch.Tensor] = None,
                         backward: bool = True, sync_grads: bool = True):
        """image [B,3,H,W] fp32 0..1 (NHWC memory behind the NCHW view, or NCHW-contiguous); labelmap [B,5,h,w]; idmap [B,2,h,w]; fmask =
        ``model.get_fmask(labelmap, fmask)`` (computed here when None).  Runs the train()-mode forward, loss_function, the CoV weighting
        (``self.cov``; or explicit ``alphas`` [9]) and ADDS d(loss * loss_scale)/d(parameter) to every ``.grad``.
        Returns (loss, rawloss dict) as 0-d device tensors -- what the reference's ``train_step`` returns (train1.py:125-131).

        Gradient accumulation under ``enable_ddp`` (the reference's ``iters_to_accumulate``, train1.py:176-179): the all-reduce SUMS the
        whole accumulated buffer in place, so pass ``sync_grads=False`` on every micro-batch but the LAST before ``optimizer.step()``;
        a second synchronising call without ``zero_grad()`` in between would re-sum already-reduced gradients and raises.
        A caller-supplied ``fmask`` must select exactly the plan's row count (min(1024 * B, B * h * w) pixels, what ``get_fmask`` selects)."""
        st = self._begin(image, fmask if fmask is not None else self.module.get_fmask(labelmap, None), keep, generator, loss_scale, check_mask=fmask is not None)
        lib = L.load()
        dev, plan = st["dev"], st["plan"]
        with torch.cuda.device(dev):
            self._set_labels(st, labelmap, idmap)
            L.check(lib.ftc_plan_run(plan["handle"], st["bases"], st["stream"], 0, plan["n_fwd"] - 1), "ftc_plan_run (train step, forward)")
            lossv = self._view(plan["lossv"], (16,))
            raw = {k: lossv[i] for i, k in enumerate(LOSS_KEYS)}
            if alphas is not None:
                a = alphas.to(device=dev, dtype=torch.float32)
                loss = (a * lossv[1:10]).sum()
            else:
                if self.cov is None:
                    from .loss_func import CoVWeightingLoss
                    self.cov = CoVWeightingLoss(device=dev, losses=COV_KEYS)
                loss = self.cov(raw)
                a = self.cov.alphas
            if backward:
                self._backward(st, a, sync_grads)
            self._end(st)
        return loss, {k: v.clone() for k, v in raw.items()}

    # ---- the phases of a step (forward_backward above = all of them; the train1.py-callable seam below runs them one by one) ----------
    def _begin(self, image: torch.Tensor, fmask: torch.Tensor, keep, generator, loss_scale: float, check_mask: bool = True) -> dict:
        """Everything in front of the plan run except the labels: plan + arena, selected-pixel indices, the frozen detector's maps
        (decoder_only), the StochasticDepth draw, the weight re-pack."""
        if not image.is_cuda:
            raise RuntimeError("findtextcenternet_amd: the train step runs on MI355X (gfx950) only (there is no CPU fallback)")
        # `optimizer.zero_grad()` (set_to_none=True, torch's default: train1.py:167, 179) drops the .grad views, and the kernels would go on adding into
        # a buffer no optimizer sees: a dropped gradient restarts at zero and is re-attached
        missing = [(n, p) for n, p in self.params if p.grad is None and p.requires_grad]
        if missing and len(missing) == sum(1 for _, p in self.params if p.requires_grad):
            self.zero_grad()
        elif missing:
            for n, p in missing:
                o = self.ptable[n] // 4
                self.grads[o: o + p.numel()].zero_()
                p.grad = self.grads[o: o + p.numel()].view(p.shape)
            self._grads_synced = False
        dev = image.device
        x = image.float()
        B, _, H, W = x.shape
        xn = x.permute(0, 2, 3, 1)
        if not xn.is_contiguous():
            xn = xn.contiguous()
        with torch.cuda.device(dev):
            ddp = getattr(self, "ddp", None)
            world = ddp.world if ddp is not None else 1
            plan = self.plan_for(B, H, W, loss_scale / world)
            if self.workspace is None or self.workspace.numel() < plan["workspace_bytes"]:
                self.workspace = torch.empty(plan["workspace_bytes"], dtype=torch.uint8, device=dev)
            mh, mw, n_rows = plan["mh"], plan["mw"], plan["n_rows"]
            from .loss_func import mask_to_index
            sel, _cnt = mask_to_index(fmask)
            if check_mask and int(_cnt.reshape(-1)[0].item()) != n_rows:       # (one host sync, only for masks this step did not compute)
                raise ValueError(f"forward_backward: fmask selects {int(_cnt.reshape(-1)[0].item())} pixels, the plan gathers {n_rows} rows "
                                 "(rows beyond the count would be read from uninitialised indices)")
            self._view(plan["sel"], (n_rows,), torch.int32).copy_(sel[:n_rows])
            if self.decoder_only:
                # the frozen detector in eval mode (train1.py:98-101): the inference engine's maps and features feed the plan
                if self.module.detector.training:
                    raise RuntimeError("TrainStep(decoder_only=True): put the detector in eval mode (model.detector.eval()), as train1.py:163-164 does")
                with torch.no_grad():
                    heat, feat = self.module.detector.forward_nhwc(x, with_nms=False)
                idx9 = torch.tensor([0, 2, 3, 4, 5, 6, 7, 8, 9], device=dev)
                self._view(plan["maps"], (B, mh, mw, 9)).copy_(heat.index_select(3, idx9))
                self._view(plan["feats"], (B, mh, mw, 100)).copy_(feat)
            # StochasticDepth draw (torchvision "row" mode)
            n_res = len(plan["res_names"])
            Bp = _align(B, 4)
            if n_res * Bp > 4096:
                raise ValueError("batch too large for the keep-scale table")
            probs = self.stochastic_depth_probs() if n_res else {}
            ks = torch.ones((n_res, B), dtype=torch.float32, device=dev)
            if n_res == 0:
                pass                                               # decoder_only: the frozen detector has no StochasticDepth draw
            elif keep is None:
                surv = torch.tensor([1.0 - probs[nm] for nm in plan["res_names"]], dtype=torch.float32, device=dev).reshape(-1, 1)
                ks = (torch.rand((n_res, B), device=dev, generator=generator) < surv).float() / surv
            else:
                for r, name in enumerate(plan["res_names"]):
                    kv = keep.get(name, keep.get("detector." + name))
                    if kv is not None:
                        ks[r] = kv.to(device=dev, dtype=torch.float32)
            if n_res:
                self._view(plan["keep"], (n_res, Bp))[:, :B].copy_(ks)
            self.pack()
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            bases = (C.c_void_p * L.NUM_BASES)(None, self.workspace.data_ptr(), self.blob.data_ptr(), xn.data_ptr(), None, None, self.grads.data_ptr())
        return dict(dev=dev, plan=plan, B=B, H=H, W=W, xn=xn, bases=bases, stream=stream, world=world, ddp=ddp, loss_scale=loss_scale)

    def _set_labels(self, st: dict, labelmap: torch.Tensor, idmap: torch.Tensor) -> None:
        plan, B = st["plan"], st["B"]
        self._view(plan["lab"], (B, 5, plan["mh"], plan["mw"])).copy_(labelmap.to(torch.float32))
        self._view(plan["idm"], (B, 2, plan["mh"], plan["mw"]), torch.int32).copy_(idmap.to(torch.int32))

    def _backward(self, st: dict, a: torch.Tensor, sync_grads: bool = True) -> None:
        """a = the nine loss weights (the detached CoV alphas, or d loss / d loss_i of whatever the caller built): the backward ops of the plan."""
        lib = L.load()
        dev, plan, ddp, world = st["dev"], st["plan"], st["ddp"], st["world"]
        bases, stream = st["bases"], st["stream"]
        self._view(plan["alphas"], (9,)).copy_(a.to(device=dev, dtype=torch.float32))
        side = None
        if self.two_streams:
            if self.side_stream is None or self.side_stream.device != dev:
                self.side_stream = torch.cuda.Stream(device=dev)
            side = C.c_void_p(self.side_stream.cuda_stream)
        if ddp is None or world == 1 or not sync_grads:
            L.check(lib.ftc_plan_run_streams(plan["handle"], bases, stream, side, plan["n_fwd"], -1), "ftc_plan_run_streams (train step, backward)")
            return
        if getattr(self, "_grads_synced", False):
            raise RuntimeError("forward_backward(sync_grads=True) twice without zero_grad(): the gradient buffer already holds the all-reduced "
                               "sum; accumulate with sync_grads=False on all but the last micro-batch")
        self._grads_synced = True
        key = (st["B"], st["H"], st["W"], float(st["loss_scale"] / world))
        if key not in self._segments:
            self._segments[key] = self._bucket_segments(plan)
        cur = torch.cuda.current_stream(dev)
        for first, last, bi in self._segments[key]:
            if first <= last:
                L.check(lib.ftc_plan_run_streams(plan["handle"], bases, stream, side, first, last), "ftc_plan_run_streams (train step, backward segment)")
            if bi is not None:
                self.comm_stream.wait_stream(cur)                 # the bucket is complete once everything enqueued so far has run
                with torch.cuda.stream(self.comm_stream):
                    ddp.reduce_bucket(bi, async_op=True)
        ddp.wait()
        cur.wait_stream(self.comm_stream)

    def _end(self, st: dict) -> None:
        torch._foreach_add_(self.counters if not self.decoder_only else self.decoder_counters, 1)
        from .optim import bump_versions
        bump_versions(self.stat_buffers)           # the kernels moved the running statistics in place

    # ---- the reference's own calling sequence ---------------------------------------------------------------------------------------
    # train1.py:125-131 + :174-179 --   heatmap, decoder_outputs = model(image, fmask);  rawloss = loss_function(fmask, map, idmap, heatmap,
    # decoder_outputs);  loss = CoWloss(rawloss);  (loss / iters_to_accumulate).backward();  optimizer.step()
    # -- runs the SAME plan in three pieces: TextDetectorModel.forward (train mode, grad enabled) -> seam_forward (every op but the loss op),
    # loss_function -> seam_losses (labels in, the loss op; the returned values carry an autograd node), CoVWeightingLoss -> a differentiable
    # sum with detached alphas, .backward() -> seam_backward with d loss / d loss_i as the nine weights.  Same kernels, same arithmetic as
    # forward_backward (tests/test_gpu_train_step.py::test_reference_calling_sequence_equals_forward_backward).
    def seam_forward(self, image: torch.Tensor, fmask: torch.Tensor, keep=None, generator=None):
        lib = L.load()
        st = self._begin(image, fmask, keep if keep is not None else self.module.__dict__.get("stochastic_depth_keep"), generator, 1.0)
        plan, dev, B = st["plan"], st["dev"], st["B"]
        with torch.cuda.device(dev):
            L.check(lib.ftc_plan_run(plan["handle"], st["bases"], st["stream"], 0, plan["n_fwd"] - 2), "ftc_plan_run (train step, forward without the loss op)")
            heat = self._view(plan["maps"], (B, plan["mh"], plan["mw"], 9)).permute(0, 3, 1, 2).clone()
            decs = [self._view(b, (plan["n_rows"], m)).clone() for b, m in zip(plan["dec_outs"], (1091, 1093, 1097))]
        from .optim import bump_versions
        bump_versions(self.stat_buffers)           # the forward ops moved the running statistics in place -- visible to the inference engine's
        st["phase"] = "forward"                     # fingerprint even if this forward is never backpropagated (_end() bumps again: harmless)
        self._seam = st
        return heat, decs

    def seam_losses(self, fmask: torch.Tensor, labelmap: torch.Tensor, idmap: torch.Tensor) -> torch.Tensor:
        st = getattr(self, "_seam", None)
        if st is None or st.get("phase") not in ("forward", "losses"):
            raise RuntimeError("loss_function on train-mode outputs needs the model(image, fmask) call of THIS step in front of it")
        lib = L.load()
        plan = st["plan"]
        with torch.cuda.device(st["dev"]):
            self._set_labels(st, labelmap, idmap)
            L.check(lib.ftc_plan_run(plan["handle"], st["bases"], st["stream"], plan["n_fwd"] - 1, plan["n_fwd"] - 1), "ftc_plan_run (train step, loss op)")
            out = self._view(plan["lossv"], (16,)).clone()
        st["phase"] = "losses"
        return out

    def seam_backward(self, weights9: torch.Tensor, sync_grads: bool = True) -> None:
        st = getattr(self, "_seam", None)
        if st is None or st.get("phase") != "losses":
            raise RuntimeError("backward of a train-mode loss: the step's activations are gone (one backward per model(image, fmask) call)")
        with torch.cuda.device(st["dev"]):
            self._backward(st, weights9, sync_grads and not getattr(self, "_no_sync", False))
            self._end(st)
        self._seam = None

    def no_sync(self):
        """Context manager for gradient accumulation under enable_ddp() with the reference's calling sequence (train1.py:125-140,
        `iters_to_accumulate` > 1): inside it `loss.backward()` adds this rank's gradients without the all-reduce; run the LAST micro-batch
        outside so the accumulated buffer is reduced once (what torch's DistributedDataParallel.no_sync() does)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prev = getattr(self, "_no_sync", False)
            self._no_sync = True
            try:
                yield self
            finally:
                self._no_sync = prev
        return cm()

    # ---- data-parallel gradients ----------------------------------------------------------------------------------------------
    def enable_ddp(self, group=None, bucket_bytes: int = 256 << 20) -> None:
        """One process per GPU, every rank a full replica: from now on forward_backward(sync_grads=True) all-reduces the flat gradient
        buffer over `group` in buckets, each bucket on a side stream as soon as the backward ops writing into it have been enqueued
        (findtextcenternet_amd.dist.BucketedAllReduce), and scales the loss by 1 / world so that the SUM is the average DDP gives."""
        from .dist import BucketedAllReduce
        self.ddp = BucketedAllReduce(self.grads, bucket_bytes, group)
        self.comm_stream = torch.cuda.Stream(device=self.dev) if self.grads.is_cuda else None
        self._segments: Dict[tuple, list] = {}

    @staticmethod
    def _grad_write_extents(op) -> list:
        """[(first element, one past the last)] of the flat gradient buffer an op writes: the FULL extent of every FTC_BASE_GRADS operand
        (a parameter may straddle a bucket boundary, and FTC_OP_SEBWD writes four consecutive parameters through one operand)."""
        k = op.kind
        sizes = {}
        if k == L.OP_WGRAD:
            sizes["out"] = op.Cout * op.Cin * op.ksize * op.ksize
        elif k == L.OP_BNBWD:
            sizes["w"] = sizes["shift"] = op.Cin
        elif k == L.OP_DWBWD:
            sizes["out2"] = 9 * op.Cin
        elif k == L.OP_SEBWD:
            sizes["out2"] = 2 * op.aux0 * op.Cin + op.aux0 + op.Cin       # fc1.weight [S][C], fc1.bias [S], fc2.weight [C][S], fc2.bias [C]
        elif k == L.OP_COLSUM:
            sizes["out"] = op.Cin
        elif k == L.OP_STEMWGRAD:
            sizes["out"] = op.Cout * 27
        ext = []
        for f in ("in_", "in2", "out", "out2", "w", "w2", "bias", "bias2", "scale", "shift", "aux"):
            r = getattr(op, f)
            if r.base != L.BASE_GRADS:
                continue
            if f not in sizes:
                raise RuntimeError(f"train plan: op kind {k} touches the gradient buffer through '{f}', whose extent _bucket_segments does not know")
            ext.append((r.offset // 4, r.offset // 4 + int(sizes[f])))
        return ext

    def _bucket_segments(self, plan: dict) -> list:
        """[(first_op, last_op, bucket index)]: the backward ops in order, cut after the last op that writes into each bucket.
        An op counts as a writer of EVERY bucket its written extent intersects (bucket boundaries are not parameter boundaries)."""
        ops, n_fwd, n = plan["ops"], plan["n_fwd"], plan["n_ops"]
        last_writer = [n_fwd - 1] * len(self.ddp.ranges)
        for i in range(n_fwd, n):
            for e0, e1 in self._grad_write_extents(ops[i]):
                for bi, (lo, hi) in enumerate(self.ddp.ranges):
                    if e0 < hi and lo < e1:
                        last_writer[bi] = max(last_writer[bi], i)
        # buckets complete in the order of `ranges` only if last_writer is monotone -- enforce it
        segs, start, done = [], n_fwd, n_fwd - 1
        for bi in range(len(self.ddp.ranges)):
            done = max(done, last_writer[bi])
            segs.append((start, done, bi))
            start = done + 1
        if start <= n - 1:
            segs.append((start, n - 1, None))
        return segs

    def maps(self, B: int, H: int, W: int, loss_scale: float = 1.0) -> torch.Tensor:
        """The [B,9,h,w] heat-map block of the last forward (a copy)."""
        plan = self.plan_for(B, H, W, loss_scale)
        return self._view(plan["maps"], (B, plan["mh"], plan["mw"], 9)).permute(0, 3, 1, 2).clone()
