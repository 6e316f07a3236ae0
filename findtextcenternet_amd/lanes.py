"""Successive batches on several HIP streams ("lanes").

One forward of the detector is two very different programs back to back: the backbone's MBConv stages are chains of small, latency-bound
kernels that leave most of the GPU idle (profiles/: 58-64 % of their wave cycles parked), the FPN heads are large matrix-bound kernels.
Nothing inside one batch can overlap them -- the heads need the backbone's taps.  ACROSS batches nothing stands in the way: batch k+1's
backbone runs under batch k's heads when the two are enqueued on different streams.  Each lane owns its activation arena, output
tensors and decode workspace (the weights and the plan are shared and immutable), so the lanes never synchronise with each other;
a lane's buffers are reused every `lanes` batches, ordered by the lane's own stream.  Measured on MI355X (tools/pipeline_experiment.py,
batch 8 x 768x768): 545 -> 623 images/s in bf16, 206 -> 232 in fp16x3, outputs bit-identical to the single-stream run.

This is a throughput device for callers that have a stream of tile batches (a page = several batches; a server = many pages): latency
of one batch is unchanged.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .decode import DecodeWorkspace, decode_peaks


class DetectorLanes:
    def __init__(self, detector, B: int, H: int = 768, W: int = 768, lanes: int = 2, max_boxes: int = 2048, device="cuda", decode=None):
        """detector: a findtextcenternet_amd.CenterNetDetector in eval mode on `device`.  Its weights must not change while batches are in
        flight (a parameter edit re-packs the weight blob every lane reads): call synchronize-then-edit, as with any stream of work.

        `decode` (default `decode_peaks`) and a CPU `device` exist for ONE purpose: the world-size-2 gloo test of the control flow
        (lane rotation, per-lane buffers, `then=` collectives issued lane after lane -- tests/test_dist.py) with a stub detector and a stub
        decode.  On a CPU device there are no streams: a submission runs inline, in order.  The product path is `device="cuda"`."""
        self.det, self.B, self.H, self.W, self.n = detector, B, H, W, lanes
        dev = torch.device(device)
        self.dev = dev
        self._decode = decode if decode is not None else decode_peaks
        h, w = H // 4, W // 4
        eng = detector.detector._engine
        import contextlib
        with (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):
            eng.ensure_model(dev)
            nbytes = eng.model.workspace_bytes(B, H, W)
            self.streams = [torch.cuda.Stream(device=dev) if dev.type == "cuda" else None for _ in range(lanes)]
            self.ws = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(lanes)]
            self.heat = [torch.empty((B, h, w, 10), dtype=torch.float32, device=dev) for _ in range(lanes)]
            self.feat = [torch.empty((B, h, w, 100), dtype=torch.float32, device=dev) for _ in range(lanes)]
            self.dws = [DecodeWorkspace(B, h, w, 100, max_boxes, dev) for _ in range(lanes)]
        self.max_boxes = max_boxes
        self.k = 0

    def submit(self, x: torch.Tensor, tiles, cut_off: float = 0.4, logit_cut: Optional[float] = None, then=None):
        """Enqueues forward + NMS + peak decode of batch x ([B,3,H,W], resident on the device, not modified) on the next lane and
        returns (lane index, Decoded of that lane) immediately; `then(decoded)` (e.g. the multi-GPU box gather) is called under the
        lane's stream.  The Decoded views are valid until the same lane is submitted to again (`lanes` submissions later).

        Lifetime of the inputs: `x` (and `tiles` when it is a device tensor) are READ ON THE LANE'S STREAM after this call returns.
        They are registered with the caching allocator (`Tensor.record_stream`), so the caller may drop or reassign them straight
        away -- their memory is not handed to a new allocation before the lane's work has passed; overwriting them IN PLACE on the
        caller's stream before `wait(lane)` is still a race, as with any asynchronous consumer."""
        i = self.k % self.n
        self.k += 1
        if self.dev.type != "cuda":                          # (the gloo test of the control flow: inline, in submission order)
            with torch.no_grad():
                self.det.forward_nhwc(x, out=(self.heat[i], self.feat[i]), workspace=self.ws[i])
                dec = self._decode(self.heat[i], self.feat[i], tiles, cut_off=cut_off, max_boxes=self.max_boxes, logit_cut=logit_cut, workspace=self.dws[i])
                return i, (then(dec) if then is not None else dec)
        cur = torch.cuda.current_stream(self.dev)
        s = self.streams[i]
        s.wait_stream(cur)                                   # whatever produced x (and the caller's earlier work) comes first
        for t in (x, tiles):                                 # allocated on the caller's stream, consumed on the lane's: tell the allocator
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(s)
        with torch.cuda.stream(s), torch.no_grad():
            self.det.forward_nhwc(x, out=(self.heat[i], self.feat[i]), workspace=self.ws[i])
            dec = self._decode(self.heat[i], self.feat[i], tiles, cut_off=cut_off, max_boxes=self.max_boxes, logit_cut=logit_cut, workspace=self.dws[i])
            out = then(dec) if then is not None else dec
        return i, out

    def wait(self, lane: Optional[int] = None) -> None:
        """Makes the CURRENT stream wait for one lane (or all): after it, that lane's outputs may be read on the current stream."""
        if self.dev.type != "cuda":
            return
        cur = torch.cuda.current_stream(self.dev)
        for j in ([lane] if lane is not None else range(self.n)):
            cur.wait_stream(self.streams[j])
