#!/usr/bin/env python3
"""What-if: the batch-8 forward as S independent sub-batches on S HIP streams (same library calls, separate workspaces).

The MBConv stages are chains of small kernels (20-40 us, 1-5 workgroups per CU) whose launch ramps, tails and prologue / epilogue
latencies nothing overlaps on a single stream; two streams let the GPU overlap one sub-batch's ramps with the other's steady state.
    python tools/two_stream_experiment.py [--precision bf16] [--reps 20]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from findtextcenternet_amd import CenterNetDetector, TextDetectorModel, deterministic_state_dict  # noqa: E402
from findtextcenternet_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = L.load()
    model = TextDetectorModel(pre_weights=False, precision=a.precision)
    model.load_state_dict(deterministic_state_dict(0))
    det = CenterNetDetector(model.detector).to("cuda").eval()
    B = a.batch
    x = torch.rand((B, 768, 768, 3), device="cuda")
    with torch.no_grad():
        det(x.permute(0, 3, 1, 2))
    eng = model.detector._engine
    heat = torch.empty((B, 192, 192, 10), device="cuda")
    feat = torch.empty((B, 192, 192, 100), device="cuda")
    main_s = torch.cuda.current_stream()

    def run(parts):
        """parts: list of (start, count); part i runs on stream i (stream 0 = the current stream)."""
        streams = [main_s] + [torch.cuda.Stream() for _ in parts[1:]]
        wss = [torch.empty(eng.model.workspace_bytes(n, 768, 768), dtype=torch.uint8, device="cuda") for _, n in parts]

        def once():
            for s in streams[1:]:
                s.wait_stream(main_s)
            for (st, n), s, ws in zip(parts, streams, wss):
                L.check(lib.ftc_forward(eng.handle, eng.wdev.data_ptr(), x[st:st + n].data_ptr(), n, 768, 768, 0, 1, heat[st:st + n].data_ptr(),
                                        feat[st:st + n].data_ptr(), ws.data_ptr(), C.c_void_p(s.cuda_stream)), "ftc_forward")
            for s in streams[1:]:
                main_s.wait_stream(s)
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            once()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    ref = None
    for name, parts in (("1 stream x %d" % B, [(0, B)]), ("2 streams x %d" % (B // 2), [(0, B // 2), (B // 2, B - B // 2)]),
                        ("4 streams x %d" % (B // 4), [(i * (B // 4), B // 4) for i in range(4)]), ("1 stream x %d (again)" % B, [(0, B)])):
        ms = run(parts)
        if ref is None:
            ref = heat.clone()
        same = bool(torch.equal(ref, heat))
        print(f"{name:24s} {ms:8.3f} ms / step  {B / ms * 1e3:8.1f} img/s   heat identical to the single-stream run: {same}", flush=True)


if __name__ == "__main__":
    main()
