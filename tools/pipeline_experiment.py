#!/usr/bin/env python3
"""What-if: SUCCESSIVE batches on S HIP streams (batch k on stream k % S, own workspace and outputs per stream, no cross-stream
synchronisation): the latency-bound backbone stages of one batch overlap the compute-bound FPN heads of the previous one.
    python tools/pipeline_experiment.py [--precision bf16] [--batch 8] [--steps 40]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from findtextcenternet_amd import CenterNetDetector, TextDetectorModel, deterministic_state_dict  # noqa: E402
from findtextcenternet_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=40)
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
    ref_heat = torch.empty((B, 192, 192, 10), device="cuda")
    ref_feat = torch.empty((B, 192, 192, 100), device="cuda")
    wsb = eng.model.workspace_bytes(B, 768, 768)
    for S in (1, 2, 3, 1):
        streams = [torch.cuda.Stream() for _ in range(S)]
        ws = [torch.empty(wsb, dtype=torch.uint8, device="cuda") for _ in range(S)]
        heat = [torch.empty_like(ref_heat) for _ in range(S)]
        feat = [torch.empty_like(ref_feat) for _ in range(S)]
        torch.cuda.synchronize()

        def run(n):
            for k in range(n):
                i = k % S
                L.check(lib.ftc_forward(eng.handle, eng.wdev.data_ptr(), x.data_ptr(), B, 768, 768, 0, 1, heat[i].data_ptr(), feat[i].data_ptr(),
                                        ws[i].data_ptr(), C.c_void_p(streams[i].cuda_stream)), "ftc_forward")
        run(2 * S)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(a.steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if S == 1:
            ref_heat.copy_(heat[0])
        same = all(bool(torch.equal(h, ref_heat)) for h in heat)
        print(f"{S} stream(s): {1e3 * el / a.steps:7.3f} ms / batch  {B * a.steps / el:8.1f} img/s   outputs identical to the single-stream run: {same}", flush=True)


if __name__ == "__main__":
    main()
