#!/usr/bin/env python3
"""What-if (round-3 verdict item 3): the two lanes on CU-MASKED HIP streams (hipExtStreamCreateWithCUMask) -- does giving the latency-bound
backbone chain of batch k+1 its own CUs next to batch k's 2 ms head kernel (which holds every CU's LDS) raise the two-lane rate?
    python tools/lanes_cumask_experiment.py [--steps 40]
Mask bit i = CU i of the device (256 bits); ROCr deals the bits round-robin over the 8 XCDs, so `i % 8 in S` selects whole XCDs and a
prefix of the bit range takes the same share of every XCD."""
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

hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipStreamDestroy.argtypes = [C.c_void_p]


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    lib = L.load()
    model = TextDetectorModel(pre_weights=False, precision="bf16")
    model.load_state_dict(deterministic_state_dict(0))
    det = CenterNetDetector(model.detector).to("cuda").eval()
    B = 8
    x = torch.rand((B, 768, 768, 3), device="cuda")
    with torch.no_grad():
        det(x.permute(0, 3, 1, 2))
    eng = model.detector._engine
    wsb = eng.model.workspace_bytes(B, 768, 768)
    ws = [torch.empty(wsb, dtype=torch.uint8, device="cuda") for _ in range(2)]
    heat = [torch.empty((B, 192, 192, 10), device="cuda") for _ in range(2)]
    feat = [torch.empty((B, 192, 192, 100), device="cuda") for _ in range(2)]
    allcu = set(range(256))
    cases = [
        ("1 stream, all CUs", [allcu]),
        ("2 lanes, all CUs each (what ships)", [allcu, allcu]),
        ("2 lanes: XCDs 0-3 | XCDs 4-7", [{i for i in allcu if i % 8 < 4}, {i for i in allcu if i % 8 >= 4}]),
        ("2 lanes: all | XCDs 0-3", [allcu, {i for i in allcu if i % 8 < 4}]),
        ("2 lanes: 3/4 of every XCD | all", [set(range(192)), allcu]),
        ("2 lanes: 3/4 of every XCD | the other 1/4", [set(range(192)), set(range(192, 256))]),
        ("2 lanes: 7/8 of every XCD | all", [set(range(224)), allcu]),
        ("1 stream, 3/4 of every XCD", [set(range(192))]),
        ("1 stream, XCDs 0-3", [{i for i in allcu if i % 8 < 4}]),
    ]
    for name, masks in cases:
        streams = [masked_stream(m) for m in masks]
        S = len(streams)
        torch.cuda.synchronize()

        def run(n):
            for k in range(n):
                i = k % S
                L.check(lib.ftc_forward(eng.handle, eng.wdev.data_ptr(), x.data_ptr(), B, 768, 768, 0, 1, heat[i].data_ptr(), feat[i].data_ptr(),
                                        ws[i].data_ptr(), streams[i]), "ftc_forward")
        run(2 * S)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(a.steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"{name:48s}: {1e3 * el / a.steps:7.3f} ms / batch  {B * a.steps / el:8.1f} img/s", flush=True)
        for s in streams:
            hip.hipStreamDestroy(s)


if __name__ == "__main__":
    main()
