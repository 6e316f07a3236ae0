#!/usr/bin/env python3
"""Marginal cost of each op class UNDER two lanes: the forward plan is rebuilt without the ops of one class (the results are then garbage;
only the clock is read) and run on 1 and 2 streams.  What a class costs once the other lane fills its idle units is the difference to
the full plan on 2 streams -- the number that says where the next millisecond is.
    python tools/lanes_ablation.py [--precision bf16] [--batch 8] [--steps 40]
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
    pl = eng.plan(B, 768, 768, False)
    n = len(pl.ops)
    wsb = eng.model.workspace_bytes(B, 768, 768)
    S = 2
    streams = [torch.cuda.Stream() for _ in range(S)]
    ws = [torch.empty(wsb, dtype=torch.uint8, device="cuda") for _ in range(S)]
    heat = [torch.empty((B, 192, 192, 10), device="cuda") for _ in range(S)]
    feat = [torch.empty((B, 192, 192, 100), device="cuda") for _ in range(S)]
    bases = [(C.c_void_p * L.NUM_BASES)(None, ws[i].data_ptr(), eng.wdev.data_ptr(), x.data_ptr(), heat[i].data_ptr(), feat[i].data_ptr()) for i in range(S)]

    def stage(m):
        p = m.name.split(".")
        return ".".join(p[:3]) if p[0] == "backbone" else p[0]

    def timed(keep, label):
        idx = [i for i in range(n) if keep(pl.meta[i])]
        ops = (L.Op * len(idx))()
        for j, i in enumerate(idx):
            ops[j] = pl.ops[i]
        hp = C.c_void_p()
        L.check(lib.ftc_plan_create(ops, len(idx), wsb, eng.wdev.numel() * eng.wdev.element_size(), C.byref(hp)), "ftc_plan_create")
        res = []
        for ns in (1, 2):
            def run(k):
                for q in range(k):
                    i = q % ns
                    L.check(lib.ftc_plan_run(hp, bases[i], C.c_void_p(streams[i].cuda_stream), 0, -1), "ftc_plan_run")
            run(4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(a.steps)
            torch.cuda.synchronize()
            res.append(1e3 * (time.perf_counter() - t0) / a.steps)
        lib.ftc_plan_destroy(hp)
        print(f"{label:44s} ops {len(idx):4d}   1 stream {res[0]:7.3f} ms   2 lanes {res[1]:7.3f} ms", flush=True)
        return res

    full = timed(lambda m: True, "full plan")
    kinds = sorted({m.kind for m in pl.meta})
    for k in kinds:
        r = timed(lambda m, k=k: m.kind != k, f"without kind {k}")
        print(f"    -> {k}: costs {full[0] - r[0]:6.3f} ms on 1 stream, {full[1] - r[1]:6.3f} ms under 2 lanes")
    stages = []
    for m in pl.meta:
        if stage(m) not in stages:
            stages.append(stage(m))
    for s in stages:
        r = timed(lambda m, s=s: stage(m) != s, f"without {s}")
        print(f"    -> {s}: costs {full[0] - r[0]:6.3f} ms on 1 stream, {full[1] - r[1]:6.3f} ms under 2 lanes")
    for s in ("backbone.features.4", "backbone.features.5", "backbone.features.6", "backbone.features.7"):
        for k in ("conv1x1", "dwconv3x3", "se"):
            r = timed(lambda m, s=s, k=k: not (stage(m) == s and m.kind == k), f"without {s} {k}")
            print(f"    -> {s} {k}: costs {full[0] - r[0]:6.3f} ms on 1 stream, {full[1] - r[1]:6.3f} ms under 2 lanes")
    timed(lambda m: True, "full plan (again)")


if __name__ == "__main__":
    main()
