#!/usr/bin/env python3
"""Benchmark of the detector hot path on MI355X: 768x768 images/s of
(detector forward + 3x3 NMS + GPU peak decode/feature gather [+ RCCL box all-gather when N > 1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass over one batch of 8 synthetic 768x768x3 tiles per GPU (BASELINE.json configs[1],
"batch=8 synthetic 768x768x3, full EffNetV2-XL detector fwd on 1xMI355X bf16"); inputs are resident
in HBM before the timed region; weak scaling (per-GPU batch fixed).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK = {"bf16": 2500.0, "fp32": 157.3}     # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(sd, seconds_budget: float = 25.0):
    """The CPU oracle (restatement of the reference path, kind "port") timed on this host's cores:
    forward + NMS + host decode of single 768x768 tiles, batch 1 as every reference caller does."""
    import synth
    from oracle import decode_oracle, detector_oracle
    x = torch.from_numpy(synth.noise_images(1234, 1, 768, 768)).permute(0, 3, 1, 2)
    rect = decode_oracle.tile_keep_rect(0, 0, 768, 768, 0.6)
    detector_oracle.detector_forward(sd, x)                       # warm-up (thread pools, allocations)
    n, t0 = 0, time.perf_counter()
    while True:
        hm, ft = detector_oracle.detector_forward(sd, x)
        decode_oracle.decode_tile(hm.numpy(), ft.numpy(), 0, 0, 768, 768, 0.4, rect)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 12:
            break
    return {"value": round(n / el, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} x (1 tile 768x768, fp32 torch-CPU oracle forward+NMS + numpy decode), {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="tiles per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--max-boxes", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--dump-ops", default="", help="write per-op timings (JSON) to this path")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import synth
    from findtextcenternet_amd import (CenterNetDetector, TextDetectorModel, TileGeom, decode_peaks, deterministic_state_dict,
                                       exact_logit_cut, tile_keep_rect, tiles_to_device)
    from findtextcenternet_amd import _lib as L
    from findtextcenternet_amd.dist import all_gather_boxes

    sd = deterministic_state_dict(0)
    model = TextDetectorModel(pre_weights=False, precision=args.precision)
    model.load_state_dict(sd)
    det = CenterNetDetector(model.detector)
    det.to(device=dev)
    det.eval()

    B = args.batch
    x = torch.from_numpy(synth.noise_images(1234 + rank, B, 768, 768)).to(dev).permute(0, 3, 1, 2)   # resident in HBM
    rect = tile_keep_rect(0, 0, 768, 768, 0.6)
    tiles = tiles_to_device([TileGeom(0, 0, 768, 768, rect) for _ in range(B)], dev, 192, 192)
    lcut = exact_logit_cut(0.4)

    def step():
        with torch.no_grad():
            heat, feat = det.forward_nhwc(x)
        dec = decode_peaks(heat, feat, tiles, cut_off=0.4, max_boxes=args.max_boxes, logit_cut=lcut)
        if world > 1:
            return all_gather_boxes(dec.counts, dec.boxes, dec.feats)
        return dec

    for _ in range(args.warmup):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    el_enqueue = 0.0
    for i in range(args.steps):
        out = step()
        if i == 0:
            el_enqueue = time.perf_counter() - t0  # host side of ONE step (later steps can block on a full hardware queue)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    peaks = float(out.counts.float().mean().item())

    result = None
    if rank == 0:
        value = world * B * args.steps / el
        result = {
            "metric": "768x768 images/s (detector fwd+NMS)", "value": round(value, 2), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * el / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: batch={B}/GPU synthetic 768x768x3 uniform-noise tiles, full "
                                   f"EfficientNetV2-XL detector forward + NMS + GPU peak decode/100-d gather"
                                   + (" + RCCL all-gather of boxes" if world > 1 else ""),
                       "global_batch": world * B, "tile": "768x768x3", "weights": "deterministic seed 0 (random-init)",
                       "parallelism": f"dp{world}", "mean_peaks_per_tile": round(peaks, 1)},
        }
        result["host_enqueue_ms_first_step"] = round(1000 * el_enqueue, 3)
        gflop = 865.0006
        result["path_tflops_per_gpu"] = round(value / world * gflop / 1000, 2)
        result["path_frac_of_mfma_peak"] = round(value / world * gflop / 1000 / PEAK[args.precision], 4)

    # ---- per-kernel attribution with HIP events on the launch stream (rank 0) -------------------
    if rank == 0 and not args.no_profile:
        lib = L.load()
        eng = model.detector._engine
        pl = eng.get_plan(B, 768, 768, False)
        heat = torch.empty((B, pl.h, pl.w, 10), dtype=torch.float32, device=dev)
        feat = torch.empty((B, pl.h, pl.w, 100), dtype=torch.float32, device=dev)
        bases = (C.c_void_p * L.NUM_BASES)(None, eng.workspace.data_ptr(), eng.wdev.data_ptr(), x.data_ptr(), heat.data_ptr(), feat.data_ptr())
        n_ops = len(pl.ops)
        ms = (C.c_float * n_ops)()
        acc = np.zeros(n_ops)
        reps = max(3, min(args.steps, 10))
        stream = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(reps):
            L.check(lib.ftc_plan_profile(pl.handle, bases, C.c_void_p(stream), ms), "ftc_plan_profile")
            acc += np.frombuffer(ms, dtype=np.float32)
        acc /= reps
        by = {}
        buf = C.create_string_buffer(128)
        for i in range(n_ops):
            lib.ftc_op_kernel_label(C.byref(pl.ops[i]), buf, 128)
            k = buf.value.decode()
            d = by.setdefault(k, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
            d["ms"] += float(acc[i]); d["flops"] += pl.meta[i].flops; d["bytes"] += pl.meta[i].bytes; d["launches"] += 1
        total_ms = float(acc.sum())
        dom = max(by.items(), key=lambda kv: kv[1]["ms"])
        name, d = dom
        is_conv = name.startswith("conv")
        if is_conv:
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK["bf16" if "<bf16" in name else "fp32"], "unit": "TFLOP/s"}
        else:
            ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s"}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        roof["traffic"] = None
        # HBM-side traffic of this kernel from the committed PMC passes (profiles/*_pmc_traffic.json: rocprofv3
        # FETCH_SIZE / WRITE_SIZE collected separately, gfx950 correction applied); null if not profiled.
        try:
            import glob
            for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
                tab = json.load(open(pth)).get("by_label", {})
                for lab, rec in tab.items():
                    if lab.split("|")[0] == name and args.precision == "bf16" and B == 8:
                        roof["traffic"] = rec["traffic_bytes"]
                        roof["traffic_note"] = f"bytes/launch, {os.path.basename(pth)}"
        except Exception:
            pass
        roof["kernel"] = name
        roof["launches_per_step"] = d["launches"]
        roof["avg_launch_ms"] = round(d["ms"] / d["launches"], 4)
        roof["algorithmic_gflop_per_launch"] = round(d["flops"] / d["launches"] / 1e9, 3)
        roof["share_of_forward_time"] = round(d["ms"] / total_ms, 3)
        result["roofline"] = roof
        conv_ms = sum(v["ms"] for k, v in by.items() if k.startswith("conv"))
        conv_fl = sum(v["flops"] for k, v in by.items() if k.startswith("conv"))
        result["all_conv_kernels"] = {"ms_per_step": round(conv_ms, 3), "tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                                      "share_of_forward_time": round(conv_ms / total_ms, 3)}
        result["forward_ms_sum_of_kernels"] = round(total_ms, 3)
        if args.dump_ops:
            ops = [{"i": i, "name": pl.meta[i].name, "kind": pl.meta[i].kind, "ms": float(acc[i]), "gflop": pl.meta[i].flops / 1e9,
                    "mbytes": pl.meta[i].bytes / 1e6} for i in range(n_ops)]
            with open(args.dump_ops, "w") as f:
                json.dump({"by_kernel": by, "ops": ops}, f, indent=1)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline({k: v for k, v in sd.items()})
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
