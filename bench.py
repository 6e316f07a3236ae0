#!/usr/bin/env python3
"""Benchmark of the detector hot path on MI355X: 768x768 images/s of
(detector forward + 3x3 NMS + GPU peak decode/feature gather [+ RCCL box all-gather when N > 1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass over one batch of 8 synthetic 768x768x3 tiles per GPU (BASELINE.json configs[1],
"batch=8 synthetic 768x768x3, full EffNetV2-XL detector fwd on 1xMI355X bf16"); inputs are resident
in HBM before the timed region and every output buffer is allocated before it; weak scaling (per-GPU batch fixed).
Rank 0 prints ONE JSON line.  `value` = images of all ranks / wall time of the K timed steps (barrier + synchronize on both
sides, max over ranks); `ms_per_step_median` is the median of the per-step HIP-event times on the launch stream.

Besides the headline (bf16 speed mode) the same line carries
  parity      -- L-inf / peak-set agreement of image 0 of the timed batch against the CPU oracle, for the timed bf16 model AND for
                 the fp32 parity mode and the fp16 mode (whose own images/s are reported there: they are different programs);
  roofline    -- the kernel with the largest share of the forward, HIP-event timed inside this run;
  cpu_baseline-- the CPU oracle ("port") on this host: thread sweep, batch 1 and 8, forward+NMS and +decode.
`--dry-run` exercises the whole N-rank control flow (sharding, decode records, gather, JSON) on CPU tensors under gloo
without touching the HIP library: it is what tests/test_bench_dryrun.py runs with world size 2.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md; fp16x3 = three fp16 MFMAs per product: a third of the 16-bit peak
PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3, "fp16x3": 833.3}
GFLOP_PER_IMAGE = 865.0006                 # SURVEY.md 8(d); reproduced by the library's own op list (tests/test_abi_and_plan.py)


def source_hash() -> str:
    """Hash of everything that decides which kernels run (kernel sources + tuning table): a PMC traffic profile is only
    quoted when it was taken from the same sources."""
    h = hashlib.sha1()
    base = os.path.join(ROOT, "findtextcenternet_amd")
    files = sorted(os.listdir(os.path.join(base, "csrc")))
    for f in files:
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(base, "csrc", f), "rb").read())
    h.update(open(os.path.join(base, "tuning_gfx950.json"), "rb").read())
    return h.hexdigest()[:12]


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle = pure-torch/numpy restatement of the reference path, kind "port"
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(sd, budget_s: float):
    """fp32, no_grad, thread count chosen by a sweep, batch 1 and batch 8, forward+NMS and
    forward+NMS+host decode separately, same seeded inputs as the GPU run.  Returns (record, oracle maps of image 0)."""
    from findtextcenternet_amd import synth
    from oracle import decode_oracle, detector_oracle
    ncpu = os.cpu_count() or 1
    x8 = torch.from_numpy(synth.noise_images(1234, 8, 768, 768)).permute(0, 3, 1, 2)
    x1 = x8[:1]
    rect = decode_oracle.tile_keep_rect(0, 0, 768, 768, 0.6)
    nthr0 = torch.get_num_threads()
    t_start = time.perf_counter()

    def fwd(x):
        t0 = time.perf_counter()
        hm, ft = detector_oracle.detector_forward(sd, x)
        return time.perf_counter() - t0, hm, ft

    def dec(hm, ft):
        t0 = time.perf_counter()
        for b in range(hm.shape[0]):
            decode_oracle.decode_tile(hm[b:b + 1].numpy(), ft[b:b + 1].numpy(), 0, 0, 768, 768, 0.4, rect)
        return time.perf_counter() - t0

    fwd(x1)                                                    # warm-up (thread pools, allocations, oneDNN primitives)
    # ONE measurement decides the thread count and is the reported batch-1 rate: per candidate, one warm-up at that thread count, then
    # the median of 3 (round 2 took min-of-2 in the sweep and a separate median-of-5 afterwards, and the two disagreed by 1.8x)
    sweep, runs = {}, {}
    for n in [t for t in (8, 16, 32, 64, 128, 256) if t <= ncpu] or [ncpu]:
        torch.set_num_threads(n)
        fwd(x1)
        runs[n] = [fwd(x1)[0] for _ in range(3)]
        sweep[n] = statistics.median(runs[n])
        # past the knee more threads only get slower (256 threads: 200 s per image on the 256-thread host): stop climbing there
        if time.perf_counter() - t_start > 0.6 * budget_s or sweep[n] > 1.3 * min(sweep.values()):
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t1 = runs[best]
    _, hm0, ft0 = fwd(x1)
    d1 = [dec(hm0, ft0) for _ in range(3)]
    t8, hm8, ft8 = fwd(x8)
    d8 = dec(hm8, ft8)
    torch.set_num_threads(nthr0)
    m1, md1 = statistics.median(t1), statistics.median(d1)
    rec = {"value": round(1.0 / (m1 + md1), 4), "unit": "images/s", "cores": best, "kind": "port",
           "sample": (f"CPU oracle (torch fp32 forward+NMS, numpy decode), 768x768 noise tiles, {best} of {ncpu} host threads chosen by "
                      f"sweep (per thread count: 1 warm-up, median of 3 -- the winner's median IS the batch-1 figure); batch 8: 1 pass; "
                      f"{time.perf_counter() - t_start:.0f} s total"),
           "threads_sweep_images_per_s_b1_fwd_nms": {str(k): round(1.0 / v, 4) for k, v in sweep.items()},
           "b1_fwd_nms_images_per_s": round(1.0 / m1, 4), "b1_fwd_nms_decode_images_per_s": round(1.0 / (m1 + md1), 4),
           "b8_fwd_nms_images_per_s": round(8.0 / t8, 4), "b8_fwd_nms_decode_images_per_s": round(8.0 / (t8 + d8), 4)}
    return rec, hm0.numpy(), ft0.numpy()


def parity_record(heat_nhwc, feat_nhwc, o_hm, o_ft):
    """Image 0 of the GPU batch against the oracle's maps of the same image."""
    from oracle import decode_oracle
    hm = heat_nhwc[:1].permute(0, 3, 1, 2).cpu().numpy()
    ft = feat_nhwc[:1].permute(0, 3, 1, 2).cpu().numpy()
    fin = np.isfinite(o_hm)
    both = fin & np.isfinite(hm)
    rect = decode_oracle.tile_keep_rect(0, 0, 768, 768, 0.6)
    z = np.zeros((1, 100, hm.shape[2], hm.shape[3]), np.float32)
    _, _, ir = decode_oracle.decode_tile(o_hm, z, 0, 0, 768, 768, 0.4, rect)
    _, _, ig = decode_oracle.decode_tile(hm, z, 0, 0, 768, 768, 0.4, rect)
    sr, sg = set(int(i) for i in ir), set(int(i) for i in ig)
    return {"heatmap_linf": float(f"{np.abs(hm[both] - o_hm[both]).max():.3e}"), "features_linf": float(f"{np.abs(ft - o_ft).max():.3e}"),
            "heatmap_range": round(float(o_hm[fin].max() - o_hm[fin].min()), 2),
            "nms_mask_flips": int((np.isfinite(hm[:, 1]) != fin[:, 1]).sum()), "peaks_ref": len(sr), "peaks_gpu": len(sg),
            "peaks_common": len(sr & sg), "peak_jaccard": round(len(sr & sg) / max(1, len(sr | sg)), 4), "peak_set_identical": sr == sg}


# ------------------------------------------------------------------------------------------------------------------
def dry_run(args, rank, world):
    """The N-rank control flow on CPU tensors (gloo): shard, build decode-shaped records, gather, reduce the time, print the
    JSON line.  No HIP library, no model: this only proves that the first real multi-GPU launch is not the first execution
    of this code path."""
    from findtextcenternet_amd.decode import REC_W
    from findtextcenternet_amd.dist import all_gather_boxes, all_gather_boxes_static, shard_range
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B = args.batch
    lo, hi = shard_range(world * B, rank, world)
    g = torch.Generator().manual_seed(77 + rank)
    counts = torch.randint(50, 400, (hi - lo,), generator=g, dtype=torch.int32)
    rec = torch.randn(hi - lo, args.max_boxes, REC_W, generator=g)
    out = None
    for _ in range(args.warmup):
        out = all_gather_boxes(counts, rec)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = all_gather_boxes(counts, rec)
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ok = out.counts.shape[0] == world * B and torch.equal(out.counts[lo:hi], counts)
    # the steady-state form the real bench uses when the capacity block is small (one collective, no host synchronisation)
    st = all_gather_boxes_static(counts, rec.clone(), world * B)
    ok_static = st.counts.shape[0] == world * B and torch.equal(st.counts[lo:hi], counts) and not bool(st.overflow)
    if rank == 0:
        print(json.dumps({"metric": "768x768 images/s (detector fwd+NMS)", "value": round(world * B * args.steps / el, 2), "unit": "images/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * el / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "dry_run": True, "gather_ok": bool(ok), "gather_message_bytes_per_rank": out.message_bytes_per_rank,
                          "static_gather_ok": bool(ok_static), "static_gather_message_bytes_per_rank": st.message_bytes_per_rank,
                          "config": {"workload": "DRY RUN (CPU tensors, gloo): control flow of the N-rank bench only, no detector work",
                                     "global_batch": world * B, "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()

# ------------------------------------------------------------------------------------------------------------------
# --train: BASELINE configs[4], "train1.py step: detector fwd+bwd with loss_func.py focal/L1 heads, batch=16, 2xMI355X DDP" = 8 tiles
# per GPU.  A step = zero_grad + get_fmask + forward (train mode) + loss_function + CoV weighting + backward (+ bucketed RCCL
# all-reduce of the 1.05 GB of gradients when N > 1) + the Schedule-Free AdamW update -- the body of the reference's loop,
# /root/reference/train1.py:170-179.
# ------------------------------------------------------------------------------------------------------------------
TRAIN_METRIC = "768x768 images/s (train step: fwd+loss+bwd+optimizer)"


def train_dry_run(args, rank, world):
    """The N-rank control flow of the train bench on CPU tensors (gloo): bucketed all-reduce of a flat gradient buffer, the
    max-over-ranks timing and the JSON line -- no HIP library."""
    from findtextcenternet_amd.dist import BucketedAllReduce
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 3_000_003
    g = torch.Generator().manual_seed(5 + rank)
    flat = torch.randn(n, generator=g)
    want = sum(torch.randn(n, generator=torch.Generator().manual_seed(5 + r)) for r in range(world))
    red = BucketedAllReduce(flat, bucket_bytes=4 << 20)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(len(red.ranges)):
        red.reduce_bucket(i, async_op=True)
    red.wait()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ok = bool(torch.allclose(flat, want, rtol=1e-6, atol=1e-6))
    if rank == 0:
        print(json.dumps({"metric": TRAIN_METRIC, "value": round(world * args.batch / el, 2), "unit": "images/s", "n_gpus": world, "steps": 1,
                          "warmup": 0, "ms_per_step": round(1000 * el, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "none", "data": "synthetic", "dry_run": True, "allreduce_ok": ok, "buckets": len(red.ranges),
                          "config": {"workload": "DRY RUN (CPU tensors, gloo): gradient all-reduce control flow of the train bench only",
                                     "global_batch": world * args.batch, "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def emit_last_line(result) -> None:
    """The ONE JSON line, as the last thing on stdout: RCCL writes its version banner through C stdio, which (piped) is flushed at exit,
    i.e. AFTER anything Python printed earlier -- flush the C buffers first, print after the process group is gone."""
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    # the contract-grade figure and the parity record go LAST: whoever keeps only the tail of the line keeps those
    tail_keys = [k for k in ("config3_b32", "parity", "north_star_value") if k in result]
    result = {**{k: v for k, v in result.items() if k not in tail_keys}, **{k: result[k] for k in tail_keys}}
    print(json.dumps(result), flush=True)


def train_bench(args, rank, local_rank, world, emit=True):
    """emit=False: the short `train_step` record of the default (inference) line -- one rank, no process group, returns the dict."""
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    # FTC_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, box gather under the lane streams, barriers) with one rank --
    # the only way to exercise it on a 1-GPU box
    dist_on = emit and (world > 1 or os.environ.get("FTC_BENCH_FORCE_DIST") == "1")
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from findtextcenternet_amd import AdamWScheduleFree, TextDetectorModel, TrainStep, deterministic_state_dict, synth
    from findtextcenternet_amd import _lib as L
    B, S = args.batch, 768
    sd = deterministic_state_dict(0)
    model = TextDetectorModel(pre_weights=False, precision=args.precision)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    ts = TrainStep(model)
    if dist_on:
        ts.enable_ddp()
    opt = AdamWScheduleFree([p for p in model.parameters() if p.requires_grad], lr=1e-4)     # train1.py:103-104
    opt.train()
    x = torch.from_numpy(synth.noise_images(1234 + rank, B, S, S)).to(dev).permute(0, 3, 1, 2)
    lab, idm = synth.train_labels(99 + rank, B, S // 4, S // 4)
    lab, idm = torch.from_numpy(lab).to(dev), torch.from_numpy(idm).to(dev)
    state = {"fmask": None, "loss": None}

    def step():
        ts.zero_grad()
        state["fmask"] = model.get_fmask(lab, state["fmask"])
        state["loss"], _ = ts.forward_backward(x, lab, idm, state["fmask"])
        opt.step()

    for _ in range(args.warmup):
        step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    if dist_on:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    loss = float(state["loss"])
    if rank != 0:
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return
    value = world * B * args.steps / el
    plan = ts.plan_for(B, S, S, 1.0 / world)
    result = {"metric": TRAIN_METRIC, "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": round(1000 * el / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": args.precision, "data": "synthetic",
              "config": {"workload": f"BASELINE configs[4]: train1.py step (train-mode detector + SimpleDecoder forward, loss_function, CoV weighting, "
                                     f"backward, Schedule-Free AdamW), batch={B}/GPU synthetic 768x768x3 tiles + synthetic label maps"
                                     + (", bucketed RCCL gradient all-reduce" if world > 1 else ""),
                         "global_batch": world * B, "tile": "768x768x3", "weights": "deterministic seed 0 (random-init)", "parallelism": f"dp{world}",
                         "fp32 master parameters, MFMA operands": args.precision},
              "ms_per_step_median": round(statistics.median(step_ms), 3), "loss_last_step": round(loss, 5), "finite": bool(np.isfinite(loss)),
              "plan_ops": plan["n_ops"], "forward_ops": plan["n_fwd"], "workspace_gb": round(plan["workspace_bytes"] / 2**30, 2),
              "gradient_bytes": int(ts.grads.numel() * 4), "source_hash": source_hash(),
              "train_gflop_per_image_dense": round(3 * GFLOP_PER_IMAGE, 1),
              "path_tflops_per_gpu": round(value / world * 3 * GFLOP_PER_IMAGE / 1000, 2),
              "path_frac_of_mfma_peak": round(value / world * 3 * GFLOP_PER_IMAGE / 1000 / PEAK[args.precision], 4)}
    if not args.no_profile:
        lib = L.load()
        n_ops = plan["n_ops"]
        ms = (C.c_float * n_ops)()
        xn = x.permute(0, 2, 3, 1).contiguous()
        bases = (C.c_void_p * L.NUM_BASES)(None, ts.workspace.data_ptr(), ts.blob.data_ptr(), xn.data_ptr(), None, None, ts.grads.data_ptr())
        acc = np.zeros((3, n_ops))
        for r in range(3):
            L.check(lib.ftc_plan_profile(plan["handle"], bases, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), ms), "ftc_plan_profile")
            acc[r] = np.frombuffer(ms, dtype=np.float32)
        acc = np.median(acc, axis=0)
        by = {}
        buf = C.create_string_buffer(256)
        for i in range(n_ops):
            o = plan["ops"][i]
            lib.ftc_op_kernel_label(C.byref(o), buf, 256)
            k = buf.value.decode()
            fl = by_ = 0.0
            if o.kind == L.OP_CONV:
                k = ("dgrad:" if plan["names"][i].startswith("dgrad:") else "fwd:") + k.split("<")[0]
                fl = 2.0 * o.B * o.Ho * o.Wo * o.Cout * o.Cin * o.ksize * o.ksize
            elif o.kind == L.OP_WGRAD:
                fl = 2.0 * o.B * o.Ho * o.Wo * o.Cout * o.Cin * o.ksize * o.ksize
            elif o.kind in (L.OP_BNBWD, L.OP_BNSTAT, L.OP_BNACT):
                by_ = float(o.B) * o.H * o.W * o.Cin * 4 * {L.OP_BNBWD: 5, L.OP_BNSTAT: 1, L.OP_BNACT: 2}[o.kind]
            d = by.setdefault(k, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
            d["ms"] += float(acc[i]); d["flops"] += fl; d["bytes"] += by_; d["launches"] += 1
        total = float(acc.sum())
        name, d = max(by.items(), key=lambda kv: kv[1]["ms"])
        if d["flops"] > 0:
            roof = {"bound": "mfma", "achieved": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2), "peak": PEAK[args.precision], "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s"}
        roof.update({"frac": round(roof["achieved"] / roof["peak"], 4), "traffic": None, "kernel": name, "launches_per_step": d["launches"],
                     "share_of_step_time": round(d["ms"] / total, 3),
                     "algorithmic_note": "MFMA kernels: 2 * pixels * Cout * Cin * k^2 per launch; BatchNorm passes: streams of 4-byte elements "
                                         "(backward: dy and z twice + dz = 5, statistics 1, normalise 2)"})
        result["roofline"] = roof
        result["kernel_time_by_label_ms"] = {k: round(v["ms"], 2) for k, v in sorted(by.items(), key=lambda kv: -kv[1]["ms"])[:12]}
        result["forward_ms_sum_of_kernels"] = round(float(acc[:plan["n_fwd"]].sum()), 2)
        result["backward_ms_sum_of_kernels"] = round(float(acc[plan["n_fwd"]:].sum()), 2)
    if not emit:
        del ts, opt, model
        torch.cuda.empty_cache()
        return result
    if world == 1 and not args.no_cpu_baseline:
        # the CPU oracle of the same step (torch autograd, fp32) on a bounded sample: batch 1, 384x384 (a quarter of a tile), once
        from oracle import train_oracle
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        xs = torch.from_numpy(synth.noise_images(7, 1, 384, 384)).permute(0, 3, 1, 2)
        l2, i2 = synth.train_labels(8, 1, 96, 96)
        t0 = time.perf_counter()
        train_oracle.train_step(sd, xs, torch.from_numpy(l2), torch.from_numpy(i2).long())
        e = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": round(0.25 / e, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                                  "sample": f"one train step (forward + loss + backward, no optimizer) of the CPU oracle on 1 x 384x384 "
                                            f"(= 1/4 of a 768x768 tile) in {e:.1f} s, scaled by pixel count"}
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    emit_last_line(result)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="tiles per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp16", "fp16x3"])
    ap.add_argument("--max-boxes", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (and with it the parity records)")
    ap.add_argument("--cpu-budget", type=float, default=60.0, help="seconds of CPU-oracle time to aim for")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 parity-mode record")
    ap.add_argument("--dump-ops", default="", help="write per-op timings (JSON) to this path")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo walk through the N-rank control flow (no GPU work)")
    ap.add_argument("--lanes", type=int, default=2, help="successive batches alternate over this many HIP streams (1 = one stream)")
    ap.add_argument("--no-seam2", action="store_true", help="skip the batch-1 host-in / host-out call_detector latency record")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 300-step / >= 4 s sustained-rate record")
    ap.add_argument("--train", action="store_true", help="BASELINE configs[4]: the train step (fwd + loss + bwd + optimizer, DDP all-reduce when N > 1)")
    ap.add_argument("--no-configs", action="store_true", help="skip the config3_b32 (batch 32 + decode + gather) and train_step (5 steps) records of the default line")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU) exactly as the driver's documented
        # command does, and pass its exit code on.  The ranks inherit stdout, so rank 0's JSON line is the last line printed.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        raise SystemExit(subprocess.run(cmd).returncode)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch one process per GPU, or run plain `python bench.py --gpus N`)")
    if args.dry_run:
        return train_dry_run(args, rank, world) if args.train else dry_run(args, rank, world)
    if args.train:
        return train_bench(args, rank, local_rank, world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist_on = world > 1 or os.environ.get("FTC_BENCH_FORCE_DIST") == "1"      # (see train_bench)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from findtextcenternet_amd import (CenterNetDetector, TextDetectorModel, TileGeom, decode_peaks, deterministic_state_dict,
                                       exact_logit_cut, synth, tile_keep_rect, tiles_to_device)
    from findtextcenternet_amd import _lib as L
    from findtextcenternet_amd.decode import DecodeWorkspace
    from findtextcenternet_amd.dist import STATIC_GATHER_BYTES, all_gather_boxes, all_gather_boxes_static

    sd = deterministic_state_dict(0)

    def make(precision):
        model = TextDetectorModel(pre_weights=False, precision=precision)
        model.load_state_dict(sd)
        d = CenterNetDetector(model.detector)
        d.to(device=dev)
        d.eval()
        return model, d

    model, det = make(args.precision)
    B = args.batch
    x = torch.from_numpy(synth.noise_images(1234 + rank, B, 768, 768)).to(dev).permute(0, 3, 1, 2)   # resident in HBM
    rect = tile_keep_rect(0, 0, 768, 768, 0.6)
    tiles = tiles_to_device([TileGeom(0, 0, 768, 768, rect) for _ in range(B)], dev, 192, 192)
    lcut = exact_logit_cut(0.4)
    # every output of a step is allocated once, before the timed region
    heat = torch.empty((B, 192, 192, 10), dtype=torch.float32, device=dev)
    feat = torch.empty((B, 192, 192, 100), dtype=torch.float32, device=dev)
    dws = DecodeWorkspace(B, 192, 192, 100, args.max_boxes, dev)

    def step():
        with torch.no_grad():
            det.forward_nhwc(x, out=(heat, feat))
        dec = decode_peaks(heat, feat, tiles, cut_off=0.4, max_boxes=args.max_boxes, logit_cut=lcut, workspace=dws)
        if dist_on:
            # one collective and no host round trip when the fixed-capacity block is small (8 tiles x 2048 rows = 7.3 MB), else counts first
            if dec.records.numel() * 4 <= STATIC_GATHER_BYTES:
                return all_gather_boxes_static(dec.counts, dec.records, world * B)
            return all_gather_boxes(dec.counts, dec.records)
        return dec

    # The timed step: successive batches alternate over `--lanes` HIP streams (findtextcenternet_amd.lanes.DetectorLanes): batch k+1's
    # latency-bound backbone stages run under batch k's matrix-bound FPN heads.  Every lane has its own arena, outputs and decode block;
    # a step still is one full pass (forward + NMS + decode [+ gather]) over one batch, and all K steps complete inside the timed region.
    # The single-stream figure (the step above, on the current stream) is measured beside it.
    single_step = step
    lanes = None
    if args.lanes > 1:
        from findtextcenternet_amd import DetectorLanes
        lanes = DetectorLanes(det, B, 768, 768, lanes=args.lanes, max_boxes=args.max_boxes, device=dev)

        def gather(dec):
            if dist_on:
                if dec.records.numel() * 4 <= STATIC_GATHER_BYTES:
                    return all_gather_boxes_static(dec.counts, dec.records, world * B)
                return all_gather_boxes(dec.counts, dec.records)
            return dec

        def step():                                            # noqa: F811
            return lanes.submit(x, tiles, cut_off=0.4, logit_cut=lcut, then=gather)[1]

    def drain():
        if lanes is not None:
            lanes.wait()

    for _ in range(args.warmup):
        out = step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # the kernels run on torch's current stream
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    el_enqueue = 0.0
    for i in range(args.steps):
        out = step()
        ev[i + 1].record(lanes.streams[(lanes.k - 1) % lanes.n] if lanes is not None else None)
        if i == 0:
            el_enqueue = time.perf_counter() - t0  # host side of ONE step (later steps can block on a full hardware queue)
    drain()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # per-step HIP events: with lanes, event i+1 is recorded on the lane that ran step i, so consecutive events of ONE lane are `lanes` steps apart
    nl = lanes.n if lanes is not None else 1
    step_ms = [ev[i].elapsed_time(ev[i + nl]) / nl for i in range(1 if nl > 1 else 0, args.steps + 1 - nl)] or [1000 * el / args.steps]
    # the same K steps on ONE stream (no overlap between batches): reported beside the headline
    single = None
    if lanes is not None:
        for _ in range(2):
            single_step()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            single_step()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        if dist_on:
            t = torch.tensor([e1], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e1 = float(t.item())
        single = {"value": round(world * B * args.steps / e1, 2), "ms_per_step": round(1000 * e1 / args.steps, 3)}
    if dist_on:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    peaks = float(out.counts.float().mean().item())

    # ---- sustained rate: >= 300 steps and >= 4 s (the 20-step burst above lasts 0.3 s; the part's clock settles over seconds) ----
    sus_steps = max(300, int(4.0 / max(1e-4, el / args.steps)) + 1) if not args.no_sustained else 0
    sus = None
    if sus_steps:
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0s = time.perf_counter()
        for _ in range(sus_steps):
            out = step()
        drain()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        els = time.perf_counter() - t0s
        if dist_on:
            t = torch.tensor([els], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            els = float(t.item())
        sus = {"steps": sus_steps, "seconds": round(els, 3), "value": round(world * B * sus_steps / els, 2), "ms_per_step": round(1000 * els / sus_steps, 3)}

    result = None
    if rank == 0:
        value = world * B * args.steps / el
        result = {
            "metric": "768x768 images/s (detector fwd+NMS)", "value": round(value, 2), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * el / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: batch={B}/GPU synthetic 768x768x3 uniform-noise tiles, full "
                                   f"EfficientNetV2-XL detector forward + NMS + GPU peak decode/100-d gather"
                                   + (" + RCCL all-gather of boxes" if dist_on else ""),
                       "global_batch": world * B, "tile": "768x768x3", "weights": "deterministic seed 0 (random-init)",
                       "parallelism": f"dp{world}", "mean_peaks_per_tile": round(peaks, 1)},
            "ms_per_step_median": round(statistics.median(step_ms), 3),
            "ms_per_step_min_max": [round(min(step_ms), 3), round(max(step_ms), 3)],
            "timing": "value from wall time around the K steps (barrier + synchronize both sides, max over ranks); "
                      "ms_per_step_median from per-step HIP events on the launch stream (rank 0)",
            "source_hash": source_hash(),
        }
        result["lanes"] = nl
        if single is not None:
            result["single_stream"] = single
            result["config"]["pipelining"] = (f"successive batches alternate over {nl} HIP streams (own arena / outputs / decode block per lane): batch k+1's "
                                              "backbone overlaps batch k's FPN heads; single_stream = the same steps on one stream")
        if sus is not None:
            result["sustained"] = sus
        if dist_on:
            result["gather_message_bytes_per_rank"] = out.message_bytes_per_rank
        result["host_enqueue_ms_first_step"] = round(1000 * el_enqueue, 3)
        result["path_tflops_per_gpu"] = round(value / world * GFLOP_PER_IMAGE / 1000, 2)
        result["path_frac_of_mfma_peak"] = round(value / world * GFLOP_PER_IMAGE / 1000 / PEAK[args.precision], 4)

    # ---- per-kernel attribution with HIP events on the launch stream (rank 0) -------------------
    def kernel_profile(model_, x_, heat_, feat_, precision, reps, dump=""):
        """Per-op HIP-event times of one forward of `model_`'s plan (ftc_plan_profile), grouped by kernel-instantiation label.  Returns the
        roofline record of the label with the largest share and the summary fields."""
        lib = L.load()
        eng = model_.detector._engine
        Bp = x_.shape[0]
        pl = eng.plan(Bp, 768, 768, False)
        bases = (C.c_void_p * L.NUM_BASES)(None, eng.workspace.data_ptr(), eng.wdev.data_ptr(), x_.data_ptr(), heat_.data_ptr(), feat_.data_ptr())
        n_ops = len(pl.ops)
        ms = (C.c_float * n_ops)()
        acc = np.zeros((reps, n_ops))
        stream = torch.cuda.current_stream(dev).cuda_stream
        for r in range(reps):
            L.check(lib.ftc_plan_profile(pl.handle, bases, C.c_void_p(stream), ms), "ftc_plan_profile")
            acc[r] = np.frombuffer(ms, dtype=np.float32)
        acc = np.median(acc, axis=0)
        by = {}
        buf = C.create_string_buffer(128)
        for i in range(n_ops):
            lib.ftc_op_kernel_label(C.byref(pl.ops[i]), buf, 128)
            k = buf.value.decode()
            d = by.setdefault(k, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
            d["ms"] += float(acc[i]); d["flops"] += pl.meta[i].flops; d["bytes"] += pl.meta[i].bytes; d["launches"] += 1
        total_ms = float(acc.sum())
        # The roofline kernel = the kernel FAMILY with the largest share of the forward (round-4 verdict: the two instantiations of
        # mbconv_slice -- whole 24x24 maps / bands of 48x48 maps -- are one kernel and together outweigh the largest single launch):
        # labels are merged on the part before the first '<' plus the tile, except that mbconv_slice merges on its name alone.
        fam = {}
        for k, v in by.items():
            fk = "mbconv_slice" if k.startswith("mbconv_slice") else k
            f_ = fam.setdefault(fk, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "labels": []})
            for q in ("ms", "flops", "bytes", "launches"):
                f_[q] += v[q]
            f_["labels"].append(k)
        name, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
        if name.startswith(("conv", "mbconv", "fmbconv")):
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            pk = "fp32" if ("<f32" in name and "x3" not in name and precision == "fp32") else precision
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK[pk], "unit": "TFLOP/s"}
        else:
            ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s"}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        roof["traffic"] = None
        # HBM-side traffic of this kernel from the committed PMC passes (profiles/*_pmc_traffic.json: rocprofv3 FETCH_SIZE /
        # WRITE_SIZE collected in separate passes, gfx950 correction applied) -- quoted only when that profile was taken from
        # the same kernel sources + tuning table as this run (source_hash), else null.
        try:
            import glob
            for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
                prof = json.load(open(pth))
                if prof.get("source_hash") != source_hash() or prof.get("batch") != Bp or prof.get("precision") != precision:
                    continue
                tr, nl_ = 0.0, 0
                for lab, rec in prof.get("by_label", {}).items():
                    if lab.split("|")[0] in d["labels"]:
                        tr += rec["traffic_bytes"] * rec.get("launches", 1); nl_ += rec.get("launches", 1)
                if nl_:
                    roof["traffic"] = int(tr / nl_)
                    roof["traffic_note"] = f"bytes/launch, {os.path.basename(pth)} (source_hash {prof['source_hash']})"
        except Exception:
            pass
        roof["kernel"] = name if len(d["labels"]) == 1 else name + " (" + " + ".join(sorted(d["labels"])) + ")"
        roof["launches_per_step"] = d["launches"]
        roof["avg_launch_ms"] = round(d["ms"] / d["launches"], 4)
        roof["algorithmic_gflop_per_launch"] = round(d["flops"] / d["launches"] / 1e9, 3)
        roof["algorithmic_mbytes_per_launch"] = round(d["bytes"] / d["launches"] / 1e6, 2)
        roof["share_of_forward_time"] = round(d["ms"] / total_ms, 3)
        # ... and the largest SINGLE launch, as rounds 1-4 reported it
        n1, d1 = max(by.items(), key=lambda kv: kv[1]["ms"] / kv[1]["launches"])
        if n1.startswith(("conv", "mbconv", "fmbconv")):
            roof["largest_launch"] = {"kernel": n1, "avg_launch_ms": round(d1["ms"] / d1["launches"], 4),
                                      "achieved": round(d1["flops"] / (d1["ms"] * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                      "frac": round(d1["flops"] / (d1["ms"] * 1e-3) / 1e12 / PEAK[precision], 4)}
        conv_ms = sum(v["ms"] for k, v in by.items() if k.startswith(("conv", "mbconv", "fmbconv")))
        conv_fl = sum(v["flops"] for k, v in by.items() if k.startswith(("conv", "mbconv", "fmbconv")))
        allconv = {"ms_per_step": round(conv_ms, 3), "tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                   "share_of_forward_time": round(conv_ms / total_ms, 3)}        # (incl. the fused MBConv heads)
        if dump:
            ops = [{"i": i, "name": pl.meta[i].name, "kind": pl.meta[i].kind, "ms": float(acc[i]), "gflop": pl.meta[i].flops / 1e9,
                    "mbytes": pl.meta[i].bytes / 1e6} for i in range(n_ops)]
            with open(dump, "w") as f:
                json.dump({"by_kernel": by, "ops": ops}, f, indent=1)
        return roof, allconv, round(total_ms, 3)

    if rank == 0 and not args.no_profile:
        roof, allconv, total_ms = kernel_profile(model, x, heat, feat, args.precision, max(3, min(args.steps, 10)), args.dump_ops)
        result["roofline"] = roof
        result["all_conv_kernels"] = allconv
        result["forward_ms_sum_of_kernels"] = total_ms

    # ---- the other numeric mode, parity of both against the CPU oracle, CPU baseline (rank 0, N = 1) -------------
    if rank == 0 and world == 1:
        others = [] if args.no_fp32 else [p_ for p_ in ("fp32", "fp16x3", "fp16", "bf16") if p_ != args.precision]
        recs, maps = {}, {}
        for other in others:
            model2, det2 = make(other)
            heat2, feat2 = torch.empty_like(heat), torch.empty_like(feat)
            dws2 = DecodeWorkspace(B, 192, 192, 100, args.max_boxes, dev)

            def step2():
                with torch.no_grad():
                    det2.forward_nhwc(x, out=(heat2, feat2))
                return decode_peaks(heat2, feat2, tiles, cut_off=0.4, max_boxes=args.max_boxes, logit_cut=lcut, workspace=dws2)
            k2 = max(3, args.steps // 4) if other == "fp32" else max(3, args.steps // 2) if other == "fp16x3" else args.steps
            step2()
            step2()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k2):
                step2()
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t0
            recs[other] = {"dtype": other, "images_per_s": round(B * k2 / e2, 2), "ms_per_step": round(1000 * e2 / k2, 3), "steps": k2,
                           "path_frac_of_mfma_peak": round(B * k2 / e2 * GFLOP_PER_IMAGE / 1000 / PEAK[other], 4)}
            maps[other] = (heat2[:1].clone(), feat2[:1].clone())
            if other == "fp16x3" and not args.no_profile:      # the contract-grade mode's own dominant kernel (north_star_value below)
                try:
                    r3, c3, t3 = kernel_profile(model2, x, heat2, feat2, other, 3, args.dump_ops.replace(".json", "_fp16x3.json") if args.dump_ops else "")
                    recs[other].update({"roofline": r3, "all_conv_kernels": c3, "forward_ms_sum_of_kernels": t3})
                except Exception as ex:
                    recs[other]["roofline"] = {"error": repr(ex)[:200]}
            if args.lanes > 1:                              # the same steps alternating over the lanes, as the headline does
                from findtextcenternet_amd import DetectorLanes
                ln2 = DetectorLanes(det2, B, 768, 768, lanes=args.lanes, max_boxes=args.max_boxes, device=dev)
                for _ in range(2 * args.lanes):
                    ln2.submit(x, tiles, cut_off=0.4, logit_cut=lcut)
                ln2.wait()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(k2):
                    ln2.submit(x, tiles, cut_off=0.4, logit_cut=lcut)
                ln2.wait()
                torch.cuda.synchronize()
                e3 = time.perf_counter() - t0
                recs[other].update({"single_stream_images_per_s": recs[other]["images_per_s"], "single_stream_ms_per_step": recs[other]["ms_per_step"],
                                    "images_per_s": round(B * k2 / e3, 2), "ms_per_step": round(1000 * e3 / k2, 3), "lanes": args.lanes,
                                    "path_frac_of_mfma_peak": round(B * k2 / e3 * GFLOP_PER_IMAGE / 1000 / PEAK[other], 4)})
                del ln2
            del model2, det2, heat2, feat2, dws2
            torch.cuda.empty_cache()
        # ---- seam 2 exactly as the reference calls it (process_ocr_torch.py:43-49): call_detector(np [1,768,768,3] 0..255) ->
        # numpy heatmap + features, batch 1, host buffers in and out (H2D 7 MB, D2H 16 MB per tile) -- latency, not the headline
        from findtextcenternet_amd import HipDetectorBackend
        seam = {}
        tile_np = (synth.noise_images(4321, 1, 768, 768) * 255.0).astype(np.float32)
        for prec in ([] if args.no_seam2 else [args.precision] + ([] if args.no_fp32 else [p_ for p_ in ("fp32", "fp16x3") if p_ != args.precision])):
            mdl, dd = (model, det) if prec == args.precision else make(prec)
            be = HipDetectorBackend(dd, device=dev)
            for _ in range(3):
                be.call_detector(tile_np)
            ts_ = []
            for _ in range(20):
                t0c = time.perf_counter()
                be.call_detector(tile_np)
                ts_.append(time.perf_counter() - t0c)
            seam[prec] = {"ms_per_call_median": round(1000 * statistics.median(ts_), 3), "images_per_s": round(1.0 / statistics.median(ts_), 2)}
            if prec != args.precision:
                del mdl, dd, be
                torch.cuda.empty_cache()
        seam["note"] = ("HipDetectorBackend.call_detector: host float32 tile in (7.1 MB), numpy heatmap [1,10,192,192] + features [1,100,192,192] out "
                        "(16.2 MB), batch 1, synchronous -- the reference's own calling convention; compare cpu_baseline.b1_fwd_nms_images_per_s")
        if not args.no_seam2:
            result["seam2_call_detector_b1"] = seam
            # ---- seam 3: one whole page through PageDetector (tile gather, batched forwards on two lanes, decode, canvases, page merge;
            # host page in, boxes out) -- what OCR_Processer.run_detector does per page (process_ocr_base.py:496-540)
            try:
                from findtextcenternet_amd import PageDetector
                page_u8 = synth.page_uint8(31, 3508, 2480)                              # A4 at 300 dpi
                pd = PageDetector(det, step_ratio=0.6, cut_off=0.4, batch=B, max_boxes=4096, device=str(dev), lanes=args.lanes)
                pd.detect_page(page_u8)
                import findtextcenternet_amd.page as page_mod
                merge_fn, merge_s = page_mod.page_merge_gpu, []

                def timed_merge(*a_, **k_):                                               # the greedy page-level selection, timed on its own
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    r_ = merge_fn(*a_, **k_)
                    torch.cuda.synchronize()
                    merge_s.append(time.perf_counter() - t1)
                    return r_
                page_mod.page_merge_gpu = timed_merge
                ts_ = []
                try:
                    for _ in range(3):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        loc, _gf, _li, _se = pd.detect_page(page_u8)
                        ts_.append(time.perf_counter() - t0)
                finally:
                    page_mod.page_merge_gpu = merge_fn
                from findtextcenternet_amd.page import padded_page_size, tile_origins
                ph, pw = padded_page_size(3508, 2480, pd.stepx, pd.stepy)
                nt = len(tile_origins(ph, pw, pd.stepx, pd.stepy))
                med = sorted(ts_)[1]
                mg = sorted(merge_s)[len(merge_s) // 2] if merge_s else 0.0
                # ... and the same selection on what a trained detector leaves on a real page: ~2 k glyph-sized candidates in clusters
                rng_ = np.random.Generator(np.random.PCG64(12))
                n2 = 2000
                cen = rng_.uniform([0, 0], [pw, ph], size=(n2 // 6, 2))
                b2 = np.zeros((n2, 9), np.float32)
                b2[:, 0] = rng_.uniform(0.3, 1.0, n2)
                b2[:, 1] = cen[rng_.integers(0, len(cen), n2), 0] + rng_.normal(0, 14, n2)
                b2[:, 2] = cen[rng_.integers(0, len(cen), n2), 1] + rng_.normal(0, 14, n2)
                b2[:, 3:5] = np.exp(rng_.uniform(np.log(10), np.log(70), (n2, 2)))
                a2 = (torch.from_numpy(b2).to(dev), torch.zeros((n2, 100), device=dev), torch.from_numpy(page_u8).to(dev).float()[:ph, :pw].contiguous()
                      if page_u8.shape[0] >= ph and page_u8.shape[1] >= pw else torch.full((ph, pw, 3), 255.0, device=dev),
                      torch.zeros((7, ph // 4, pw // 4), device=dev), 0.4)
                merge_fn(*a2)
                t2 = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    k2 = merge_fn(*a2)[0].shape[0]
                    torch.cuda.synchronize()
                    t2.append(time.perf_counter() - t1)
                result["seam3_page_a4_300dpi"] = {"ms_per_page_median": round(1000 * med, 2), "ms_page_merge": round(1000 * mg, 2),
                                                  "ms_page_merge_2k_candidates": round(1000 * sorted(t2)[2], 3), "kept_of_2k": int(k2), "tiles": nt,
                                                  "tiles_per_s": round(nt / med, 1), "tiles_per_s_without_page_merge": round(nt / max(1e-9, med - mg), 1),
                                                  "boxes": int(len(loc)),
                                                  "note": "PageDetector.detect_page(uint8 3508x2480 synthetic page): host page in, merged boxes out, synchronous; "
                                                          "the random-init network finds ~1600 peaks per tile (56 k candidates per page), which is what the "
                                                          "page-level selection (process_ocr_base.py:559-650; parallel since round 4: neighbour lists + rank-ordered "
                                                          "resolution, in-tree rank / median kernels) is timed on; ms_page_merge_2k_candidates = the same call on 2000 "
                                                          "clustered glyph-sized boxes"}
                del pd
            except Exception as ex:                                                          # (never let the extra record break the line)
                result["seam3_page_a4_300dpi"] = {"error": repr(ex)[:200]}
        if not args.no_cpu_baseline:
            cpu, o_hm, o_ft = cpu_baseline({k: v for k, v in sd.items()}, args.cpu_budget)
            result["cpu_baseline"] = cpu
            result["parity"] = {"reference": "CPU oracle (restatement of the reference path pinned by tests/golden), image 0 of the timed batch",
                                args.precision: parity_record(heat, feat, o_hm, o_ft)}
            for other in others:
                result["parity"][other] = parity_record(maps[other][0], maps[other][1], o_hm, o_ft)
        names = {"fp32": "fp32_parity_mode", "fp16x3": "fp16x3_parity_mode", "fp16": "fp16_mode", "bf16": "bf16_speed_mode"}
        for other in others:
            result[names[other]] = recs[other]
            if "parity" in result:
                result[names[other]].update({k: result["parity"][other][k] for k in ("heatmap_linf", "features_linf", "peak_set_identical", "peak_jaccard")})
        # ---- north_star_value: the fastest mode of this line that meets BASELINE.json's north_star tolerance ("within fp32 1e-3, peak indices
        # bit-exact") against the CPU oracle.  `value` above is the bf16 configuration BASELINE configs[1] names; bf16 arithmetic is outside that
        # tolerance (parity.bf16), so the contract-grade rate is quoted beside it, with the roofline of ITS dominant kernel.
        if "parity" in result:
            cands = []
            for prec_ in [args.precision] + list(others):
                pr = result["parity"].get(prec_)
                rate = result["value"] if prec_ == args.precision else recs[prec_]["images_per_s"]
                if pr and pr["heatmap_linf"] < 1e-3 and pr["features_linf"] < 1e-3 and pr["peak_set_identical"]:
                    cands.append((rate, prec_))
            if cands:
                rate, prec_ = max(cands)
                rec_ = result if prec_ == args.precision else recs[prec_]
                result["north_star_value"] = {
                    "value": rate, "unit": "images/s", "dtype": prec_, "tolerance": "heatmap and features L-inf < 1e-3 vs the CPU oracle, peak index set identical",
                    "heatmap_linf": result["parity"][prec_]["heatmap_linf"], "features_linf": result["parity"][prec_]["features_linf"],
                    "peak_set_identical": True, "ms_per_step": rec_["ms_per_step"],
                    "single_stream_images_per_s": rec_.get("single_stream_images_per_s", (rec_.get("single_stream") or {}).get("value")),
                    "path_frac_of_mfma_peak": rec_["path_frac_of_mfma_peak"], "roofline": rec_.get("roofline"),
                    "modes_in_tolerance": {p_: r_ for r_, p_ in sorted(cands, reverse=True)}}
                # the same figure inside `config` (the driver's record keeps `config`, not the line's extra keys): the rate of the fastest mode INSIDE
                # north_star's tolerance, its measured parity, and the fraction of ITS matrix roof (fp16x3: three fp16 MFMAs per product -> 833 TF)
                result["config"]["contract_mode"] = {
                    "dtype": prec_, "images_per_s": rate, "single_stream_images_per_s": result["north_star_value"]["single_stream_images_per_s"],
                    "heatmap_linf": result["parity"][prec_]["heatmap_linf"], "features_linf": result["parity"][prec_]["features_linf"],
                    "peak_set_identical": True, "frac_of_its_roof": rec_["path_frac_of_mfma_peak"], "roof_tflops": PEAK[prec_]}
        # ---- BASELINE configs[3]: "detector fwd + keyheatmap peak-NMS + 100-d feature gather end-to-end, batch=32, 1 GPU" -- the same step at batch 32
        # (the step above already is forward + NMS + decode + gather), bf16 and the contract-grade fp16x3; and configs[4]: a short train step record
        if not args.no_configs:
            cfg3 = {"workload": "BASELINE configs[3]: batch=32 synthetic 768x768x3 tiles, forward + NMS + GPU peak decode + 100-d feature gather, 1 GPU; "
                                "outputs [32,max,9] boxes, [32,max,100] features, counts"}
            try:
                B3 = 32
                x3 = torch.from_numpy(synth.noise_images(777, B3, 768, 768)).to(dev).permute(0, 3, 1, 2)
                tiles3 = tiles_to_device([TileGeom(0, 0, 768, 768, rect) for _ in range(B3)], dev, 192, 192)
                from findtextcenternet_amd import DetectorLanes
                for prec_, k3 in ((args.precision, 6), ("fp16x3", 4)):
                    if prec_ != args.precision and args.no_fp32:
                        continue
                    m3, d3 = (model, det) if prec_ == args.precision else make(prec_)
                    h3 = torch.empty((B3, 192, 192, 10), dtype=torch.float32, device=dev)
                    f3 = torch.empty((B3, 192, 192, 100), dtype=torch.float32, device=dev)
                    w3 = DecodeWorkspace(B3, 192, 192, 100, args.max_boxes, dev)

                    def step3():
                        with torch.no_grad():
                            d3.forward_nhwc(x3, out=(h3, f3))
                        return decode_peaks(h3, f3, tiles3, cut_off=0.4, max_boxes=args.max_boxes, logit_cut=lcut, workspace=w3)
                    step3(); step3()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(k3):
                        o3 = step3()
                    torch.cuda.synchronize()
                    e3 = time.perf_counter() - t0
                    r3 = {"single_stream_images_per_s": round(B3 * k3 / e3, 2), "single_stream_ms_per_step": round(1000 * e3 / k3, 3), "steps": k3,
                          "mean_peaks_per_tile": round(float(o3.counts.float().mean().item()), 1)}
                    del h3, f3, w3
                    if args.lanes > 1:
                        ln3 = DetectorLanes(d3, B3, 768, 768, lanes=args.lanes, max_boxes=args.max_boxes, device=dev)
                        for _ in range(args.lanes):
                            ln3.submit(x3, tiles3, cut_off=0.4, logit_cut=lcut)
                        ln3.wait()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(k3):
                            ln3.submit(x3, tiles3, cut_off=0.4, logit_cut=lcut)
                        ln3.wait()
                        torch.cuda.synchronize()
                        e4 = time.perf_counter() - t0
                        r3.update({"images_per_s": round(B3 * k3 / e4, 2), "ms_per_step": round(1000 * e4 / k3, 3), "lanes": args.lanes})
                        del ln3
                    else:
                        r3.update({"images_per_s": r3["single_stream_images_per_s"], "ms_per_step": r3["single_stream_ms_per_step"], "lanes": 1})
                    r3["path_frac_of_mfma_peak"] = round(r3["images_per_s"] * GFLOP_PER_IMAGE / 1000 / PEAK[prec_], 4)
                    cfg3[prec_] = r3
                    if prec_ != args.precision:
                        del m3, d3
                    torch.cuda.empty_cache()
                del x3
            except Exception as ex:
                cfg3["error"] = repr(ex)[:200]
            result["config3_b32"] = cfg3
            cm = result["config"].get("contract_mode")
            if cm and isinstance(cfg3.get(cm["dtype"]), dict):
                cm["config3_b32_images_per_s"] = cfg3[cm["dtype"]]["images_per_s"]
            try:
                lanes = None                                     # (the lanes' arenas go back to the allocator before the train step's 30 GB arena is made)
                torch.cuda.empty_cache()
                ta = argparse.Namespace(batch=B, steps=5, warmup=2, precision=args.precision, no_profile=args.no_profile, no_cpu_baseline=True)
                tr = train_bench(ta, 0, local_rank, 1, emit=False)
                result["train_step"] = {k: tr[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "ms_per_step_median", "dtype", "finite",
                                                           "loss_last_step", "path_tflops_per_gpu", "path_frac_of_mfma_peak", "roofline", "kernel_time_by_label_ms",
                                                           "forward_ms_sum_of_kernels", "backward_ms_sum_of_kernels", "workspace_gb") if k in tr}
                result["train_step"]["workload"] = tr["config"]["workload"]
            except Exception as ex:
                result["train_step"] = {"error": repr(ex)[:200]}
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit_last_line(result)


if __name__ == "__main__":
    main()
