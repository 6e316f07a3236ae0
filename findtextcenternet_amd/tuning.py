"""Measured kernel selection for the implicit-GEMM convolution ("measure, don't guess").

Every dense conv of the plan can run on several variants of the same kernel -- 7 tile shapes x
{register-staged, direct-to-LDS 2/3-slot ring} x K step {32, 64, 128} -- which all produce
bit-identical results (same K summation order) but differ up to 2x in speed depending on how many
workgroups the layer yields and how long its K loop is.  ``tune_plan`` times every legal variant of
every distinct conv shape of a plan on the GPU (single-op plans over the real operand buffers, HIP
events through ``ftc_plan_profile``) and records the fastest in a table keyed by the layer signature;
the library applies the choice (``ftc_op.aux0``) when it builds a plan (csrc/model.hip: the table is compiled in as
csrc/tuning_table.inc, generated from ``tuning_gfx950.json`` by build.py -- rebuild after re-tuning).  Shapes missing from
the table fall back to the heuristics in conv_igemm_impl.h.

    python -m findtextcenternet_amd.tuning --batch 8 --precision bf16 [--out path.json]
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Optional

from . import _lib as L

TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning_gfx950.json")
CFG_NAMES = ["192x128", "128x128", "96x128", "64x128", "128x64", "32x256", "64x64", "64x144", "80x144", "128x144", "96x144"]
N_GENERIC_CFGS = 7           # the others are the 144-pixel 1x1 kernel (csrc/conv1x1_px144.hip), chosen by aux0 = 8 .. 11 alone
_table: Optional[Dict[str, int]] = None


def signature(o, merge16: bool = False) -> str:
    """merge16: fp16 operands look up their bf16 twins (same kernels, same rate), as csrc/model.hip does."""
    d = (lambda t: L.BF16 if t == L.F16 else t) if merge16 else (lambda t: t)
    return (f"w{d(o.w_dtype)}i{d(o.in_dtype)}o{d(o.out_dtype)}_B{o.B}_{o.H}x{o.W}_c{o.Cin}of{o.Cin_total}_n{o.Cout}of{o.Cout_total}"
            f"_k{o.ksize}s{o.stride}_f{o.flags & ~(L.FLAG_KBLOCK32 | L.FLAG_PRESPLIT)}_a{o.act}" + (f"_g{o.groups}" if o.groups > 1 else ""))


def encode(cfg: int, stage: int, bk: int, halo: bool = False, splitk: int = 1) -> int:
    return (cfg + 1) | (stage << 4) | ({0: 0, 32: 1, 64: 2, 128: 3}[bk] << 8) | (64 if halo else 0) | ({1: 0, 2: 1, 4: 2}[splitk] << 10)


def describe(aux0: int) -> str:
    if aux0 == 0:
        return "default"
    if aux0 & 64:
        return f"halo,channels={CFG_NAMES[(aux0 & 15) - 1].split('x')[0]}" + (",bk=32" if (aux0 >> 8) & 3 == 1 else "")
    sk = [1, 2, 4, 1][(aux0 >> 10) & 3]
    if (aux0 & 15) - 1 >= N_GENERIC_CFGS:
        return f"px144,tile={CFG_NAMES[(aux0 & 15) - 1]}"
    return (f"tile={CFG_NAMES[(aux0 & 15) - 1]},stage={['auto', 'reg', 'dma2', 'dma3'][(aux0 >> 4) & 3]},bk={[0, 32, 64, 128][(aux0 >> 8) & 3]}"
            + (f",splitk={sk}" if sk > 1 else ""))


def load_table(path: str = TABLE_PATH) -> Dict[str, int]:
    global _table
    if _table is None:
        _table = {}
        if os.path.exists(path) and os.environ.get("FTC_NO_TUNING", "0") != "1":
            with open(path) as f:
                _table = {k: int(v) for k, v in json.load(f).get("choices", {}).items()}
    return _table


def apply(ops, table: Optional[Dict[str, int]] = None) -> int:
    """Writes tuned choices into ops[i].aux0 for the conv ops found in the table; returns how many."""
    table = load_table() if table is None else table
    n = 0
    for o in ops:
        if o.kind == L.OP_CONV:
            v = table.get(signature(o)) or table.get(signature(o, merge16=True))
            if v:
                o.aux0 = v
                n += 1
    return n


def tune_train_step(ts, B: int, H: int, W: int, reps: int = 5, verbose: bool = False, known: Optional[Dict[str, int]] = None) -> Dict[str, int]:
    """The convolutions of the TRAIN plan (forward with raw weights, data gradients): findtextcenternet_amd.train_step.TrainStep `ts`
    after one forward_backward at this shape (buffers filled), plan built with FTC_NO_TUNING=1."""
    import numpy as np
    import torch
    lib = L.load()
    plan = ts.plan_for(B, H, W)
    dev = ts.dev
    x = torch.rand((B, H, W, 3), dtype=torch.float32, device=dev)
    bases = (C.c_void_p * L.NUM_BASES)(None, ts.workspace.data_ptr(), ts.blob.data_ptr(), x.data_ptr(), None, None, ts.grads.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    choices: Dict[str, int] = {}
    ms = (C.c_float * 1)()
    for i in range(plan["n_ops"]):
        o = plan["ops"][i]
        if o.kind != L.OP_CONV:
            continue
        key = signature(o, merge16=True)
        if key in choices or (known is not None and key in known):      # (--merge: only the signatures the table does not hold yet)
            continue
        best, best_t, base_t = 0, 1e30, None
        one = (L.Op * 1)()
        for aux in [0] + candidates(o):
            C.memmove(C.byref(one[0]), C.byref(o), C.sizeof(L.Op))
            one[0].aux0 = aux
            h = C.c_void_p()
            if lib.ftc_plan_create(one, 1, plan["workspace_bytes"], ts.blob.numel(), C.byref(h)) != 0:
                continue
            ts_ = []
            ok = lib.ftc_plan_run(h, bases, stream, 0, -1) == 0
            for _ in range(reps if ok else 0):
                if lib.ftc_plan_profile(h, bases, stream, ms) != 0:
                    ok = False
                    break
                ts_.append(ms[0])
            lib.ftc_plan_destroy(h)
            if not ok:
                continue
            t = float(np.median(ts_))
            if aux == 0:
                base_t = t
            if t < best_t * 0.98:
                best, best_t = aux, t
        choices[key] = best
        if verbose:
            base = f"{base_t * 1e3:8.1f}" if base_t is not None else " illegal"
            print(f"{key:74s} default {base} us -> {best_t * 1e3:8.1f} us  {describe(best)}", flush=True)
    return choices


def candidates(o) -> List[int]:
    bks = [32] if o.w_dtype == L.F32 else [b for b in (32, 64, 128) if o.Cin % b == 0 or b == 32]
    if o.w_dtype == L.BF16 and o.Cin % 64 == 0:
        bks = [b for b in bks if b != 32]                 # 32 never wins when 64 is legal
    out = []
    for cfg in range(N_GENERIC_CFGS):
        tn = int(CFG_NAMES[cfg].split("x")[0])
        if tn > 2 * max(32, o.Cout):                      # more than half the tile rows would be padding
            continue
        for bk in bks:
            for stage in (1, 2, 3):
                out.append(encode(cfg, stage, bk))
    if o.w_dtype == L.BF16 and o.in_dtype == L.BF16 and o.Cin * o.ksize * o.ksize >= 768:
        for cfg in (3, 4, 6):                             # intra-workgroup split-K on the small tiles
            for bk in [b for b in bks if b >= 64]:
                for sk in (2, 4):
                    out.append(encode(cfg, 1, bk, splitk=sk))
    if o.ksize == 1 and o.in_dtype == o.w_dtype and o.out_dtype == L.F32 and (o.Ho * o.Wo) % 144 == 0 and (o.w_dtype != L.F32 or o.flags & L.FLAG_PRESPLIT):
        out += [8, 9, 10, 11]                              # 144-pixel tiles (the illegal ones are refused at plan creation)
    if o.ksize == 3 and o.stride == 1:
        for cfg in (0, 1, 3):                             # LDS-halo kernel with 192 / 128 / 64 channel tiles
            out.append(encode(cfg, 0, 0, halo=True))
            if o.w_dtype == L.BF16 and o.Cin % 64 == 0:   # ... and with 64-byte rows (two workgroups per CU)
                out.append(encode(cfg, 0, 32, halo=True))
    return out


def tune_plan(engine, plan, reps: int = 5, verbose: bool = False, only: str = "") -> Dict[str, int]:
    """engine: detector._HipEngine with weights + workspace resident and one forward already run; plan: model.PlanView built with
    FTC_NO_TUNING=1 (so that aux0 holds the untuned defaults)."""
    import numpy as np
    import torch
    lib = L.load()
    dev = engine.wdev.device
    heat = torch.empty((plan.B, plan.h, plan.w, 10), dtype=torch.float32, device=dev)
    feat = torch.empty((plan.B, plan.h, plan.w, 100), dtype=torch.float32, device=dev)
    x = torch.rand((plan.B, plan.H, plan.W, 3), dtype=torch.float32, device=dev)
    bases = (C.c_void_p * L.NUM_BASES)(None, engine.workspace.data_ptr(), engine.wdev.data_ptr(), x.data_ptr(), heat.data_ptr(), feat.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(lib.ftc_plan_run(plan.handle, bases, stream, 0, -1), "forward (fills the activation buffers)")
    torch.cuda.synchronize()
    choices: Dict[str, int] = {}
    ms = (C.c_float * 1)()
    for i in range(len(plan.ops)):
        o = plan.ops[i]
        if o.kind != L.OP_CONV:
            continue
        key = signature(o)
        if key in choices:
            continue
        if only:
            import re
            if not re.search(only, key):
                continue
        best, best_t, base_t = 0, 1e30, None
        one = (L.Op * 1)()
        for aux in [0] + candidates(o):
            C.memmove(C.byref(one[0]), C.byref(o), C.sizeof(L.Op))
            one[0].aux0 = aux
            h = C.c_void_p()
            if lib.ftc_plan_create(one, 1, plan.workspace_bytes, engine.model.weights_bytes, C.byref(h)) != 0:
                continue                                     # not legal for this op
            ts = []
            ok = lib.ftc_plan_run(h, bases, stream, 0, -1) == 0
            for _ in range(reps if ok else 0):
                if lib.ftc_plan_profile(h, bases, stream, ms) != 0:
                    ok = False
                    break
                ts.append(ms[0])
            lib.ftc_plan_destroy(h)
            if not ok:                                       # variant refused by the runtime (e.g. resources): skip it
                continue
            t = float(np.median(ts))
            if aux == 0:
                base_t = t
            if t < best_t * 0.98:                            # keep the earlier (simpler) choice on ties
                best, best_t = aux, t
        choices[key] = best
        if verbose:
            base = f"{base_t * 1e3:8.1f}" if base_t is not None else " illegal"
            print(f"{key:70s} default {base} us -> {best_t * 1e3:8.1f} us  {describe(best)}", flush=True)
    return choices


def main():
    import argparse
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from findtextcenternet_amd import CenterNetDetector, TextDetectorModel, deterministic_state_dict
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[8])
    ap.add_argument("--precision", nargs="+", default=["bf16"])
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--out", default=TABLE_PATH)
    ap.add_argument("--merge", action="store_true", help="keep existing entries of --out")
    ap.add_argument("--filter", default="", help="only re-measure conv signatures matching this regular expression (e.g. _k1s1_)")
    ap.add_argument("--train", action="store_true", help="tune the convolutions of the TRAIN plan (TrainStep) instead of the inference plan")
    a = ap.parse_args()
    os.environ["FTC_NO_TUNING"] = "1"
    allc: Dict[str, int] = {}
    if a.merge and os.path.exists(a.out):
        allc = {k: int(v) for k, v in json.load(open(a.out)).get("choices", {}).items()}
    sd = deterministic_state_dict(0)
    if a.train:
        from findtextcenternet_amd import TrainStep, synth
        for prec in a.precision:
            model = TextDetectorModel(pre_weights=False, precision=prec)
            model.load_state_dict(sd)
            model = model.to("cuda").train()
            ts = TrainStep(model)
            for B in a.batch:
                x = torch.rand((B, a.size, a.size, 3), device="cuda").permute(0, 3, 1, 2)
                lab, idm = synth.train_labels(1, B, a.size // 4, a.size // 4)
                ts.zero_grad()
                ts.forward_backward(x, torch.from_numpy(lab).cuda(), torch.from_numpy(idm).cuda())
                torch.cuda.synchronize()
                allc.update(tune_train_step(ts, B, a.size, a.size, verbose=True, known=allc if a.merge else None))
            del ts, model
            torch.cuda.empty_cache()
        with open(a.out, "w") as f:
            json.dump({"device": "MI355X gfx950", "note": "aux0 per conv signature, measured by findtextcenternet_amd.tuning",
                       "choices": dict(sorted(allc.items()))}, f, indent=0)
        print("wrote", a.out, len(allc), "entries")
        return
    for prec in a.precision:
        model = TextDetectorModel(pre_weights=False, precision=prec)
        model.load_state_dict(sd)
        det = CenterNetDetector(model.detector).to("cuda").eval()
        for B in a.batch:
            x = torch.rand((B, a.size, a.size, 3), device="cuda").permute(0, 3, 1, 2)
            with torch.no_grad():
                det(x)
            eng = model.detector._engine
            plan = eng.plan(B, a.size, a.size, False)
            ch = tune_plan(eng, plan, verbose=True, only=a.filter)
            allc.update(ch)
        del det, model
        torch.cuda.empty_cache()
    with open(a.out, "w") as f:
        json.dump({"device": "MI355X gfx950", "note": "aux0 per conv signature, measured by findtextcenternet_amd.tuning",
                   "choices": dict(sorted(allc.items()))}, f, indent=0)
    print("wrote", a.out, len(allc), "entries -- rebuild the library (python -m findtextcenternet_amd.build) to compile them in")


if __name__ == "__main__":
    main()
