#!/usr/bin/env python3
"""Per-kernel-label summary of rocprofv3 PMC passes over `python bench.py ...` (tools/profile_round.sh).

    python profiles/pmc_kernels.py gpurun_out/prof_<tag> --tag <tag> [--batch 8 --precision bf16]

Writes profiles/<tag>_bf16_b8_pmc_kernels.json (+ .txt) and profiles/<tag>_bf16_b8_pmc_traffic.json.

Dispatches are mapped to plan ops by sequence: every forward starts with the stem kernel and launches the plan's kernels in
plan order (an SE op is two launches, one after FTC_OP_MBHEAD), so dispatch k after a stem dispatch belongs to a known op; its label is what
ftc_op_kernel_label reports for the plan the library builds for the same shape (host-only call, no GPU needed).

Counters and corrections (/opt/skills/guides/MI355X_MICROARCH.md):
  FETCH_SIZE / WRITE_SIZE  KB per dispatch, separate passes; gfx950 counts a 128-byte request as 64 B for wide coalesced reads, so
                           traffic = 2*FETCH + WRITE (upper bound where reads are narrower); Infinity-Cache hits are included.
  SQ_VALU_MFMA_BUSY_CYCLES cycles an MFMA pipe is busy, summed over the chip's 1024 SIMDs (32 per 32x32x16 bf16 MFMA);
  GRBM_GUI_ACTIVE          busy cycles summed over the 8 XCDs (checked: 45.2 M for a 2.48 ms kernel = 8 x 5.65 M cycles at 2.27 GHz), so
                           mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); for the dominant kernel this reproduces the
                           FLOP-derived figure (64 M MFMAs x 32 cycles).
  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE   extra LDS cycles lost to bank conflicts / all LDS-array cycles.
  SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY  (quad-cycles) parked / issue-stalled / issuing, ~ disjoint shares of SQ_WAVE_CYCLES.
"""
import argparse
import collections
import csv
import ctypes as C
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def source_hash():
    sys.path.insert(0, ROOT)
    import bench
    return bench.source_hash()


def read_pass(path):
    """-> {counter: [(dispatch id, kernel name, value)]} sorted by dispatch id."""
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        out[r["Counter_Name"]].append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    for v in out.values():
        v.sort()
    return out


def forwards(rows, n_k):
    out, i = [], 0
    while i < len(rows):
        if "stem_kernel" in rows[i][1] and i + n_k <= len(rows):
            out.append(rows[i:i + n_k])
            i += n_k
        else:
            i += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--size", type=int, default=768)
    a = ap.parse_args()

    from findtextcenternet_amd import _lib as L
    from findtextcenternet_amd.model import FtcModel
    from findtextcenternet_amd.weights import deterministic_state_dict
    lib = L.load()
    pl = FtcModel(deterministic_state_dict(0), a.precision, "xl").plan(a.batch, a.size, a.size, False)
    buf = C.create_string_buffer(128)
    labels, op_of = [], []
    for i in range(len(pl.ops)):
        lib.ftc_op_kernel_label(C.byref(pl.ops[i]), buf, 128)
        n = 2 if pl.ops[i].kind == L.OP_SE and not (pl.ops[i].flags & L.FLAG_SE_HPART) else 1      # (SE after FTC_OP_MBHEAD: fc1 is done, one launch)
        labels += [buf.value.decode()] * n
        op_of += [i] * n
    n_k = len(labels)

    agg = collections.defaultdict(lambda: collections.defaultdict(float))       # label -> counter -> sum over launches of ONE forward
    nfw = collections.defaultdict(int)
    kern_name = {}
    for path in sorted(glob.glob(os.path.join(a.dir, "*_counter_collection.csv")) + glob.glob(os.path.join(a.dir, "*", "*_counter_collection.csv"))):
        for counter, rows in read_pass(path).items():
            fw = forwards(rows, n_k)
            if not fw:
                print(f"warning: no complete forward for {counter} in {path}")
                continue
            key = (counter, os.path.basename(path))
            for seq in fw:
                for k, (_, kname, val) in enumerate(seq):
                    stem = labels[k].split("<")[0].split("+")[0].replace("se_fc1", "se_fc").replace("se_gate", "se_fc2").replace("memset", "fillBuffer").replace("conv_igemm_glds", "glds").replace("conv3x3_halo", "halo")
                    if stem not in kname:
                        raise SystemExit(f"{path}: dispatch {k} is {kname!r} but the plan expects {labels[k]!r}")
                    agg[labels[k]][key] += val
                    kern_name[labels[k]] = kname[:100]
            nfw[key] = len(fw)
    launches = collections.Counter()
    alg_bytes, alg_flops = collections.defaultdict(float), collections.defaultdict(float)
    seen = set()
    for k, lab in enumerate(labels):
        if op_of[k] not in seen:
            seen.add(op_of[k])
            launches[lab] += 1
            alg_bytes[lab] += pl.meta[op_of[k]].bytes
            alg_flops[lab] += pl.meta[op_of[k]].flops
    by = {}
    for lab in launches:
        c = {}
        for (counter, fname), v in agg[lab].items():
            c.setdefault(counter, []).append(v / nfw[(counter, fname)])          # per forward, all launches of the label
        c = {k: sum(v) / len(v) for k, v in c.items()}                          # GRBM_GUI_ACTIVE appears in two passes: average
        rec = {"launches_per_forward": launches[lab], "kernel": kern_name.get(lab, ""),
               "algorithmic_bytes_per_launch": int(alg_bytes[lab] / launches[lab]), "algorithmic_gflop_per_launch": round(alg_flops[lab] / launches[lab] / 1e9, 3)}
        n = launches[lab]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rec["fetch_kb_per_launch"] = round(c["FETCH_SIZE"] / n, 1)
            rec["write_kb_per_launch"] = round(c["WRITE_SIZE"] / n, 1)
            rec["traffic_bytes"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / n)
            rec["traffic_over_algorithmic"] = round(rec["traffic_bytes"] / max(1, rec["algorithmic_bytes_per_launch"]), 2)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            rec["mfma_util"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
            rec["gui_active_cycles_per_launch"] = int(c["GRBM_GUI_ACTIVE"] / 8 / n)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            rec["wave_cycles_parked_frac"] = round(c.get("SQ_WAIT_ANY", 0) / w, 3)
            rec["wave_cycles_issue_stall_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0) / w, 3)
            rec["wave_cycles_issuing_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / w, 3)
            rec["wave_cycles_lds_issue_stall_frac"] = round(c.get("SQ_WAIT_INST_LDS", 0) / w, 3)
        for k2 in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU", "SQ_WAVES"):
            if k2 in c:
                rec[k2.lower() + "_per_launch"] = int(c[k2] / n)
        if rec.get("sq_insts_mfma_per_launch"):
            rec["valu_per_mfma"] = round(rec.get("sq_insts_valu_per_launch", 0) / rec["sq_insts_mfma_per_launch"], 2)
            rec["lds_per_mfma"] = round(rec.get("sq_insts_lds_per_launch", 0) / rec["sq_insts_mfma_per_launch"], 2)
        by[lab] = rec
    sh = source_hash()
    note = ("rocprofv3 PMC passes (counters only with --kernel-trace; FETCH_SIZE and WRITE_SIZE in separate passes) over `python bench.py "
            f"--steps 3 --warmup 2 --no-cpu-baseline --no-fp32 --no-profile` ({a.precision}, batch {a.batch}) on MI355X; per launch, averaged "
            "over every launch that carries the label.  traffic_bytes = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes; gfx950 FETCH_SIZE counts "
            "128-B requests as 64 B; Infinity-Cache hits included); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs).")
    base = os.path.join(ROOT, "profiles", f"{a.tag}_{a.precision}_b{a.batch}")
    json.dump({"note": note, "source_hash": sh, "batch": a.batch, "precision": a.precision, "by_label": by}, open(base + "_pmc_kernels.json", "w"), indent=1)
    json.dump({"note": note, "source_hash": sh, "batch": a.batch, "precision": a.precision,
               "by_label": {k: {kk: v[kk] for kk in ("launches_per_forward", "fetch_kb_per_launch", "write_kb_per_launch", "traffic_bytes",
                                                      "algorithmic_bytes_per_launch", "kernel") if kk in v} for k, v in by.items() if "traffic_bytes" in v}},
              open(base + "_pmc_traffic.json", "w"), indent=1)
    lines = [f"# {note}", f"# source_hash {sh}",
             f"{'n':>3} {'GFLOP':>9} {'alg MB':>9} {'HBM MB':>9} {'x alg':>6} {'MFMA%':>6} {'LDScf%':>6} {'park%':>6} {'stall%':>6} {'issue%':>6} {'VALU/MFMA':>9} {'LDS/MFMA':>8}  label"]
    for lab, r in sorted(by.items(), key=lambda kv: -kv[1].get("gui_active_cycles_per_launch", 0) * kv[1]["launches_per_forward"]):
        lines.append(f"{r['launches_per_forward']:3d} {r['algorithmic_gflop_per_launch']:9.2f} {r['algorithmic_bytes_per_launch'] / 1e6:9.2f} "
                     f"{r.get('traffic_bytes', 0) / 1e6:9.2f} {r.get('traffic_over_algorithmic', 0):6.2f} {100 * r.get('mfma_util', 0):6.1f} "
                     f"{100 * r.get('lds_bank_conflict_frac', 0):6.1f} {100 * r.get('wave_cycles_parked_frac', 0):6.1f} "
                     f"{100 * r.get('wave_cycles_issue_stall_frac', 0):6.1f} {100 * r.get('wave_cycles_issuing_frac', 0):6.1f} "
                     f"{r.get('valu_per_mfma', 0):9.2f} {r.get('lds_per_mfma', 0):8.2f}  {lab}")
    open(base + "_pmc_kernels.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:25]))


if __name__ == "__main__":
    main()
