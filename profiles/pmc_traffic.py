#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes over `python bench.py --no-profile ...` into per-kernel-label HBM traffic.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D -o X_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d D -o X_write -- python bench.py ... (same command)
    python profiles/pmc_traffic.py D/X_fetch_counter_collection.csv D/X_write_counter_collection.csv --batch 8 --precision bf16 --out profiles/X_pmc_traffic.json

Dispatches are mapped to plan ops by sequence: every forward starts with the stem kernel and launches exactly
ftc_plan_num_ops kernels in plan order, so dispatch k after a stem dispatch is op k.  The label of op k is what
ftc_op_kernel_label (host-only, works without a GPU) reports for the same plan built here; the tuning table is the
committed one, i.e. the one the profiled run used.

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE
counts a 128-B request as 64 B for wide coalesced reads, so traffic_bytes = 2*FETCH_KB*1024 + WRITE_KB*1024 (an upper
bound for kernels whose reads are not all 128-B requests).
"""
import argparse
import csv
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_pass(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def forwards(rows, n_ops):
    out = []
    i = 0
    while i < len(rows):
        if "stem_kernel" in rows[i][1] and i + n_ops <= len(rows):
            out.append(rows[i:i + n_ops])
            i += n_ops
        else:
            i += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    from findtextcenternet_amd import _lib as L
    from findtextcenternet_amd.model import FtcModel
    from findtextcenternet_amd.weights import deterministic_state_dict
    lib = L.load()
    sd = deterministic_state_dict(0)
    pl = FtcModel(sd, a.precision, "xl").plan(a.batch, a.size, a.size, False)      # the plan ftc_forward runs (host-only call)
    n_ops = len(pl.meta)
    buf = C.create_string_buffer(128)
    labels = []
    for i in range(n_ops):
        lib.ftc_op_kernel_label(C.byref(pl.ops[i]), buf, 128)
        lab = buf.value.decode()
        labels.extend([lab] * (2 if pl.ops[i].kind == L.OP_SE else 1))       # an SE op is two launches (fc1, fc2)
    op_of = []
    for i in range(n_ops):
        op_of.extend([i] * (2 if pl.ops[i].kind == L.OP_SE else 1))
    n_k = len(labels)

    agg = {}
    for path, counter, key in ((a.fetch_csv, "FETCH_SIZE", "fetch_kb"), (a.write_csv, "WRITE_SIZE", "write_kb")):
        fw = forwards(read_pass(path, counter), n_k)
        if not fw:
            raise SystemExit(f"no complete forward found in {path}")
        for seq in fw:
            for k, (_, kname, val) in enumerate(seq):
                stem = labels[k].split("<")[0].split("+")[0].replace("se_fc1", "se_fc").replace("conv_igemm_glds", "glds").replace("conv3x3_halo", "halo")
                if stem not in kname:
                    raise SystemExit(f"dispatch {k} is {kname!r} but the plan expects {labels[k]!r}")
                d = agg.setdefault(labels[k], {"fetch_kb": 0.0, "write_kb": 0.0, "n_fetch_kb": 0, "n_write_kb": 0,
                                               "algorithmic_bytes": 0.0, "launches_per_forward": 0, "kernel": kname})
                d[key] += val
                d["n_" + key] += 1
    seen = set()
    for k, lab in enumerate(labels):
        if op_of[k] not in seen:                                              # SE: bytes/launch counted once per op
            seen.add(op_of[k])
            agg[lab]["algorithmic_bytes"] += pl.meta[op_of[k]].bytes
            agg[lab]["launches_per_forward"] += 1
    by = {}
    for lab, d in agg.items():
        per_op = 2 if lab.startswith("se_") else 1                            # sum the two SE launches of one op
        f = per_op * d["fetch_kb"] / max(1, d["n_fetch_kb"])
        w = per_op * d["write_kb"] / max(1, d["n_write_kb"])
        by[lab] = {"launches_per_forward": d["launches_per_forward"], "fetch_kb": round(f, 2), "write_kb": round(w, 2),
                   "traffic_bytes": int(2 * f * 1024 + w * 1024),
                   "algorithmic_bytes": int(d["algorithmic_bytes"] / d["launches_per_forward"]),
                   "kernel": d["kernel"][:120]}
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over `python bench.py --steps 3 "
            f"--warmup 1 --no-cpu-baseline --no-profile` ({a.precision}, batch {a.batch}) on MI355X; all figures are per launch, "
            "averaged over every launch that carries the label.  traffic_bytes = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes; gfx950 "
            "FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); algorithmic_bytes = the plan's "
            "compulsory bytes (inputs + weights + outputs once) for the same launches.")
    json.dump({"note": note, "batch": a.batch, "precision": a.precision, "by_label": by}, open(a.out, "w"), indent=1)
    tot_t = sum(v["traffic_bytes"] * v["launches_per_forward"] for v in by.values())
    tot_a = sum(v["algorithmic_bytes"] * v["launches_per_forward"] for v in by.values())
    print(f"{len(by)} labels, forward traffic {tot_t / 1e9:.3f} GB vs algorithmic {tot_a / 1e9:.3f} GB")
    for lab, v in sorted(by.items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches_per_forward"])[:12]:
        print(f"  {v['launches_per_forward']:3d} x {v['traffic_bytes'] / 1e6:9.2f} MB (alg {v['algorithmic_bytes'] / 1e6:9.2f})  {lab}")


if __name__ == "__main__":
    main()
