#!/usr/bin/env python3
"""Stream-resolved view of the two-lane steady state from a rocprofv3 --kernel-trace database (rocpd sqlite):

    rocprofv3 --kernel-trace -d D -o X -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-fp32 --no-profile --no-sustained --no-seam2
    python tools/lanes_trace.py D/X_results.db [out.txt]

Kernels are attributed to their HIP stream (queue); inside a window of whole forwards (third stem kernel .. last stem kernel) it prints,
per lane, busy / idle time, the time BOTH lanes have a kernel in flight, and per kernel class how much of its run time a kernel of the
OTHER lane was in flight as well -- which kernels really co-run, which only fill each other's gaps."""
import collections
import sqlite3
import sys


def union_len(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


def merged(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def overlap_with(s, e, mv, idx):
    """length of [s, e) covered by the merged interval list mv; idx = moving start index (callers walk in time order)"""
    tot = 0
    i = idx[0]
    while i < len(mv) and mv[i][1] <= s:
        i += 1
    idx[0] = i
    while i < len(mv) and mv[i][0] < e:
        tot += min(e, mv[i][1]) - max(s, mv[i][0])
        i += 1
    return tot


def short(name):
    n = name.split("(")[0]
    for k in ("conv3x3_wl1", "conv3x3_halo", "conv_igemm_glds", "conv_igemm", "mbconv_slice", "se_fc2", "se_fc1", "dwconv_strip", "dwconv", "upcat", "tapsum",
              "stem", "nms", "decode_select", "decode_rank", "thin_conv"):
        if k in n:
            return k
    return n[-40:]


def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    sc = "start" if "start" in cols else "start_timestamp"
    ec = "end" if "end" in cols else "end_timestamp"
    qc = next((q for q in ("stream_id", "stream", "queue_id", "queue") if q in cols), None)
    if qc is None:
        raise SystemExit(f"no stream / queue column in the kernels view: {cols}")
    rows = c.execute(f"select name, {sc}, {ec}, {qc} from kernels order by {sc}").fetchall()
    per = collections.defaultdict(list)
    for name, s, e, q in rows:
        per[q].append((s, e, name))
    # the two-lane leg of the bench: the longest run of forwards (stem kernels, in time order) whose streams ALTERNATE between two ids
    stem_seq = sorted((s, q) for q in per for (s, e, n) in per[q] if "stem_kernel" in n)
    best, cur = (0, 0), 0
    for i in range(1, len(stem_seq) + 1):
        ok = i < len(stem_seq) and stem_seq[i][1] != stem_seq[i - 1][1] and (i < 2 or stem_seq[i][1] == stem_seq[i - 2][1])
        if not ok:
            if i - cur > best[1] - best[0]:
                best = (cur, i)
            cur = i
    run = stem_seq[best[0]:best[1]]
    if len(run) < 6:
        raise SystemExit(f"no two-lane leg in the trace (streams: { {q: len(v) for q, v in per.items()} })")
    lanes = sorted({q for _, q in run})
    stems = [s for s, _ in run]
    w0, w1 = stems[2], stems[-1]
    lines = [f"# {db}: streams by column `{qc}`; window = third .. last stem kernel = {(w1 - w0) / 1e6:.3f} ms, {sum(1 for s in stems if w0 <= s < w1)} forwards "
             f"({(w1 - w0) / 1e6 / max(1, sum(1 for s in stems if w0 <= s < w1)):.3f} ms per forward)"]
    iv = {q: [(max(s, w0), min(e, w1), n) for (s, e, n) in per[q] if e > w0 and s < w1] for q in lanes}
    mv = {q: merged([(s, e) for s, e, _ in iv[q]]) for q in lanes}
    busy = {q: sum(e - s for s, e in mv[q]) for q in lanes}
    both = 0
    idx = [0]
    for s, e in mv[lanes[0]]:
        both += overlap_with(s, e, mv[lanes[1]], idx)
    any_busy = union_len([(s, e) for q in lanes for s, e in mv[q]])
    W = w1 - w0
    for q in lanes:
        lines.append(f"lane (stream {q}): kernels {len(iv[q])}, busy {busy[q] / 1e6:.3f} ms = {100 * busy[q] / W:.1f} % of the window, idle {100 - 100 * busy[q] / W:.1f} %")
    lines.append(f"both lanes have a kernel in flight: {both / 1e6:.3f} ms = {100 * both / W:.1f} % of the window;  neither: {100 * (W - any_busy) / W:.1f} %;  "
                 f"sum of kernel durations / window = {sum(busy.values()) / W:.3f}")
    lines.append(f"{'class':18s} {'launches':>8s} {'ms in window':>12s} {'avg us':>8s} {'other lane in flight':>20s}")
    agg = collections.defaultdict(lambda: [0, 0, 0])
    for q in lanes:
        o = lanes[1] if q == lanes[0] else lanes[0]
        idx = [0]
        for s, e, n in sorted(iv[q]):
            ov = overlap_with(s, e, mv[o], idx)
            a = agg[short(n)]
            a[0] += 1
            a[1] += e - s
            a[2] += ov
    for k, (n, t, ov) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:18s} {n:8d} {t / 1e6:12.3f} {t / n / 1e3:8.1f} {100 * ov / max(1, t):19.1f} %")
    # who runs under the dominant kernel: classes of the other lane's kernels in flight during conv3x3_wl1
    under = collections.Counter()
    for q in lanes:
        o = lanes[1] if q == lanes[0] else lanes[0]
        for s, e, n in iv[q]:
            if "conv3x3_wl1" in n and e - s > 1e6:
                for s2, e2, n2 in iv[o]:
                    if e2 > s and s2 < e:
                        under[short(n2)] += min(e, e2) - max(s, s2)
    tot_wl1 = sum(e - s for q in lanes for s, e, n in iv[q] if "conv3x3_wl1" in n and e - s > 1e6)
    lines.append(f"under the dominant kernel (conv3x3_wl1+top, {tot_wl1 / 1e6:.3f} ms in the window) the other lane runs: " +
                 ", ".join(f"{k} {v / 1e6:.3f} ms" for k, v in under.most_common(8)))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
