#!/usr/bin/env python3
"""In-stream view of one bench step from a rocprofv3 --kernel-trace database (rocpd sqlite):

    rocprofv3 --kernel-trace -d D -o X -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile
    python tools/stream_trace.py D/X_results.db [--labels]

Takes every complete step (stem kernel .. last kernel before the next stem), prints their wall times, and for the step with the
median wall the busy time per kernel name,
the idle gaps between consecutive kernels and the wall time of the step."""
import sqlite3
import sys
import collections


def main(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    sc = "start" if "start" in cols else "start_timestamp"
    ec = "end" if "end" in cols else "end_timestamp"
    rows = c.execute(f"select name, {sc}, {ec} from kernels order by {sc}").fetchall()
    stems = [i for i, r in enumerate(rows) if "stem_kernel" in r[0]]
    if len(stems) < 2:
        raise SystemExit("need at least two forwards in the trace")
    # every complete step in the trace (forward + decode kernels up to the next stem); the one with the median wall time is detailed
    steps = [rows[stems[i]:stems[i + 1]] for i in range(len(stems) - 1)]
    steps = [q for q in steps if len(q) == max(len(x) for x in steps)]          # (same kernel count: drops warm-up / other-mode forwards)
    walls = sorted((q[-1][2] - q[0][1], i) for i, q in enumerate(steps))
    print("walls of the", len(steps), "complete steps (ms):", " ".join(f"{w / 1e6:.3f}" for w, _ in sorted(walls, key=lambda x: x[1])))
    seq = steps[walls[(len(walls) - 1) // 2][1]]
    wall = seq[-1][2] - seq[0][1]
    busy = collections.Counter()
    cnt = collections.Counter()
    gaps = 0
    overlap = 0
    prev_end = seq[0][1]
    gap_list, prev_name = [], ""
    for name, s, e in seq:
        short = name.split("(")[0][-70:]
        busy[short] += e - s
        cnt[short] += 1
        if s > prev_end:
            gaps += s - prev_end
            gap_list.append((s - prev_end, prev_name, short))
        else:
            overlap += min(prev_end, e) - s
        prev_end = max(prev_end, e)
        prev_name = short
    print(f"step wall {wall / 1e6:.3f} ms   kernels {len(seq)}   sum of durations {sum(busy.values()) / 1e6:.3f} ms   "
          f"idle gaps {gaps / 1e6:.3f} ms   overlap {overlap / 1e6:.3f} ms")
    for g, a, b in sorted(gap_list, reverse=True)[:3]:
        print(f"  largest gaps: {g / 1e3:8.1f} us between ...{a[-40:]} and ...{b[-40:]}")
    for k, v in busy.most_common(40):
        print(f"  {v / 1e6:8.3f} ms  {cnt[k]:4d} x {v / cnt[k] / 1e3:8.1f} us  {k}")


if __name__ == "__main__":
    main(sys.argv[1])
