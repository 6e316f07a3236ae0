#!/usr/bin/env python3
"""In-stream view of one bench step from a rocprofv3 --kernel-trace database (rocpd sqlite):

    rocprofv3 --kernel-trace -d D -o X -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile
    python tools/stream_trace.py D/X_results.db [--labels]

Takes the LAST forward (stem kernel .. last kernel before the next stem / end), prints busy time per kernel name,
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
    a, b = stems[-2], stems[-1]
    seq = rows[a:b]                       # one full step: forward + decode kernels up to the next stem
    wall = seq[-1][2] - seq[0][1]
    busy = collections.Counter()
    cnt = collections.Counter()
    gaps = 0
    overlap = 0
    prev_end = seq[0][1]
    for name, s, e in seq:
        short = name.split("(")[0][-70:]
        busy[short] += e - s
        cnt[short] += 1
        if s > prev_end:
            gaps += s - prev_end
        else:
            overlap += min(prev_end, e) - s
        prev_end = max(prev_end, e)
    print(f"step wall {wall / 1e6:.3f} ms   kernels {len(seq)}   sum of durations {sum(busy.values()) / 1e6:.3f} ms   "
          f"idle gaps {gaps / 1e6:.3f} ms   overlap {overlap / 1e6:.3f} ms")
    for k, v in busy.most_common(40):
        print(f"  {v / 1e6:8.3f} ms  {cnt[k]:4d} x {v / cnt[k] / 1e3:8.1f} us  {k}")


if __name__ == "__main__":
    main(sys.argv[1])
