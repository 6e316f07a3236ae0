#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd sqlite database by default; this prints the
`--kernel-trace --stats` summary (per-kernel calls / total / average / share) as text."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"# source: {db}", f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches",
             f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel"]
    for name, n, s, a, mn, mx in rows:
        lines.append(f"{n:7d} {s / 1e6:10.3f} {a / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * s / tot:6.2f}  {name}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
