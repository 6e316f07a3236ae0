"""CPU: the linedetect request produced by findtextcenternet_amd.page is accepted by the REAL reference
parser -- the `linedetect` CLI built from /root/reference/textline_detect by oracle/Makefile into
oracle/_ref/ (kind "reference").  Skipped where the binary was not built."""
import os
import subprocess

import numpy as np
import pytest

from findtextcenternet_amd import page

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "linedetect")
G = os.path.join(ROOT, "tests", "golden")


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/linedetect not built (make -C oracle)")
def test_reference_linedetect_accepts_our_request():
    # a synthetic page with three text lines of evenly spaced glyph boxes
    mh, mw = 192, 192
    lines = np.zeros((mh, mw), np.float32)
    seps = np.zeros((mh, mw), np.float32)
    rows = []
    for ly in (30, 80, 130):
        lines[ly - 1:ly + 2, 10:180] = 1.0
        for cx in range(60, 700, 40):
            rows.append([0.9, cx, ly * 4, 32.0, 34.0, 0.0, 0.0, 0.0, 0.0])
    loc = np.array(rows, np.float32)
    req = page.linedetect_request(loc, lines, seps)
    out = subprocess.run([BIN], input=req, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120).stdout
    res = page.linedetect_parse(out)
    assert len(res) >= len(rows)                                   # every box comes back (plus possible space markers)
    ids = sorted(r[0] for r in res if 0 <= r[0] < len(rows))
    assert ids == list(range(len(rows)))
    by_id = {r[0]: r for r in res}
    # boxes of one text line share (block, line index), the three lines get three distinct ones, and
    # the position inside the line (subidx) follows the reading order we sent
    line_of = [(by_id[i][1], by_id[i][2]) for i in range(len(rows))]            # (block, line index)
    per_line = len(rows) // 3
    assert len({tuple(line_of[k * per_line:(k + 1) * per_line]) for k in range(3)}) == 3
    for k in range(3):
        assert len(set(line_of[k * per_line:(k + 1) * per_line])) == 1
        assert [by_id[i][3] for i in range(k * per_line, (k + 1) * per_line)] == list(range(per_line))
