#!/usr/bin/env python3
"""GPU idle time between consecutive kernels of a rocprofv3 kernel trace (rocpd .db):
prints the dispatch sequence of one bench step with start offsets, durations and gaps."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) "
                      "from kernels order by start"))
# one bench step = from a k_exp_transpose to the next one, taken near the end of the run
idx = [i for i, r in enumerate(rows) if "k_exp_transpose" in r[0]]
sel = [i for i in idx if i + 8 < len(rows) and any(r[3] > 1000 for r in rows[i + 1:i + 7])]
i0 = sel[-3]
i1 = [i for i in idx if i > i0][0]
t0 = rows[i0][1]
prev_end = None
busy = 0
for name, st, en, grid in rows[i0:i1]:
    gap = 0 if prev_end is None else (st - prev_end) / 1e3
    print("%-34s grid %6d  +%9.1f us  dur %8.1f us  gap %7.1f us" % (
        re.sub(r"\(.*", "", name)[:34], grid, (st - t0) / 1e3, (en - st) / 1e3, gap))
    prev_end = en
    busy += en - st
print("step span %.1f us, kernel busy %.1f us, idle inside %.1f us (next step starts +%.1f us)" % (
    (prev_end - t0) / 1e3, busy / 1e3, (prev_end - t0 - busy) / 1e3, (rows[i1][1] - t0) / 1e3))
