#!/usr/bin/env python3
"""Timeline of one device-resident SVI iteration from a rocprofv3 kernel trace (rocpd .db) of
tools/svi_e2e.py: kernels between two consecutive end-of-iteration markers of the 64-window loop (k_svi_elbo; since
round 6 the ELBO total rides in k_svi_vlb's last workgroup when the loop runs on device-side counters: k_finalize then)."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) "
                      "from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_svi_elbo" in r[0]]
if len(idx) < 8:
    idx = [i for i, r in enumerate(rows) if r[0].startswith("k_finalize")]
# an iteration of the 64-window loop: contains a k_wave_lin4 launch
sel = [(a, b) for a, b in zip(idx, idx[1:]) if any(("k_wave_lin" in r[0] or "k_sweep_stats" in r[0]) for r in rows[a:b])]
# (bench.py runs several loops -- fp64, fp32 mode -- and other work in between: take the iteration of median
#  SPAN among those that look like steady-state iterations, not the middle one of the list)
sel = [ab for ab in sel if rows[ab[1]][2] - rows[ab[0]][2] < 2e6] or sel
sel.sort(key=lambda ab: rows[ab[1]][2] - rows[ab[0]][2])
a, b = sel[len(sel) // 2]
t0 = rows[a][2]
last_end = t0
for name, st, en, grid in rows[a + 1:b + 1]:
    print("%-40s grid %6d  start +%7.1f us  dur %7.1f us  end +%7.1f" % (
        re.sub(r"\(.*", "", name)[:40], grid, (st - t0) / 1e3, (en - st) / 1e3, (en - t0) / 1e3))
print("iteration span %.1f us" % ((rows[b][2] - t0) / 1e3))
