#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd SQLite output (ROCm 7.2 default format) as text:
per (kernel, grid) launch count / avg / min / max duration, and PMC counter means.
Usage: rocpd_summary.py <results.db> [more.db ...]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "")


def main():
    for path in sys.argv[1:]:
        c = sqlite3.connect(path)
        print("## %s" % path)
        print("%-28s %10s %6s %12s %12s %12s %10s %6s %6s" % (
            "kernel", "grid", "calls", "avg_us", "min_us", "max_us", "total_ms", "vgpr", "lds"))
        q = ("select name, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z), count(*), "
             "avg(duration), min(duration), max(duration), sum(duration), max(vgpr_count+accum_vgpr_count), "
             "max(lds_size) from kernels group by 1,2 order by 7 desc")
        for r in c.execute(q):
            print("%-28s %10d %6d %12.1f %12.1f %12.1f %10.3f %6d %6d" % (
                short(r[0]), r[1], r[2], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6] / 1e6, r[7], r[8]))
        try:
            rows = list(c.execute(
                "select k.name, k.grid_x*k.grid_y*k.grid_z/(k.workgroup_x*k.workgroup_y*k.workgroup_z), "
                "p.counter_name, count(*), avg(p.counter_value) from pmc_events p join kernels k "
                "on p.dispatch_id = k.dispatch_id group by 1,2,3 order by 1,2,3"))
        except sqlite3.OperationalError:
            rows = []
        if rows:
            print("%-28s %10s %-14s %6s %16s" % ("kernel", "grid", "counter", "calls", "mean_value"))
            for r in rows:
                print("%-28s %10d %-14s %6d %16.1f" % (short(r[0]), r[1], r[2], r[3], r[4]))
        print()


if __name__ == "__main__":
    main()
