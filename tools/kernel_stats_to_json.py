#!/usr/bin/env python3
"""profiles/<tag>_kernel_stats.txt (tools/rocpd_summary.py output of a `rocprofv3 --kernel-trace
--stats` run of bench.py) -> profiles/kernel_stats.json: the profiler's average duration of the
headline step's kernels (K=64, D=32, 3891 windows; picked by kernel name + launch grid like
tools/pmc_to_traffic.py).  bench.py computes roofline.frac from the `stats` entry, so that the
figure in the bench line is the one a reader recomputes from profiles/ (the live HIP-event figure
stays beside it as frac_events).
Usage: kernel_stats_to_json.py profiles/r04x_kernel_stats.txt > profiles/kernel_stats.json"""
import json
import re
import sys

from pmc_to_traffic import HEADLINE, F32


def read_rows(path):
    rows = []
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+(\d+)\s*$", line)
        if m:
            rows.append({"kernel": m.group(1), "grid": int(m.group(2)), "calls": int(m.group(3)),
                         "avg_us": float(m.group(4)), "min_us": float(m.group(5)), "max_us": float(m.group(6)),
                         "vgpr": int(m.group(8)), "lds": int(m.group(9))})
    return rows


def pick(rows, table):
    out = {}
    for fam, (rx, grid) in table.items():
        for r in rows:
            if r["grid"] == grid and re.fullmatch(rx, r["kernel"]):
                out[fam] = {k: r[k] for k in ("kernel", "grid", "calls", "avg_us", "min_us", "max_us")}
                out[fam]["workgroups"] = out[fam].pop("grid")
    return out


def main(path):
    rows = read_rows(path)
    out = {"_note": "rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2 --reps 2 --no-cpu-baseline` "
                    "(tools/profile_round.sh); average kernel durations of the headline step; source %s" % path}
    out.update(sorted(pick(rows, HEADLINE).items()))
    f32 = pick(rows, F32)
    if f32:
        out["_f32_mode"] = f32
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
