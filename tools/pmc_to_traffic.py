#!/usr/bin/env python3
"""profiles/<tag>_pmc_hbm.txt (tools/rocpd_summary.py output of the FETCH_SIZE and WRITE_SIZE
passes) -> profiles/pmc_traffic.json, the per-launch HBM bytes bench.py reports as
roofline.traffic.  Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
both counters are KiB; FETCH_SIZE on gfx950 reports half of a wide coalesced read stream and
is doubled; WRITE_SIZE is exact.  Only the launches of the bench step (the launch shape with
the largest counter value of each kernel family) are used.
Usage: pmc_to_traffic.py profiles/r01e_pmc_hbm.txt > profiles/pmc_traffic.json"""
import json
import re
import sys

FAMILY = [("k_emission", "emission"), ("k_stats", "stats"), ("k_sweeps_lin<", "forward_backward"),
          ("k_fwd_mfma", "forward_backward"), ("k_bwd_mfma", "posterior"), ("k_finalize", "finalize")]


def main(path):
    shapes = {}   # (family, kernel, grid) -> {counter: mean KiB}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(\d+)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s*$", line)
        if not m:
            continue
        name, grid, ctr, val = m.group(1), int(m.group(2)), m.group(3), float(m.group(5))
        for pref, fam in FAMILY:
            if name.startswith(pref):
                shapes.setdefault((fam, name, grid), {})[ctr] = val
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, FETCH doubled "
                    "per MI355X_MICROARCH.md (HBM section); bench workload K=64 D=32 Lm=257 B=3891; source %s" % path}
    best = {}
    for (fam, name, grid), c in shapes.items():
        fb, wb = c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0, c.get("WRITE_SIZE", 0.0) * 1024.0
        if fam not in best or fb + wb > best[fam]["hbm_bytes_per_launch"]:
            best[fam] = {"kernel": name, "workgroups": grid, "fetch_bytes": fb, "write_bytes": wb,
                         "hbm_bytes_per_launch": fb + wb}
    out.update(sorted(best.items()))
    # the epoch step = one launch of each family's bench-shape kernel (+ the small ones, < 1 %)
    out["_step_total"] = {"hbm_bytes_per_step": sum(v["hbm_bytes_per_launch"] for v in best.values()),
                          "families": sorted(best)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
