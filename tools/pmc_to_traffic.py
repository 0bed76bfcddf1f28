#!/usr/bin/env python3
"""profiles/<tag>_pmc_hbm.txt (tools/rocpd_summary.py output of the FETCH_SIZE and WRITE_SIZE
passes) -> profiles/pmc_traffic.json, the per-launch HBM bytes bench.py reports as
roofline.traffic.  Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
both counters are KiB; FETCH_SIZE on gfx950 reports half of a wide coalesced read stream and
is doubled; WRITE_SIZE is exact.  The trace holds every workload bench.py runs (epoch step,
64-window minibatch, SVI iterations, whole chain, configs[4]); the launches of the headline
epoch step (K=64, D=32, 3891 windows) are picked by kernel name + launch grid.
Usage: pmc_to_traffic.py profiles/r02c_pmc_hbm.txt [profiles/r02c_pmc_hbm_f32.txt] > profiles/pmc_traffic.json"""
import json
import re
import sys

# family -> (kernel-name regex, workgroups) of the fp64 headline step, and of the fp32-mode step
HEADLINE = {"emission": (r"k_emission_orbit<4, 4(, 2)?>", 7813),
            "forward_backward": (r"k_sweeps_lin<4, true, 0, false(, double)?>", 488),
            "stats": (r"k_stats_mfma4<5, 2, 2, 3, true, false(, double, double)?(, [23])?>", 256),
            "finalize": (r"k_finalize", 161)}
F32 = {"emission": (r"k_emission_bf16x3", 3907),
       "forward_backward": (r"k_sweeps_lin<4, true, 0, false, float>", 488),
       "stats": (r"k_stats_bf16x3.*", 253),
       "finalize": (r"k_finalize", 161)}


def read_rows(path):
    rows = {}   # (kernel, grid) -> {counter: mean KiB}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(\d+)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s*$", line)
        if m:
            rows.setdefault((m.group(1), int(m.group(2))), {})[m.group(3)] = float(m.group(5))
    return rows


def main(path, path32=None):
    rows = read_rows(path)

    def pick(table, rows=rows):
        out = {}
        for fam, (rx, grid) in table.items():
            for (name, g), c in rows.items():
                if g == grid and re.fullmatch(rx, name):
                    fb, wb = c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0, c.get("WRITE_SIZE", 0.0) * 1024.0
                    out[fam] = {"kernel": name, "workgroups": g, "fetch_bytes": fb, "write_bytes": wb,
                                "hbm_bytes_per_launch": fb + wb}
        return out
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, FETCH doubled "
                    "per MI355X_MICROARCH.md (HBM section); bench workload K=64 D=32 Lm=257 B=3891; source %s" % path}
    best = pick(HEADLINE)
    out.update(sorted(best.items()))
    # the epoch step = one launch of each family's kernel (+ the small ones, < 1 %)
    out["_step_total"] = {"hbm_bytes_per_step": sum(v["hbm_bytes_per_launch"] for v in best.values()),
                          "families": sorted(best)}
    if path32:
        f32 = pick(F32, read_rows(path32))
        out["_f32_mode"] = dict(f32, hbm_bytes_per_step=sum(v["hbm_bytes_per_launch"] for v in f32.values()),
                                source=path32)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
