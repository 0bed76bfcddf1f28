#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a gfx950 assembly listing (hipcc -S
--cuda-device-only of a small TU that instantiates just that kernel: seconds instead of the
library's two minutes).  Usage: isa_loop_hist.py file.s <substring of the mangled kernel name>"""
import collections
import re
import sys


def main(path, key):
    s = open(path).read().splitlines()
    start = next(i for i, l in enumerate(s) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
    end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
    body = s[start:end]
    print(s[start].split(":")[0], len(body), "lines")
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.search(r"(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if not (m and m.group(2) in labels and labels[m.group(2)] < i):
            continue
        a, b = labels[m.group(2)], i
        ops = collections.Counter()
        for x in body[a:b]:
            x = x.strip()
            if not x or x[0] in ";." or x.endswith(":"):
                continue
            ops[x.split()[0]] += 1
        nm = sum(v for k, v in ops.items() if "mfma" in k)
        print("loop", m.group(2), "lines", b - a, "instructions", sum(ops.values()), "mfma", nm)
        if nm == 0:
            continue
        grp = collections.Counter()
        for k, v in ops.items():
            g = ("mfma" if "mfma" in k else "valu" if k.startswith("v_") else "lds" if k.startswith("ds_")
                 else "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "salu" if k.startswith("s_") else k)
            grp[g] += v
        print("   groups:", dict(grp))
        for k, v in sorted(ops.items(), key=lambda t: -t[1])[:40]:
            print("     %-28s %d" % (k, v))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
