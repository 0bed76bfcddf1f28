#!/usr/bin/env python3
"""Median period of the device-resident 64-window SVI loop from a rocprofv3 kernel trace (rocpd .db):
the interval between the starts of consecutive k_svi_global_step launches that have a k_wave_lin4
launch between them (works for experimental builds without the ELBO kernels, unlike svi_trace.py)."""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if r[0].startswith("k_svi_global_step") or r[0].startswith("k_svi_step_theta")]
per = [(rows[b][1] - rows[a][1]) / 1e3 for a, b in zip(idx, idx[1:])
       if any(("k_wave_lin" in r[0] or "k_sweep_stats" in r[0]) for r in rows[a:b])]
per = np.array(per)
print("iterations %d  period median %.1f us  p10 %.1f  p90 %.1f" % (len(per), np.median(per), np.percentile(per, 10), np.percentile(per, 90)))
