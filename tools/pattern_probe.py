"""HBM access-pattern probe of the sweeps (window-major vs step-major rows); tools-only library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _probe import probe_fp64
for w, n in ((400, 'window-major (16 x 512 B per step)'), (401, 'step-major (8 KB per step)')):
    print('%-40s %.2f TB/s (1 GB read + 1 GB write)' % (n, probe_fp64(w)))
